#!/bin/bash
# Timing-only ablations (results are wrong by construction): VQ_TSN_DBG bit0 = no activation traffic, bit1 = no filter traffic
for v in ${DBGS:-0 1 2 3}; do echo "VQ_TSN_DBG=$v"; VQ_TSN_DBG=$v python tools/layer_table.py 3 96 3 2>&1 | grep -E "conv2/3x3 |inception_4d/double_3x3_2|inception_3a/double_3x3_2|inception_5a/double_3x3_2|total"; done
