#!/bin/bash
for v in ${DBGS:-0 8 16}; do echo "VQ_TSN_DBG=$v (forced ${TILE:-128x96x32})"; VQ_TSN_DBG=$v VQ_TSN_TILE=${TILE:-128x96x32} python tools/layer_table.py 3 96 3 2>&1 | grep -E "conv2/3x3 |inception_4d/double_3x3_2|inception_3a/double_3x3_2|total"; done
