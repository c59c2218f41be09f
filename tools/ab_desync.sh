#!/bin/bash
for v in 0 1; do echo "VQ_TSN_DESYNC=$v"; VQ_TSN_DESYNC=$v python tools/layer_table.py 3 96 3 2>&1 | tail -1; done
for v in 0 1; do echo "VQ_TSN_DESYNC=$v (forced 128x96x32)"; VQ_TSN_DESYNC=$v VQ_TSN_TILE=128x96x32 python tools/layer_table.py 3 96 3 2>&1 | grep -E "conv2/3x3 |total"; done
