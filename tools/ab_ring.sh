#!/bin/bash
for v in 0 1 4 5 32 37; do echo "DBG=$v TILE=128x128x3x2"; VQ_TSN_DBG=$v VQ_TSN_TILE=128x128x3x2 python tools/layer_table.py 3 96 3 2>&1 | grep -E "conv2/3x3 |inception_4e/double_3x3_1"; done
