#!/bin/bash
for t in ${TILES:-128x128x3x2 128x128x4x2 128x64x3x2 64x64x3x2 128x128x32x0 128x96x32x0}; do echo "TILE=$t"; VQ_TSN_TILE=$t python tools/layer_table.py 3 96 3 2>&1 | grep -E "conv2/3x3 |inception_3a/double_3x3_2|inception_4d/double_3x3_2|inception_5a/3x3 |total"; done
