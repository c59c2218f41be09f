#!/bin/bash
# A/B of the pool2 fold (pooled-input kernels) on one box
set -e
mkdir -p gpurun_out
python -m pytest tests/test_tsn_gpu.py -q -x -k "pooled_input or layerwise or fused_and_unfused or cfg2_shape or flow_features" > gpurun_out/exp_fold_tests.log 2>&1 || { tail -40 gpurun_out/exp_fold_tests.log; exit 1; }
tail -3 gpurun_out/exp_fold_tests.log
for v in 128 256; do
  VQ_TSN_FOLD_POOL_COUT=$v python tools/layer_table.py 3 96 3 > gpurun_out/lt_fold$v.txt 2>&1
  head -8 gpurun_out/lt_fold$v.txt; tail -1 gpurun_out/lt_fold$v.txt
done
bash tools/bench_ab.sh VQ_TSN_FOLD_POOL_COUT=128 VQ_TSN_FOLD_POOL_COUT=256 2>&1 | tee gpurun_out/exp_fold_ab.txt
