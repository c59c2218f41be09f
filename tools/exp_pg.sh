#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_tsn_gpu.py -q -x -k "pooled_input or layerwise" > gpurun_out/exp_pg_tests.log 2>&1 || { tail -40 gpurun_out/exp_pg_tests.log; exit 1; }
tail -2 gpurun_out/exp_pg_tests.log
VQ_TUNE_CACHE=0 python tools/layer_table.py 3 96 3 2>/dev/null | grep -E "conv1|conv2/3x3_reduce|inception_3a/1x1|^total"
bash tools/bench_ab.sh VQ_TSN_FOLD_POOL_COUT=128 VQ_TSN_FOLD_POOL_COUT=256 2>&1
