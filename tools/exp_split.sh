#!/bin/bash
# production-mode sub-batch splits, with and without the pool2 fold, on one box
mkdir -p gpurun_out
bash tools/bench_ab.sh "VQ_TSN_SPLIT=2 VQ_TSN_FOLD_POOL_COUT=128" "VQ_TSN_SPLIT=2 VQ_TSN_FOLD_POOL_COUT=256" "VQ_TSN_SPLIT=3 VQ_TSN_FOLD_POOL_COUT=128" "VQ_TSN_SPLIT=3 VQ_TSN_FOLD_POOL_COUT=256" "VQ_TSN_SPLIT=4 VQ_TSN_FOLD_POOL_COUT=128" "VQ_TSN_SPLIT=2,1 VQ_TSN_FOLD_POOL_COUT=128" 2>&1 | tee gpurun_out/exp_split_ab.txt
