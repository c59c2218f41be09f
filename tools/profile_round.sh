#!/bin/bash
# One round's profile set (run on the GPU box; copy what should be judged from gpurun_out/ into profiles/):
#   <tag>_bench.json                 python bench.py (default K, W; CPU baselines included)
#   <tag>_bench_under_rocprof.json   bench.py --skip-cpu --profile-only under rocprofv3 --kernel-trace --stats
#   <tag>_bench_kernel_stats.csv     rocprofv3's per-kernel summary of that run
#   <tag>_rocprof_summary.txt        the summary by kernel family next to bench.py's HIP-event averages
#   <tag>_mfma_util.json, <tag>_tsn_traffic.json   separate --pmc passes (tools/pmc_mfma.sh, tools/pmc_tsn.sh)
#   <tag>_flow_kernel_stats.csv, <tag>_flow_trace_summary.txt, <tag>_flow_summary.json   TV-L1: kernel trace + FETCH / WRITE passes (tools/pmc_flow.sh)
set -e
TAG=${1:-r05}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
rm -rf gpurun_out/tune_${TAG}              # the first run times the tilings itself and leaves its tables for the run under the profiler
python3 bench.py --tune-cache gpurun_out/tune_${TAG} --details gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG} --output-format csv -- python3 bench.py --tune-cache gpurun_out/tune_${TAG} --skip-cpu --profile-only \
    > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_bench_under_rocprof.err
STATS=$(ls gpurun_out/prof_${TAG}/*/*kernel_stats.csv | head -1)
cp "$STATS" gpurun_out/${TAG}_bench_kernel_stats.csv
python3 tools/summarize_rocprof.py "$STATS" gpurun_out/${TAG}_bench_under_rocprof.json 50 > gpurun_out/${TAG}_rocprof_summary.txt
cat gpurun_out/${TAG}_rocprof_summary.txt
bash tools/pmc_mfma.sh > gpurun_out/${TAG}_pmc_mfma.log 2>&1 && cp gpurun_out/mfma_util.json gpurun_out/${TAG}_mfma_util.json
bash tools/pmc_tsn.sh > gpurun_out/${TAG}_pmc_tsn.log 2>&1 && cp gpurun_out/tsn_traffic.json gpurun_out/${TAG}_tsn_traffic.json
bash tools/pmc_flow.sh ${TAG} > gpurun_out/${TAG}_pmc_flow.log 2>&1 || true
bash tools/pmc_sim.sh ${TAG} > gpurun_out/${TAG}_pmc_sim.log 2>&1 || true
tail -n 2 gpurun_out/${TAG}_pmc_mfma.log gpurun_out/${TAG}_pmc_tsn.log gpurun_out/${TAG}_pmc_flow.log gpurun_out/${TAG}_pmc_sim.log
