# The product's forward (two sub-batch streams) under rocprofv3 --kernel-trace: per forward the span, the time at least
# one kernel runs (union), the sum of kernel durations and the queues -- the evidence that a step's wall time can be
# below the sum of its kernels' durations (profiles/<tag>_two_queue_trace.txt).  Usage: bash tools/trace_prod.sh [tag]
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rm -rf gpurun_out/trace_prod
rocprofv3 --kernel-trace -d gpurun_out/trace_prod --output-format csv -- python3 tools/quick_tsn_bench.py 3:96:3 > gpurun_out/trace_prod.log 2>&1
OUT=gpurun_out/${TAG}_two_queue_trace.txt
{
echo "# rocprofv3 --kernel-trace -- python3 tools/quick_tsn_bench.py 3:96:3   (cfg 2: 96 RGB crops, T = 3; vq_tsn_forward's default: 2 sub-batch streams)"
tail -1 gpurun_out/trace_prod.log
python3 tools/trace_gaps.py gpurun_out/trace_prod/*/*kernel_trace.csv
echo "# one forward, every launch: start_us end_us duration_us queue kernel"
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/trace_prod/*/*kernel_trace.csv")[0])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "0")) for r in rows))
pre = [i for i, k in enumerate(ks) if "preprocess" in k[2]]
starts = [i for n, i in enumerate(pre) if n == 0 or ks[i][0] - ks[pre[n - 1]][0] > 100_000]   # one preprocess launch per sub-batch stream
a, b = starts[-2], starts[-1]
seg = ks[a:b]
t0 = seg[0][0]
for s, e, n, q in seg:
    print("%8.1f %8.1f  %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
PY
} > $OUT
head -12 $OUT
