cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rm -rf gpurun_out/trace_prod
rocprofv3 --kernel-trace -d gpurun_out/trace_prod --output-format csv -- python3 tools/quick_tsn_bench.py 3:96:3 > gpurun_out/trace_prod.log 2>&1
tail -2 gpurun_out/trace_prod.log
python3 tools/trace_gaps.py gpurun_out/trace_prod/*/*kernel_trace.csv
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/trace_prod/*/*kernel_trace.csv")[0])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "0")) for r in rows))
starts = [i for i, k in enumerate(ks) if "preprocess" in k[2]]
a, b = starts[-4], starts[-2]        # two preprocess launches per forward (one per sub-batch)
seg = ks[a:b]
t0 = seg[0][0]
for s, e, n, q in seg:
    print("%8.1f %8.1f  %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
PY
