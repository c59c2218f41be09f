"""Aggregate a rocprofv3 --kernel-trace --stats summary by kernel family and compare with bench.py's roofline."""
import csv
import json
import sys

stats, bench_json, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
fam = {}
for r in csv.DictReader(open(stats)):
    name = r["Name"]
    key = ("conv direct (conv_igemm*, pool_gemm)" if ("conv_igemm" in name or "pool_gemm" in name) else "conv winograd (wino_f2x2_3x3, incl. the pooling that rides along)" if "wino_f2x2" in name
           else "scan_tiled_kernel" if "scan_tiled_kernel" in name else "scan_kernel (row-major block, before the in-place tiling)" if "scan_kernel" in name
           else "batch_fused_kernel (16-query pass)" if "batch_fused" in name and "true>" in name
           else "batch_fused_kernel on the row-major block" if "batch_fused" in name
           else "pool/gavgpool/preprocess/consensus" if any(k in name for k in ("pool_kernel", "gavgpool", "preprocess", "consensus"))
           else "other")
    f = fam.setdefault(key, [0, 0.0])
    f[0] += int(r["Calls"])
    f[1] += float(r["TotalDurationNs"])
b = json.load(open(bench_json))
print("family, calls, total_ms, avg_us")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%s, %d, %.3f, %.2f" % (k, c, t / 1e6, t / c / 1e3))
c = sum(v[0] for k, v in fam.items() if k.startswith("conv"))
t = sum(v[1] for k, v in fam.items() if k.startswith("conv"))
for k in ("direct", "winograd"):
    fk = [v for kk, v in fam.items() if kk.startswith("conv " + k)][0]
    bf = b["roofline"]["families"][k]
    print("%s: rocprof %d launches, %.4f ms avg; bench.py events %.4f ms avg over %d launches per step"
          % (k, fk[0], fk[1] / fk[0] / 1e6, bf["ms_per_step"] / bf["launches"], bf["launches"]))
print("conv: rocprof avg launch %.4f ms over %d launches (%d per forward); bench.py HIP-event avg launch %.4f ms"
      % (t / c / 1e6, c, b["roofline"]["launches_per_step"], b["roofline"]["avg_launch_ms"]))
if "scan_tiled_kernel" in fam:
    c, t = fam["scan_tiled_kernel"]
    print("scan: rocprof avg launch %.4f ms over %d launches; bench.py HIP-event avg launch %.4f ms"
          % (t / c / 1e6, c, b["similarity"]["roofline"]["avg_launch_ms"]))
bk = [k for k in fam if k.startswith("batch_fused_kernel (16")]
if bk and b.get("similarity", {}).get("batched"):
    c, t = fam[bk[0]]
    print("batched scan: rocprof avg launch %.4f ms over %d launches; bench.py HIP events around a pass (upload + launch) %.4f ms"
          % (t / c / 1e6, c, b["similarity"]["batched"].get("pass_ms_by_hip_events", b["similarity"]["batched"]["ms_per_pass"])))
