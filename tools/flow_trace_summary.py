"""Summarise a rocprofv3 --kernel-trace of tools/flow_profile.py: per kernel (and per pyramid level for the iteration kernel) calls,
time per batch, and the gaps between consecutive dispatches.   python tools/flow_trace_summary.py <kernel_trace.csv> <batches>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "tvl1_block" in name or "tvl1_tile" in name or "tvl1_primal" in name or "tvl1_dual" in name:
        name += " grid %sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d[name][0] += 1
    d[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(v[1] for v in d.values())
print("kernel, launches per batch, ms per batch, avg us")
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
    print("%s, %.1f, %.3f, %.1f" % (k, v[0] / batches, v[1] / 1e3 / batches, v[1] / v[0]))
print("all kernels: %.1f launches and %.2f ms busy per batch; first dispatch to last end %.2f ms per batch (incl. the host's part between batches)"
      % (len(rows) / batches, busy / 1e3 / batches, (t1 - t0) / 1e6 / batches))
