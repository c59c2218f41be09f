#!/bin/bash
# HBM traffic of the similarity scan and of the 16-query batched pass from the L2 memory-side counters (separate --pmc passes,
# kernel-trace only).  Writes gpurun_out/<tag>_scan_traffic.json.
set -e
TAG=${1:-r05}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
cat > gpurun_out/sim_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import video_query_algorithms_amd as vqa
db = vqa.FeatureDB.synthetic(1_000_000, 2, 5, 1024, seed=17, scales=(4.0, 1.0))
db.set_query_from_row(12345, want=False)
rng = np.random.default_rng(0)
t = rng.standard_normal((16, 2, 5, 1024)) / 1024
w = 0.5 + rng.random((16, 2))
for _ in range(3):                      # the row-major block as loaded (scan_kernel, batch_fused_kernel<..., false>)
    db.scan(weights=[1.0, 1.5])
db.scan_batch(t, w, want=False)
db.set_layout("tiled")                  # in place: mirror_build_kernel + device copies, no second block
for _ in range(3):                      # scan_tiled_kernel, batch_fused_kernel<..., true>
    db.scan(weights=[1.0, 1.5])
db.scores()
for _ in range(4):
    db.scan_batch(t, w, want=False)
db.scan_batch(t[:1], w[:1])
PY
rm -rf gpurun_out/pmc_sim_fetch gpurun_out/pmc_sim_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_sim_fetch --output-format csv -- python3 gpurun_out/sim_once.py > gpurun_out/pmc_sim_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_sim_write --output-format csv -- python3 gpurun_out/sim_once.py > gpurun_out/pmc_sim_write.log 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
def per_launch(pattern, counter, key, skip_small):
    rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(glob.glob(pattern)[0])) if r["Counter_Name"] == counter and key in r["Kernel_Name"]]
    rows = [v for v in rows if not skip_small or v > 0.5 * max(rows)]          # (the 1-query pass at the end of the script is not a 16-query pass)
    return sum(rows) / len(rows), len(rows)
out = {"workload": "cfg 4: 1M clips x 2 streams x 5 splits x 1024 fp32 = 40.96 GB on one GPU",
       "correction": "gfx950: FETCH_SIZE x2 for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is"}
for name, key, algo in (("scan_tiled_kernel", "scan_tiled_kernel", 1_000_000 * 2 * 5 * 1024 * 4 + 1_000_000 * 8),
                        ("scan_kernel", "scan_kernel<", 1_000_000 * 2 * 5 * 1024 * 4 + 1_000_000 * 8),
                        ("batch_fused_kernel", "8, true>", 1_000_000 * 2 * 5 * 1024 * 4 + 16 * 1_000_000 * 8),
                        ("batch_fused_kernel_row_major", "8, false>", 1_000_000 * 2 * 5 * 1024 * 4 + 16 * 1_000_000 * 8)):
    f, n = per_launch("gpurun_out/pmc_sim_fetch/*/*counter_collection.csv", "FETCH_SIZE", key, True)
    w, _ = per_launch("gpurun_out/pmc_sim_write/*/*counter_collection.csv", "WRITE_SIZE", key, True)
    out[name] = {"launches": n, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                 "algorithmic_bytes_per_launch": algo, "ratio": (2 * f + w) * 1024 / algo}
try:
    f, n = per_launch("gpurun_out/pmc_sim_fetch/*/*counter_collection.csv", "FETCH_SIZE", "mirror_build_kernel", False)
    w, _ = per_launch("gpurun_out/pmc_sim_write/*/*counter_collection.csv", "WRITE_SIZE", "mirror_build_kernel", False)
    out["mirror_build_kernel"] = {"launches": n, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                                  "note": "vq_db_set_layout(TILED): runs of tiles re-ordered through a 256 MB scratch block, then copied back in place"}
except (IndexError, ZeroDivisionError):
    pass
out["hbm_bytes_per_launch"] = out["scan_tiled_kernel"]["hbm_bytes_per_launch"]
json.dump(out, open("gpurun_out/%s_scan_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out))
PY
