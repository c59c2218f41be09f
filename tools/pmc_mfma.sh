#!/bin/bash
# Matrix-pipe utilisation of the conv launches of a cfg-2 step from the SQ counters (one --pmc pass, kernel-trace only):
#   utilisation = SQ_INSTS_MFMA x 64 cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), per kernel family (rocprofv3 reports
#   GRBM_GUI_ACTIVE summed over the 8 XCDs: 1.76 M per launch for kernels that last ~90 us), over the profiled forwards
# of `bench.py --profile-only` (one stream, tilings from gpurun_out/tiles_cfg2.json so that no autotune launch is counted).
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
test -f gpurun_out/tiles_cfg2.json || python3 bench.py --tiles gpurun_out/tiles_cfg2.json --skip-sim --skip-cpu --steps 3 > /dev/null 2>&1
rm -rf gpurun_out/pmc_mfma
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d gpurun_out/pmc_mfma --output-format csv \
  -- python3 bench.py --tiles gpurun_out/tiles_cfg2.json --skip-sim --skip-cpu --profile-only --steps 5 --warmup 1 > gpurun_out/pmc_mfma.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
rows = list(csv.DictReader(open(glob.glob("gpurun_out/pmc_mfma/*/*counter_collection.csv")[0])))
fam = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    f = "wino_f2x2_3x3" if "wino_f2x2" in k else "conv_igemm_pipe" if "conv_igemm_pipe" in k else "conv_igemm" if "conv_igemm" in k else "pool_gemm" if "pool_gemm" in k else None
    if f is None:
        continue
    fam[f][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_MFMA":
        n[f] += 1
out = {}
for f, c in fam.items():
    util = c["SQ_INSTS_MFMA"] * 64.0 / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if c["GRBM_GUI_ACTIVE"] else None
    out[f] = {"launches": n[f], "SQ_INSTS_MFMA": c["SQ_INSTS_MFMA"], "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"],
              "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "valu_per_mfma": c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"] if c["SQ_INSTS_MFMA"] else None,
              "matrix_pipe_utilisation": util}
json.dump(out, open("gpurun_out/mfma_util.json", "w"), indent=1)
print(json.dumps(out))
PY
