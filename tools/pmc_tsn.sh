#!/bin/bash
# HBM traffic of the convolution kernels of one cfg-2 step (separate --pmc passes, kernel-trace only; the program runs
# with --profile-only: every forward on one stream, no autotune if gpurun_out/tiles_cfg2.json exists).
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
test -f gpurun_out/tiles_cfg2.json || python3 bench.py --tiles gpurun_out/tiles_cfg2.json --skip-sim --skip-cpu --steps 3 > /dev/null 2>&1
rm -rf gpurun_out/pmc_tsn_fetch gpurun_out/pmc_tsn_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_tsn_fetch --output-format csv -- python3 bench.py --tiles gpurun_out/tiles_cfg2.json --skip-sim --skip-cpu --profile-only --steps 5 --warmup 1 > gpurun_out/pmc_tsn_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_tsn_write --output-format csv -- python3 bench.py --tiles gpurun_out/tiles_cfg2.json --skip-sim --skip-cpu --profile-only --steps 5 --warmup 1 > gpurun_out/pmc_tsn_write.log 2>&1
python3 - <<'PY'
import csv, glob, json
def total(pattern, counter):
    rows = [r for r in csv.DictReader(open(glob.glob(pattern)[0])) if r["Counter_Name"] == counter]
    conv = [r for r in rows if "conv_igemm" in r["Kernel_Name"] or "wino_f2x2" in r["Kernel_Name"] or "pool_gemm" in r["Kernel_Name"]]
    forwards = sum(1 for r in rows if "gavgpool" in r["Kernel_Name"])        # one global-pool launch per forward
    return sum(float(r["Counter_Value"]) for r in conv), len(conv), forwards
f, nf, fw_f = total("gpurun_out/pmc_tsn_fetch/*/*counter_collection.csv", "FETCH_SIZE")
w, nw, fw_w = total("gpurun_out/pmc_tsn_write/*/*counter_collection.csv", "WRITE_SIZE")
per_step = nf / fw_f
out = {"workload": "cfg 2: 96 crops per step, all convolution launches of a step (%.0f: the Winograd launches carry sibling layers and the pooling)" % per_step,
       "launches_counted": nf, "steps": fw_f, "conv_launches_per_step": per_step,
       "FETCH_SIZE_KB_per_step": f / fw_f, "WRITE_SIZE_KB_per_step": w / fw_w,
       "correction": "gfx950: FETCH_SIZE x2 for 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is",
       "hbm_bytes_per_step": (2 * f / fw_f + w / fw_w) * 1024, "hbm_bytes_per_launch": (2 * f / fw_f + w / fw_w) * 1024 / per_step}
json.dump(out, open("gpurun_out/tsn_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
