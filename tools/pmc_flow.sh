#!/bin/bash
# TV-L1 flow: kernel trace + HBM traffic of the iteration kernel (separate --pmc passes, kernel-trace only) for one batch of 64 pairs of
# 340 x 256 frames (tools/flow_profile.py).  Writes gpurun_out/<tag>_flow_kernel_stats.csv, <tag>_flow_trace_summary.txt, <tag>_flow_summary.json.
set -e
TAG=${1:-r05}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
rm -rf gpurun_out/flow_prof gpurun_out/pmc_flow_fetch gpurun_out/pmc_flow_write
rocprofv3 --kernel-trace --stats -d gpurun_out/flow_prof --output-format csv -- python3 tools/flow_profile.py 64 3 > gpurun_out/flow_prof.log 2>&1
cp gpurun_out/flow_prof/*/*kernel_stats.csv gpurun_out/${TAG}_flow_kernel_stats.csv
python3 tools/flow_trace_summary.py gpurun_out/flow_prof/*/*kernel_trace.csv 3 > gpurun_out/${TAG}_flow_trace_summary.txt
grep pixel-it gpurun_out/flow_prof.log >> gpurun_out/${TAG}_flow_trace_summary.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_flow_fetch --output-format csv -- python3 tools/flow_profile.py 64 2 > gpurun_out/pmc_flow_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_flow_write --output-format csv -- python3 tools/flow_profile.py 64 2 > gpurun_out/pmc_flow_write.log 2>&1
rm -rf gpurun_out/pmc_flow_valu
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES -d gpurun_out/pmc_flow_valu --output-format csv -- python3 tools/flow_profile.py 64 2 > gpurun_out/pmc_flow_valu.log 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
def total(pattern, counter):
    rows = [r for r in csv.DictReader(open(glob.glob(pattern)[0])) if r["Counter_Name"] == counter and ("tvl1_tile_kernel" in r["Kernel_Name"] or "tvl1_block_kernel" in r["Kernel_Name"])]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows)
f, nf = total("gpurun_out/pmc_flow_fetch/*/*counter_collection.csv", "FETCH_SIZE")
w, nw = total("gpurun_out/pmc_flow_write/*/*counter_collection.csv", "WRITE_SIZE")
valu, _ = total("gpurun_out/pmc_flow_valu/*/*counter_collection.csv", "SQ_INSTS_VALU")
gui, _ = total("gpurun_out/pmc_flow_valu/*/*counter_collection.csv", "GRBM_GUI_ACTIVE")
batches = 2
stats = {r["Name"]: r for r in csv.DictReader(open("gpurun_out/%s_flow_kernel_stats.csv" % tag))}
blk = [v for k, v in stats.items() if "tvl1_tile_kernel" in k or "tvl1_block_kernel" in k][0]
line = [l for l in open("gpurun_out/flow_prof.log") if "pixel-iterations" in l][0]
pix_it = float(line.split("pixel-iterations per batch")[1].split(";")[0])
ms = float(blk["TotalDurationNs"]) / 1e6 / 3
out = {"workload": "64 pairs of 340x256 frames, OpenCV default TV-L1 parameters, one batch (tools/flow_profile.py)",
       "kernel": "tvl1_tile_kernel<512> (tiles fitted to the level, two workgroups per compute unit)", "launches_per_batch": int(blk["Calls"]) / 3, "kernel_ms_per_batch": ms,
       "avg_launch_us": float(blk["AverageNs"]) / 1e3, "pixel_iterations_per_batch": pix_it, "pixel_iterations_per_second": pix_it / ms * 1e3,
       "FETCH_SIZE_KB_per_batch": f / batches, "WRITE_SIZE_KB_per_batch": w / batches,
       "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section; calibrated there for 16-byte-per-lane streaming reads -- this kernel "
                     "reads 4 bytes per lane, so the absolute figure is indicative); WRITE_SIZE as is",
       # the bound that matters: vector-ALU issue.  A wave64 VALU instruction occupies its SIMD (16 lanes) for 4 cycles (more for the
       # quarter-rate divisions / square roots, so this UNDER-states the busy time); GRBM_GUI_ACTIVE is summed over the 8 XCDs
       "SQ_INSTS_VALU_per_batch": valu / batches, "valu_wave_insts_per_pixel_iteration": valu / batches / pix_it,
       "valu_thread_insts_per_pixel_iteration": valu / batches / pix_it * 64,
       "valu_issue_utilisation": valu * 4.0 / (gui / 8.0 * 1024.0) if gui else None,
       "hbm_bytes_per_batch": (2 * f + w) / batches * 1024, "hbm_bytes_per_launch": (2 * f + w) / batches * 1024 / (nf / batches),
       "hbm_GBps_during_the_kernel": (2 * f + w) / batches * 1024 / ms / 1e6}
json.dump(out, open("gpurun_out/%s_flow_summary.json" % tag, "w"), indent=1)
print(json.dumps(out))
PY
