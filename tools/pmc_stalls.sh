#!/bin/bash
# Where the waves of the convolution launches of a cfg-2 step spend their cycles, from the SQ counters (three --pmc passes, kernel-trace only;
# same workload and tiling file as tools/pmc_mfma.sh).  Per kernel family: issue / wait shares of the wave cycles, LDS bank conflicts per
# LDS cycle, matrix-pipe busy and co-execution cycles.  -> gpurun_out/stalls.json
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
test -f gpurun_out/tiles_cfg2.json || python3 bench.py --tiles gpurun_out/tiles_cfg2.json --skip-sim --skip-cpu --steps 3 > /dev/null 2>&1
pass() {
  rm -rf gpurun_out/pmc_stalls_$1
  shift_name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_stalls_$shift_name --output-format csv \
    -- python3 bench.py --tiles gpurun_out/tiles_cfg2.json --skip-sim --skip-cpu --profile-only --steps 5 --warmup 1 > gpurun_out/pmc_stalls_$shift_name.log 2>&1
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
pass b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for p in "abc":
    files = glob.glob("gpurun_out/pmc_stalls_%s/*/*counter_collection.csv" % p)
    if not files:
        continue
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        f = "wino_f2x2_3x3" if "wino_f2x2" in k else "conv_igemm_pipe" if "conv_igemm_pipe" in k else "conv_igemm" if "conv_igemm" in k else "pool_gemm" if "pool_gemm" in k else None
        if f:
            fam[f][r["Counter_Name"]] += float(r["Counter_Value"])
    for f, c in fam.items():
        for name, v in c.items():
            out[f][name + ("" if name not in out[f] else "_pass_" + p)] = v
res = {}
for f, c in out.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    r = {"counters": c}
    r["share_of_wave_cycles"] = {k: c[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS") if k in c}
    wc_b = c.get("SQ_WAVE_CYCLES_pass_b", c.get("SQ_WAVE_CYCLES", 1.0))
    r["issue_share_of_wave_cycles"] = {k: c[k] / wc_b for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                                                                  "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC") if k in c}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        r["lds_bank_conflict_cycles_per_active_cycle"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    res[f] = r
json.dump(res, open("gpurun_out/stalls.json", "w"), indent=1)
print(json.dumps({f: {k: v for k, v in r.items() if k != "counters"} for f, r in res.items()}, indent=1))
PY
