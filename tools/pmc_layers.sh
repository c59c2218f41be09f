#!/bin/bash
# PMC counters of the conv kernels (one rocprofv3 pass per counter group; --pmc never combined with tracing of
# other domains).  Output CSVs land in gpurun_out/pmc_<tag>_<group>/.
set -e
TAG=${1:-r1}
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
  -d gpurun_out/pmc_${TAG}_sq --output-format csv -- python3 tools/layer_table.py 3 96 3 > gpurun_out/pmc_${TAG}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_WAVES \
  -d gpurun_out/pmc_${TAG}_sq2 --output-format csv -- python3 tools/layer_table.py 3 96 3 > gpurun_out/pmc_${TAG}_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum \
  -d gpurun_out/pmc_${TAG}_tcc --output-format csv -- python3 tools/layer_table.py 3 96 3 > gpurun_out/pmc_${TAG}_tcc.log 2>&1
ls gpurun_out/pmc_${TAG}_*/*/ | head
