#!/bin/bash
# Hardware-counter passes over tools/kbench.py on the GPU box (one rocprofv3 --pmc run per pass,
# counters only; see MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage, from the repo root:
#   tools/pmc_passes.sh <out-subdir> <kbench args...>
# writes gpurun_out/<out-subdir>/<pass>/..._counter_collection.csv and a per-kernel summary.
set -u
out=gpurun_out/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
passes=(
 "sq1:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"
 "sq2:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
 "sq3:SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_FLAT"
 "fetch:FETCH_SIZE"
 "write:WRITE_SIZE"
 "grbm:GRBM_GUI_ACTIVE GRBM_TA_BUSY"
)
for p in "${passes[@]}"; do
  name=${p%%:*}; ctrs=${p#*:}
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/$name" -o p -- python tools/kbench.py "$@" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; exit 1; }
done
python tools/pmc_summary.py "$out"
