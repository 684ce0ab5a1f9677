#!/bin/bash
# PMC counter passes for any helper script (run on the GPU box via gpurun).  One rocprofv3 invocation per counter
# group, --pmc only with --kernel-trace (the combination gpurun allows); per-kernel means printed as CSV.
# usage: bash profiles/pmc_run.sh <tag> <python script> [args...]      (PMC_GROUPS="a b ..." selects groups)
TAG=$1; SCRIPT=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
declare -A G
G[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
G[b]="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
G[c]="GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES"
G[fetch]="FETCH_SIZE TCC_HIT_sum"
G[write]="WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
mkdir -p gpurun_out/pmc
for g in ${PMC_GROUPS:-a b c fetch write}; do
  timeout 300 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d gpurun_out/pmc/${TAG}_$g -o p -- \
    python $SCRIPT "$@" > gpurun_out/pmc/${TAG}_$g.log 2>&1
done
python profiles/pmc_summarize.py gpurun_out/pmc/${TAG}_* > gpurun_out/pmc/${TAG}_summary.csv
cat gpurun_out/pmc/${TAG}_summary.csv
