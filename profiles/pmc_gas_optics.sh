#!/bin/bash
# PMC counter passes for the LW gas-optics kernels only (tools/time_gas_optics.py).  usage: bash profiles/pmc_gas_optics.sh <tag>
TAG=${1:-go}; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
declare -A G
G[a]="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
G[b]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
G[c]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
G[d]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_ANY SQ_LEVEL_WAVES"
for g in ${PMC_GROUPS:-a b c d}; do
  timeout 300 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d gpurun_out/pmc/${TAG}_$g -o p -- \
    python tools/time_gas_optics.py "$@" > gpurun_out/pmc/${TAG}_$g.log 2>&1
  echo "$g: $(ls gpurun_out/pmc/${TAG}_$g 2>/dev/null | tr '\n' ' ')"
done
python profiles/pmc_summarize.py gpurun_out/pmc/${TAG}_* > gpurun_out/pmc/${TAG}_summary.csv
