#!/bin/bash
# PMC counter passes for bench.py (run on the GPU box via gpurun).  One rocprofv3 invocation per
# counter group, --pmc only with --kernel-trace (the combination gpurun allows); CSV output under
# gpurun_out/pmc/<tag>_<group>/.  usage: bash profiles/pmc_passes.sh <tag> [bench args...]
TAG=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
declare -A G
G[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G[sq2]="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
G[fetch]="FETCH_SIZE TCC_HIT_sum"
G[write]="WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
G[tcp]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
for g in ${PMC_GROUPS:-sq1 sq2 fetch write tcp}; do
  timeout 300 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d gpurun_out/pmc/${TAG}_$g -o p -- \
    python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc/${TAG}_$g.log 2>&1
  echo "$g: $(ls gpurun_out/pmc/${TAG}_$g 2>/dev/null | tr '\n' ' ')"
done
