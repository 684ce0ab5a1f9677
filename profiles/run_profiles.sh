#!/bin/bash
# One round's evidence (run on the GPU box via gpurun): rocprofv3 kernel-trace stats of the three bench workloads,
# the bench JSON lines, and the HBM-traffic PMC passes of the headline command.  Outputs under gpurun_out/prof/<tag>*;
# copy the summaries into profiles/.   usage: bash profiles/run_profiles.sh <tag>
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof gpurun_out/pmc
for w in lw sw allsky; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-secondary $( [ $w = lw ] || echo --no-cpu-baseline ) > gpurun_out/prof/${TAG}_${w}_bench.json 2> gpurun_out/prof/${TAG}_${w}_bench.err
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_${w} -- python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-plain-abi --no-factored --no-secondary > gpurun_out/prof/${TAG}_${w}_prof.log 2>&1
  DB=$(ls gpurun_out/prof/${TAG}_${w}_results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python profiles/summarize.py "$DB" gpurun_out/prof/${TAG}_${w}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-plain-abi --no-factored --no-secondary" > /dev/null
done
# HBM traffic of the headline chain: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), --pmc with --kernel-trace only
for g in fetch write; do
  if [ $g = fetch ]; then C="FETCH_SIZE TCC_HIT_sum"; else C="WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; fi
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc/${TAG}_$g -o p -- \
    python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-plain-abi --no-factored --no-secondary > gpurun_out/pmc/${TAG}_$g.log 2>&1
done
python profiles/pmc_summarize.py gpurun_out/pmc/${TAG}_fetch gpurun_out/pmc/${TAG}_write > gpurun_out/prof/${TAG}_pmc_summary.csv
# SQ counters (issue / wait / LDS) and traffic of the LW chain, the SW chain (60 layers) and the 72-layer solvers
PMC_GROUPS="a b" bash profiles/pmc_run.sh ${TAG}lwsq bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-plain-abi --no-factored --no-secondary > /dev/null 2>&1
cp gpurun_out/pmc/${TAG}lwsq_summary.csv gpurun_out/prof/${TAG}_lw_sq_summary.csv
PMC_GROUPS="a b fetch write" bash profiles/pmc_run.sh ${TAG}sw tools/time_sw.py > /dev/null 2>&1
cp gpurun_out/pmc/${TAG}sw_summary.csv gpurun_out/prof/${TAG}_sw_pmc_summary.csv
PMC_GROUPS="a b fetch write" bash profiles/pmc_run.sh ${TAG}l72 tools/time_72_layers.py > /dev/null 2>&1
cp gpurun_out/pmc/${TAG}l72_summary.csv gpurun_out/prof/${TAG}_72layers_pmc_summary.csv
# the LW step with factored sources beside the ABI chain (tools/time_lw_factored.py runs both)
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_lwfact -- python tools/time_lw_factored.py > gpurun_out/prof/${TAG}_lwfact.log 2>&1
DB=$(ls gpurun_out/prof/${TAG}_lwfact_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python profiles/summarize.py "$DB" gpurun_out/prof/${TAG}_lwfact_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tools/time_lw_factored.py" > /dev/null
ls gpurun_out/prof | grep ${TAG}
