#!/bin/bash
# A/B of experiment builds (tools/fastbuild.py tags) in ONE gpurun call, interleaved: the solver kernels' event times from short
# bench.py runs (SW chain and LW chain) per build and round.  usage: tools/ab_solvers.sh <rounds> <tag> [<tag> ...]  ("-" = product)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    tag=$v; [ "$v" = "-" ] && tag=""
    for w in sw lw; do
      RTE_HIP_VARIANT=$tag timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-plain-abi --no-factored --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); pk = r['roofline']['per_kernel']
print('round $r %-8s %-3s step %.3f ms  ' % ('$v', '$w', r['ms_per_step']) + '  '.join('%s %.3f' % (k.replace('_kernel', ''), v['avg_ms']) for k, v in pk.items()))"
    done
  done
done
