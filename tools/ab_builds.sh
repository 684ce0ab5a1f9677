#!/bin/bash
# A/B of experiment builds (tools/fastbuild.py tags) in ONE gpurun call, interleaved: prints the tau / Planck kernel times of
# tools/time_gas_optics.py per build and round.  usage: tools/ab_builds.sh <rounds> <tag> [<tag> ...]   ("-" = the product library)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    tag=$v; [ "$v" = "-" ] && tag=""
    RTE_HIP_VARIANT=$tag timeout 300 python tools/time_gas_optics.py 2>&1 | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip()); print('round $r %-10s tau %.3f  planck %.3f  interp %.3f' % ('$v', d['tau_absorption_kernel'], d['planck_source_kernel'], d['interpolation_kernel']))"
  done
done
