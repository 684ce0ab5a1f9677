#!/bin/bash
# GPU-box helper: the unchanged frontend on host arrays (8 OpenMP threads, host-mirror mode) with the GPU's clocks sampled beside
# it: are the slow PHASES of a run (0.06 s vs 0.28 s passes) times at which the device sits in a low power state?
# usage: tools/host_array_clocks.sh [starts] [passes]
S=${1:-3}; P=${2:-8}
( for i in $(seq 1 400); do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power \(W\)' | sed -E 's/.*\((.*)\).*/\1/; s/.*: ([0-9.]+)$/\1W/' | tr '\n' ' ')"; sleep 0.05; done ) > gpurun_out/clk_samples.txt &
SMP=$!
python tools/host_array_spread.py $S $P 8 RTE_HIP_BIND_NUMA=1 REF_DRIVER_TIMING=1
kill $SMP 2>/dev/null
awk '{print $2, $3, $4, $5}' gpurun_out/clk_samples.txt | sort | uniq -c | sort -rn | head -12
