#!/bin/bash
# GPU-box helper: sample sclk / power with rocm-smi every 0.1 s beside a command; prints the histogram of (sclk, power) samples.
# usage: tools/clocks_beside.sh <tag> <command...>
TAG=$1; shift
( for i in $(seq 1 300); do echo "$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power \(W\)' | sed -E 's/.*\((.*)\).*/\1/; s/.*: ([0-9.]+)$/\1W/' | tr '\n' ' ')"; sleep 0.1; done ) > gpurun_out/clk_$TAG.txt &
SMP=$!
"$@"
kill $SMP 2>/dev/null
echo "== $TAG: samples (count sclk power)"; sort gpurun_out/clk_$TAG.txt | uniq -c | sort -rn | head -8
