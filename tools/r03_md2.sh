#!/bin/bash
mkdir -p gpurun_out/r03md
for k in 1 2; do
  timeout -k 10 300 python tools/md_bench.py --steps 300 2>&1 | grep -E "N=|sorts" | cut -c1-120 || exit 1
  timeout -k 10 300 python tools/md_bench.py --steps 1000 2>&1 | grep -E "N=|sorts" | cut -c1-120 || exit 1
done
timeout -k 10 300 python tools/md_bench.py --steps 1000 --buffer 0.6 2>&1 | grep -E "N=|sorts" | cut -c1-120
