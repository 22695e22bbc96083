#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03md
O=gpurun_out/r03md
for w in ns c3 c5; do
  timeout -k 10 300 python tools/md_bench.py --workload $w --steps 300 > $O/md_$w.log 2>&1 || exit 1
  head -2 $O/md_$w.log
done
timeout -k 10 300 python tools/md_bench.py --steps 300 --buffer 0.6 > $O/md_ns_b06.log 2>&1 || exit 1
head -1 $O/md_ns_b06.log
