#!/usr/bin/env bash
# same-box A/B of builds of the library on the centroidal bench (B = 1024 headline and B = 8192): tools/ab_c.sh <repeats> libA.so libB.so ...
n=$1; shift
for r in $(seq 1 $n); do
  for lib in "$@"; do
    for B in 1024 8192; do
      NMPC_HIP_LIB=$PWD/$lib python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-cold-start --headline-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'B=$B', '%.0f solves/s' % d['value'], '%.4f ms' % d['ms_per_step'])"
    done
  done
done
