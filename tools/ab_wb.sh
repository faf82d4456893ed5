#!/usr/bin/env bash
# same-box A/B of builds of the library on the whole-body bench: tools/ab_wb.sh <repeats> libA.so libB.so ...
n=$1; shift
for r in $(seq 1 $n); do
  for lib in "$@"; do
    NMPC_HIP_LIB=$PWD/$lib python bench.py --workload wholebody --steps 10 --warmup 2 --no-cpu-baseline --no-cold-start 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '%.0f solves/s' % d['value'], '%.3f ms' % d['ms_per_step'])"
  done
done
