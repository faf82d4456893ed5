#!/usr/bin/env bash
# batch-size sweep of the whole-body bench (one line per size: solves/s, ms per step)
for b in "$@"; do
  python bench.py --workload wholebody --batch $b --steps 10 --warmup 2 --no-cpu-baseline --no-cold-start 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', $b, 'solves/s %.0f' % d['value'], 'ms %.3f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"
done
