#!/usr/bin/env bash
# Throughput of the two QP kernel variants over the batch size (run on the GPU box).
for v in resident lean; do
  for b in "$@"; do
    NMPC_QP_VARIANT=$v python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --batch $b 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v B=$b: %.3f ms/step  %.0f solves/s' % (d['ms_per_step'], d['value']))"
  done
done
