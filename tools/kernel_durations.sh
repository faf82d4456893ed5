#!/usr/bin/env bash
# Diagnostic (GPU box): average kernel durations of bench.py at the given batch sizes (rocprofv3 --stats).
export TMPDIR=/tmp
for b in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dur_$b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch $b > /dev/null 2>&1
  python3 - "$b" <<'PY'
import csv, glob, sys
b = sys.argv[1]
f = glob.glob(f"gpurun_out/dur_{b}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "nmpc_" in r["Name"]:
        print(f"B={b:>5} {r['Name'][:48]:50s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
done
