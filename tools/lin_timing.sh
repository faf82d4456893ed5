#!/usr/bin/env bash
# Diagnostic (GPU box): duration of nmpc_linearize_kernel for timing builds of the library (wrong results,
# one cost removed each): libnmpc_t_<VARIANT>.so built with -DLIN_T_<VARIANT>.
export TMPDIR=/tmp
for lib in iterative_learning_nmpc_amd/libnmpc_hip.so iterative_learning_nmpc_amd/libnmpc_t_*.so; do
  tag=$(basename $lib .so)
  NMPC_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lin_$tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/lin_{tag}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "nmpc_" in r["Name"]:
        print(f"{tag:24s} {r['Name'][:40]:42s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
done
