#!/usr/bin/env bash
# Diagnostic: duration of nmpc_linearize_kernel for timing builds of the library (wrong results, one cost
# removed each: -DLIN_T_NOFLUSH / NOSTORE / NOTAIL in nmpc_solve.hip).
#   tools/lin_timing.sh build     (anywhere: hipcc cross-compiles)  -> iterative_learning_nmpc_amd/libnmpc_t_<VARIANT>.so
#   tools/lin_timing.sh           (GPU box) rocprofv3 averages per library; delete the variants afterwards
if [ "${1:-}" = build ]; then
  cd "$(dirname "$0")/../iterative_learning_nmpc_amd/csrc" || exit 1
  for v in NOFLUSH NOSTORE NOTAIL; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DLIN_T_$v \
          -o ../libnmpc_t_$v.so nmpc_api.hip nmpc_policy.hip nmpc_dataset.hip nmpc_torque.hip || exit 1
  done
  exit 0
fi
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
