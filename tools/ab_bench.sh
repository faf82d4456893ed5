#!/usr/bin/env bash
# A/B timing of build variants on the GPU box (one process per variant, same device):
#   tools/ab_bench.sh "<flags A>" "<flags B>" ...
# Each variant is compiled into gpurun_out/ and bench.py is run against it through NMPC_HIP_LIB.
set -uo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "$here/gpurun_out"
i=0
for flags in "$@"; do
  i=$((i+1))
  lib="$here/gpurun_out/libnmpc_ab_$i.so"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -mllvm -amdgpu-mfma-vgpr-form -o "$lib" "$here"/iterative_learning_nmpc_amd/csrc/nmpc_{api,policy,dataset,torque}.hip 2>/dev/null || { echo "variant $i [$flags]: BUILD FAILED"; continue; }
  for rep in 1 2; do
    NMPC_HIP_LIB="$lib" python3 "$here/bench.py" --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $i [$flags] rep $rep: kernel_ms %.4f  solves/s %.0f' % (d['roofline']['kernel_ms'], d['value']))"
  done
done
