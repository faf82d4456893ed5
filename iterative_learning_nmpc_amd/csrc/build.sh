#!/usr/bin/env bash
# Build libnmpc_hip.so (the C-ABI of include/nmpc.h, include/nmpc_policy.h, include/nmpc_dataset.h and include/nmpc_torque.h) for gfx950 in-tree.
# hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libnmpc_hip.so"
# -amdgpu-mfma-vgpr-form: MFMA operands stay in VGPRs when a kernel also has AGPRs (nmpc_solve.hip, nmpc_qp_kernel)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
      -fno-gpu-rdc -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
      "$@" -o "${out}" "${here}/nmpc_api.hip" "${here}/nmpc_policy.hip" "${here}/nmpc_dataset.hip" "${here}/nmpc_torque.hip"
echo "built ${out}"
