#!/usr/bin/env bash
# Is the centroidal QP kernel bound by its workspace traffic?  Timing builds (results wrong, times meaningful), same box:
#   plain | every stage reads the images of ONE stage (cache-resident) | no K~ / Acl~ stores | both
# at B = 1024 (one wave per SIMD, 1.36 GB per launch) and B = 8192 (two waves per SIMD, 14.2 GB per launch).
#   (build the four libraries in the container first: tools/qp_traffic_timing.sh build ; then on the GPU box: ... run)
set -uo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$here/iterative_learning_nmpc_amd/csrc"
mkdir -p "$here/tools/_ab"
if [ "${1:-run}" = "build" ]; then
  i=0
  for flags in "" "-DQP_T_ONEIMG" "-DQP_T_NOSTORE" "-DQP_T_ONEIMG -DQP_T_NOSTORE"; do
    i=$((i+1))
    (cd "$src" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $flags \
        -o "$here/tools/_ab/lib_qpt_$i.so" nmpc_api.hip nmpc_policy.hip nmpc_dataset.hip nmpc_torque.hip) &
  done
  wait
  exit 0
fi
names=("plain" "one cache-resident image" "no gain stores" "both")
for B in 1024 8192; do
  for rep in 1 2; do
    for i in 1 2 3 4; do
      NMPC_HIP_LIB="$here/tools/_ab/lib_qpt_$i.so" python3 "$here/bench.py" --batch $B --no-cpu-baseline --no-cold-start --steps 40 --warmup 5 2>/dev/null | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$B  %-26s rep $rep: kernel_ms %.4f  solves/s %.0f' % ('${names[$((i-1))]}', d['roofline']['kernel_ms'], d['value']))"
    done
  done
done
