#!/usr/bin/env bash
# Diagnostic (GPU box): kernel durations of the solve at 1, 2, 3, 6 interior-point iterations ->
# per-iteration and fixed cost of nmpc_qp_kernel and the linearisation kernel (rocprofv3 kernel trace).
export TMPDIR=/tmp
for k in 1 2 3 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fixed_$k -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --ipm $k > /dev/null 2>&1
  f=$(find gpurun_out/fixed_$k -name "*kernel_stats.csv" | head -n 1)
  echo "n_ipm=$k"; grep -E "nmpc_(qp|linearize|shift)" "$f" | awk -F, '{printf "   %-60s avg %8.1f us\n", substr($1,1,60), $4/1000}'
done
