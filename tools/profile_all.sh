#!/usr/bin/env bash
# Round profiles (run on the GPU box through gpurun): kernel trace + PMC passes of the headline, the same at B = 8192 (two waves per SIMD), the whole-body workload,
# its mixed-precision variant, and a kernel trace of the rollout mode.  Summaries -> profiles/ by tools/summarize_profile.py.
set -uo pipefail
tag="${1:-r02}"
export TMPDIR=/tmp
bash tools/profile.sh ${tag}_c; echo "headline done"
bash tools/profile.sh ${tag}_c8k --batch 8192; echo "c8k done"
bash tools/profile.sh ${tag}_wb --workload wholebody --steps 5 --warmup 1
bash tools/profile.sh ${tag}_wbp3 --workload wholebody --precision 3 --steps 5 --warmup 1
mkdir -p gpurun_out/prof_${tag}_roll
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_roll/kt -- python3 bench.py --rollouts 8192 --steps 2 --warmup 1 > gpurun_out/prof_${tag}_roll/kt.json 2> gpurun_out/prof_${tag}_roll/kt.err
echo done
