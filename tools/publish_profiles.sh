#!/usr/bin/env bash
# gpurun_out/prof_<tag>_* (tools/profile_all.sh) -> the summaries committed under profiles/ and profiles/traffic.json
#   tools/publish_profiles.sh r03
set -euo pipefail
# (gpurun MERGES a call's files into gpurun_out/: after a second profile run of the same tag, delete the older run's files first --
#  find gpurun_out/prof_<tag>_* -type f -mmin +N -delete -- or the summaries average two builds)
tag="$1"
python tools/summarize_profile.py gpurun_out/prof_${tag}_c profiles/${tag}_solve_b1024.md nmpc_qp_kernel > /dev/null
python tools/summarize_profile.py gpurun_out/prof_${tag}_c8k profiles/${tag}_solve_b8192.md nmpc_qp_kernel > /dev/null
python tools/summarize_profile.py gpurun_out/prof_${tag}_wb profiles/${tag}_wholebody_b8192.md nmpc_wb_qp_kernel > /dev/null
python tools/summarize_profile.py gpurun_out/prof_${tag}_wbp3 profiles/${tag}_wholebody_p3.md nmpc_wb_qp_kernel > /dev/null
python tools/summarize_profile.py gpurun_out/prof_${tag}_roll profiles/${tag}_rollouts_b8192.md nmpc_qp_kernel > /dev/null
python tools/traffic_from_profile.py gpurun_out/prof_${tag}_c B1024_ipm6_sqp1_p0 nmpc_qp_kernel
python tools/traffic_from_profile.py gpurun_out/prof_${tag}_c8k B8192_ipm6_sqp1_p0 nmpc_qp_kernel
python tools/traffic_from_profile.py gpurun_out/prof_${tag}_wb wb_B8192_ipm6_sqp1_p0 nmpc_wb_qp_kernel
python tools/traffic_from_profile.py gpurun_out/prof_${tag}_wbp3 wb_B8192_ipm6_sqp1_p3 nmpc_wb_qp_kernel
