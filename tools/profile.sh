#!/usr/bin/env bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).
#   tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/{kt,pmc_*}/...csv
# Kernel-trace/stats and each PMC group are separate runs (counters never share a run with a trace
# domain other than --kernel-trace).  Summarise with tools/summarize_profile.py.
set -uo pipefail
tag="${1:-r01}"
shift || true
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
export TMPDIR=/tmp
bench=(python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cold-start "$@")
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- "${bench[@]}" > "$out/kt.json" 2> "$out/kt.err"
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_WAVES" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/pmc_$i" -- "${bench[@]}" > "$out/pmc_$i.json" 2> "$out/pmc_$i.err" || echo "pass $i failed"
done
find "$out" -name "*.csv" | head -40
