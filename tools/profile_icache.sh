#!/usr/bin/env bash
# Instruction-cache counters of the solve kernels (run on the GPU box through gpurun):
#   tools/profile_icache.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/icache/...csv
set -uo pipefail
tag="${1:-r03}"
shift || true
out="gpurun_out/prof_${tag}/icache"
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$out" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold-start --headline-only "$@" > "$out.json" 2> "$out.err" || echo "icache pass failed"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, v in acc.items():
    d = max(n[k], 1)
    print(k, {c: round(x / d) for c, x in v.items()})
PY
