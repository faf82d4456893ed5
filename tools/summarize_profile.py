#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into the summary committed under profiles/.
   python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>.md [kernel-substring]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
kname = sys.argv[3] if len(sys.argv) > 3 else "nmpc_qp_kernel"
out = [f"# rocprofv3 summary: {os.path.basename(src)}", ""]
for f in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
    out += ["## kernel-trace --stats (all kernels, top 6)", "", "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
    for r in list(csv.DictReader(open(f)))[:6]:
        out.append(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
    out.append("")
try:
    out += ["## bench line of the traced run", "", "```", open(os.path.join(src, "kt.json")).read().strip(), "```", ""]
except OSError:
    pass
vals = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
if vals:
    out += [f"## PMC counters of `{kname}` (mean per dispatch over {max(len(v) for v in vals.values())} dispatches)", "",
            f"dispatch: {json.dumps(meta)}", "",
            "(rocprofv3's `VGPR_Count` is (granulated_workitem_vgpr_count + 1) x 4 of the kernel descriptor: on gfx90a+ the granule of the "
            "unified register file is 8, so the figure is HALF the allocated arch + accumulation registers -- 132 for `.amdhsa_next_free_vgpr 257` "
            "(264 allocated), 128 for 256, 256 for 512 -- and `Accum_VGPR_Count` is not decoded (always 0).  The compiler's own figures are in "
            "profiles/*_isa_resources.md: 231 + 0, 256 + 0, 256 + 255.)", "",
            "| counter | mean per dispatch |", "|---|---|"]
    for k in sorted(vals):
        out.append(f"| {k} | {sum(vals[k]) / len(vals[k]):.4g} |")
    m = {k: sum(v) / len(v) for k, v in vals.items()}
    out.append("")
    if "FETCH_SIZE" in m:
        out.append(f"- FETCH_SIZE {m['FETCH_SIZE']:.4g} KB/dispatch (gfx950: x2 for wide coalesced reads, MI355X_MICROARCH.md HBM section)")
    if "WRITE_SIZE" in m:
        out.append(f"- WRITE_SIZE {m['WRITE_SIZE']:.4g} KB/dispatch")
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        out.append(f"- L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if k in m:
                out.append(f"- {k}/SQ_WAVE_CYCLES = {m[k] / wc:.3f}")
# HBM traffic per launch for bench.py's roofline.traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
# FETCH_SIZE counts 64 B per 128 B request of wide coalesced reads (16 B/lane), i.e. half the bytes --
# doubled here as MI355X_MICROARCH.md (HBM section) prescribes; WRITE_SIZE is exact for 16 B/lane stores.
if vals and "FETCH_SIZE" in m and "WRITE_SIZE" in m and len(sys.argv) > 4:
    key = sys.argv[4]
    tpath = os.path.join(os.path.dirname(os.path.abspath(dst)), "traffic.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        tj = {}
    tj[key] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    tj[key + "_detail"] = {"kernel": kname, "FETCH_SIZE_KiB": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"],
                           "fetch_correction": 2.0, "source": os.path.basename(src)}
    json.dump(tj, open(tpath, "w"), indent=1, sort_keys=True)
    out.append(f"- traffic per launch (2 x FETCH + WRITE) = {tj[key] / 1e6:.1f} MB -> {tpath}")
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out))
