#!/usr/bin/env python3
"""Register / LDS / scratch footprint of every kernel of libnmpc_hip.so as the compiler reports it (kept under profiles/
next to the rocprof summaries: occupancy claims in DESIGN.md are checked against this file, not against memory).
    python tools/isa_resources.py > profiles/rNN_isa_resources.md"""
import os, re, subprocess

src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "iterative_learning_nmpc_amd", "csrc")
print("# Kernel resources (hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form, -Rpass-analysis=kernel-resource-usage)\n")
print("(the dispatch records of rocprofv3 show other numbers for the same kernels -- `VGPR_Count` 132 / 128 / 256, `Accum_VGPR_Count` 0: it decodes the\n"
      "kernel descriptor's granulated register count with a granule of 4 where gfx90a+ has 8 for the unified file, i.e. it shows half of\n"
      "`.amdhsa_next_free_vgpr` rounded up to 8 -- 257 -> 132, 256 -> 128, 512 -> 256 -- and does not decode the accumulation registers.  This table is\n"
      "the compiler's.)\n")
print("| kernel | VGPRs | AGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | waves/SIMD |")
print("|---|---|---|---|---|---|---|---|")
for f in ("nmpc_api.hip", "nmpc_policy.hip", "nmpc_dataset.hip", "nmpc_torque.hip"):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-mllvm", "-amdgpu-mfma-vgpr-form",
                          "-Rpass-analysis=kernel-resource-usage", os.path.join(src, f), "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:.*?\s{2,}([A-Za-z][A-Za-z \[\]/]*): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        print(f"| `{name[:110]}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | {r.get('SGPRs Spill')} | {r.get('VGPRs Spill')} | "
              f"{r.get('ScratchSize [bytes/lane]')} | {r.get('Occupancy [waves/SIMD]')} |")
