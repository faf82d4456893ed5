#!/usr/bin/env python3
"""PMC-measured HBM bytes per launch of a kernel -> profiles/traffic.json (what bench.py reports as roofline.traffic).
   python tools/traffic_from_profile.py gpurun_out/prof_<tag> <key> <kernel-substring>
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (64 B units counted as 32), both counters are in KB.
The record carries the fingerprint of the kernel sources it was measured on; bench.py withholds it for any other build."""
import csv, glob, hashlib, json, os, sys

src, key, kname = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(root, "iterative_learning_nmpc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".inc")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"] and r["Counter_Name"] in vals:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
assert vals["FETCH_SIZE"] and vals["WRITE_SIZE"], "counters not found"
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024 * 2
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
path = os.path.join(root, "profiles", "traffic.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
rec = {k: v for k, v in rec.items() if isinstance(v, dict)}      # drop the round-1 flat entries
rec[key] = {"bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "kernel": kname,
            "source": os.path.basename(src.rstrip("/")), "kernel_sources_sha": kernel_sources_sha(),
            "dispatches": len(vals["FETCH_SIZE"])}
json.dump(rec, open(path, "w"), indent=1)
print(key, rec[key])
