#!/usr/bin/env python3
"""Per-dispatch HBM traffic of one kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_groups.sh.
FETCH_SIZE / WRITE_SIZE are in KiB-sized units of 1024 B (rocprofv3); the match kernel's requests are 64-B random
sectors, so the guide's x2 correction for 128-B streaming requests does not apply (TCC_EA0_RDREQ x 64 B agrees).
usage: tools/pmc_traffic.py <pmc dir> <kernel substring> <out.json> [launches per bench step] [schedule]
schedule as bench.py names it: "two passes" (2 launches: forward, RC), "screened" (3: screen, forward, RC), "dual" (3: the
dual kernel, then the two ordinary passes over the reads it left undecided)"""
import csv, glob, json, os, sys
root, kern, out = sys.argv[1:4]
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 2
schedule = sys.argv[5] if len(sys.argv) > 5 else {2: "two passes", 3: "screened"}[per_step]
vals = {}
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]) * 1024.0)
res = {"kernel": kern, "per_step": per_step, "schedule": schedule, "dispatches": []}
for i in range(min(len(vals.get("FETCH_SIZE", [])), len(vals.get("WRITE_SIZE", [])))):
    res["dispatches"].append({"fetch_bytes": vals["FETCH_SIZE"][i], "write_bytes": vals["WRITE_SIZE"][i],
                              "hbm_bytes": vals["FETCH_SIZE"][i] + vals["WRITE_SIZE"][i]})
# the index build of one strand (idxsweep.hip / idxsort.hip kernels): bytes over all their dispatches of the run, divided by
# the strands built (2 per step; steps = launches of the dual kernel, or of the match kernel / 2).  Their reads are wide
# coalesced streams, which gfx950's FETCH_SIZE reports at half their bytes (MI355X_MICROARCH.md, HBM): doubled here.
idx = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
steps = 0
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    names = {r["Counter_Name"] for r in rows}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        if cn in names:
            idx[cn] += sum(float(r["Counter_Value"]) * 1024.0 for r in rows if r["Counter_Name"] == cn and
                           any(k in r["Kernel_Name"] for k in ("k_os_", "k_ps_", "k_psc_")))
            if cn == "FETCH_SIZE":
                nm = sum(1 for r in rows if r["Counter_Name"] == cn and kern in r["Kernel_Name"])
                steps = nm // per_step if per_step else 0
if steps:
    res["index_per_strand"] = {"read_bytes": 2.0 * idx["FETCH_SIZE"] / (2 * steps), "write_bytes": idx["WRITE_SIZE"] / (2 * steps),
                               "hbm_bytes": (2.0 * idx["FETCH_SIZE"] + idx["WRITE_SIZE"]) / (2 * steps), "steps": steps,
                               "note": "FETCH_SIZE of the build's streaming reads doubled (gfx950 reports half); per strand"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
