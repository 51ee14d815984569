#!/usr/bin/env python3
"""Per-dispatch HBM traffic of one kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_groups.sh.
FETCH_SIZE / WRITE_SIZE are in units of 1024 B (rocprofv3).  On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but every such
request is a 128-BYTE line -- for random 16-byte gathers exactly as for coalesced streams: calibrated in round 5 on a known
number of random lines (tools/ubench/gather_pmc.hip, profiles/r05_ubench_gather_pmc.txt: a second load into the other
64-byte half of a lane's line adds no request, one into the adjacent line adds one; TCC_EA0_RDREQ_32B = 0).  So read bytes =
2 x FETCH_SIZE for every kernel (rounds 1-4 doubled it for the index build's streams only and reported the match kernels'
reads at half their bytes).  WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM).
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
    res["dispatches"].append({"fetch_size_bytes": vals["FETCH_SIZE"][i], "read_bytes": 2.0 * vals["FETCH_SIZE"][i], "write_bytes": vals["WRITE_SIZE"][i],
                              "read_lines_128B": vals["FETCH_SIZE"][i] / 64.0,
                              "hbm_bytes": 2.0 * vals["FETCH_SIZE"][i] + vals["WRITE_SIZE"][i]})
res["unit_note"] = "read_bytes = 2 x FETCH_SIZE: a TCC_EA0_RDREQ is a 128-byte line on gfx950, tallied at 64 B (profiles/r05_ubench_gather_pmc.txt)"
# the index build of one strand (idxsweep.hip / idxsort.hip kernels): bytes over all their dispatches of the run, divided by
# the strands built (2 per step; steps = launches of the dual kernel, or of the match kernel / 2); reads = 2 x FETCH_SIZE as above.
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
                               "note": "reads = 2 x FETCH_SIZE (128-byte lines tallied at 64 B); per strand"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
