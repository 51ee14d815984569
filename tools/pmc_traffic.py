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
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
