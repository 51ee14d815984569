#!/usr/bin/env python3
"""HBM traffic of one bench step of modes d / i / e (seedidx.hip + the sort and scan kernels it calls) from the FETCH_SIZE /
WRITE_SIZE passes of tools/pmc_groups.sh: bytes summed over every dispatch of those kernels, divided by the steps of the run
(= dispatches of k_seed_best_store).  read bytes = 2 x FETCH_SIZE (a TCC_EA0_RDREQ is a 128-byte line tallied at 64 B:
profiles/r05_ubench_gather_pmc.txt); WRITE_SIZE is exact.  usage: tools/pmc_seed_traffic.py <pmc dir> <out.json> <workload>"""
import csv, glob, json, os, sys
root, out, wl = sys.argv[1:4]
KERNS = ("k_seed_", "k_rx_", "k_sco_", "k_psc_")
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
per = {}
steps = {}
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        cn = r["Counter_Name"]
        if cn not in tot or not any(k in r["Kernel_Name"] for k in KERNS):
            continue
        v = float(r["Counter_Value"]) * 1024.0
        tot[cn] += v
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dispatches": 0})
        per[name][cn] += v
        if cn == "FETCH_SIZE":
            per[name]["dispatches"] += 1
        if "k_seed_best_store" in r["Kernel_Name"]:
            steps[cn] = steps.get(cn, 0) + 1
n = steps.get("FETCH_SIZE", 0)
assert n and n == steps.get("WRITE_SIZE", 0), steps
res = {"workload": wl, "steps": n,
       "read_bytes_per_step": 2.0 * tot["FETCH_SIZE"] / n, "write_bytes_per_step": tot["WRITE_SIZE"] / n,
       "hbm_bytes_per_step": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / n,
       "unit_note": "read bytes = 2 x FETCH_SIZE: a TCC_EA0_RDREQ is a 128-byte line on gfx950, tallied at 64 B (profiles/r05_ubench_gather_pmc.txt)",
       "kernels": {k: {"read_GB_per_step": 2.0 * v["FETCH_SIZE"] / n / 1e9, "write_GB_per_step": v["WRITE_SIZE"] / n / 1e9,
                       "read_lines_M_per_step": v["FETCH_SIZE"] / 64.0 / n / 1e6, "dispatches_per_step": v["dispatches"] / n}
                   for k, v in sorted(per.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("steps", "read_bytes_per_step", "write_bytes_per_step", "hbm_bytes_per_step")}))
