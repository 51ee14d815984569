#!/usr/bin/env python3
"""Golden fixtures of the export (row f1), generated from the REAL reference compiled in the build container
(oracle/_ref): for a few small cases the reference matches the reads (serial index), sorts the matched reads by
position and exports them -- in pseudogenome order and in original order -- through its own
exportMatchesInPgOrder / exportMatchesInOriginalOrder + SeparatedPseudoGenomeOutputBuilder::build
(matching/ReadsMatchers.cpp:563-675, pseudogenome/persistence/SeparatedPseudoGenomePersistence.cpp:961-1019).
Stored: the reference's match results, its order of the matched reads (ties in its sort's order), and the six
stream files of both exports.  Inputs are re-derived from the generator parameters (tests/export_util.py).

    python tests/golden/make_golden_export.py      # needs /root/reference (run `make -C oracle ref` first)
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import export_util as xu  # noqa: E402
import oracle as orc  # noqa: E402

# name -> (export_case keyword arguments, paired-file rule)
CASES = {
    "export_se": (dict(seed=21, G=120_000, n=5000, n_with_n=150, dups=200), False),
    "export_pe_pairfile": (dict(seed=22, G=120_000, n=5000, n_with_n=150, dups=200, paired=True), True),
    "export_short_list": (dict(seed=23, G=120_000, n=4000, n_with_n=100, dups=100, short_list=True), False),
    "export_L250": (dict(seed=24, G=150_000, n=2500, L=250, n_with_n=60, dups=60, list_gap=110), False),
}


def inputs_digest(case):
    h = hashlib.sha256()
    for k in ("pg", "reads", "list_off", "list_org", "list_rc", "read_org"):
        h.update(np.ascontiguousarray(case[k]).tobytes())
    return h.hexdigest()


def main():
    manifest = {}
    for name, (kw, pair) in CASES.items():
        case = xu.export_case(**kw)
        kmax = case["L"] // 3
        res = orc.ref_match("c", case["pg"], case["reads"], 38, kmax, 0, n_nset=case["n_n"])
        order = xu.position_order(res["pos"])
        out = {"pos": res["pos"], "rc": res["rc"], "mism": res["mism"], "order": order}
        with tempfile.TemporaryDirectory() as d:
            for tag, preserve in (("pg", False), ("org", True)):
                st = xu.ref_export_run(case, os.path.join(d, tag), 0, kmax=kmax, preserve_order=preserve, pair_file_mode=pair,
                                       rev_compl_pair_file=pair)
                for k in xu.STREAMS:
                    out[f"{tag}_{k}"] = np.frombuffer(st[k], dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        manifest[name] = {"kw": kw, "paired_file_rule": pair, "kmax": kmax, "matched": int((res["mism"] != 255).sum()),
                          "inputs_sha256": inputs_digest(case)}
        print(name, manifest[name]["matched"], "matched,", os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
    with open(os.path.join(HERE, "manifest_export.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
