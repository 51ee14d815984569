#!/usr/bin/env python3
"""What the drop-in buys inside the reference's own encoder: runs PgRCEncoder (oracle/_ref) on a synthetic FASTQ once on
the CPU and once with HipReadsMatcher (stage 4) + HipTextMatcher (stage 7) in place, both at the same thread count, and
prints the seconds spent inside mapReadsIntoPg (stage 4) and inside SimplePgMatcher's TextMatcher (stage 7: index
build + the matchTexts calls), measured by the harness around whichever implementation ran, next to the whole encode.  Test infrastructure
(uses oracle/_ref); at -t > 1 the reference is not deterministic, so the archives are not compared here
(tests/test_e2e_dropin.py does that at -t 1).

usage: python tests/e2e_timing.py WORKDIR [--reads N] [--genome G] [--threads T]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_fastq(path, genome, n, L, seed):
    rng = np.random.default_rng(seed)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    starts = rng.integers(0, genome.size - L, size=n)
    rc = rng.random(n) < 0.5
    nsub = rng.choice([0, 0, 0, 0, 0, 0, 1, 1, 2, 4], size=n)
    qual = b"I" * L
    with open(path, "wb") as f:
        chunk = []
        for i in range(n):
            r = genome[starts[i]: starts[i] + L]
            r = comp[r[::-1]] if rc[i] else r.copy()
            for _ in range(nsub[i]):
                p = int(rng.integers(0, L))
                r[p] = (r[p] + int(rng.integers(1, 4))) & 3
            chunk.append(b"@r%d\n" % i + lut[r].tobytes() + b"\n+\n" + qual + b"\n")
            if len(chunk) == 20000:
                f.write(b"".join(chunk))
                chunk = []
        f.write(b"".join(chunk))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workdir")
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--genome", type=int, default=20_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    import pgrc_amd  # noqa: F401
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpgrc_ref.so"))
    lib.pgrc_ref_encode.argtypes = [C.c_char_p] * 3 + [C.c_int] * 3 + [C.c_char, C.c_int, C.c_int, C.c_char, C.c_int]
    os.makedirs(a.workdir, exist_ok=True)
    rng = np.random.default_rng(99)
    genome = rng.integers(0, 4, a.genome, dtype=np.uint8)
    fq = os.path.join(os.path.abspath(a.workdir), "in.fastq")
    t = time.time()
    write_fastq(fq, genome, a.reads, a.read_len, 7)
    print(json.dumps({"fastq_s": round(time.time() - t, 1), "reads": a.reads, "genome": a.genome}), flush=True)
    out = {"reads": a.reads, "read_len": a.read_len, "genome": a.genome, "threads": a.threads, "legs": {}}
    has_div = hasattr(lib, "pgrc_ref_division_seconds")
    if has_div:
        lib.pgrc_ref_division_seconds.restype = C.c_double
    for leg, use_gpu in (("cpu", 0), ("gpu", 7 if has_div else 3)):       # GPU leg: stages 1 (read sets), 4 and 7
        d = os.path.join(os.path.abspath(a.workdir), leg)
        os.makedirs(d, exist_ok=True)
        os.chdir(d)
        t = time.time()
        lib.pgrc_ref_encode(fq.encode(), b"", b"out.pgrc", a.threads, use_gpu, 0, b"\0", 0, 0, b"\0", 0)
        wall = time.time() - t
        s4, s7 = C.c_double(0), C.c_double(0)
        lib.pgrc_ref_stage_seconds(C.byref(s4), C.byref(s7))
        out["legs"][leg] = {"encode_wall_s": round(wall, 2), "archive_bytes": os.path.getsize("out.pgrc"),
                            "mapReadsIntoPg_s (stage 4, incl. export)": round(s4.value, 3),
                            "text_matcher_s (stage 7: index + matchTexts)": round(s7.value, 3)}
        if has_div:
            out["legs"][leg]["read_sets_s (stage 1: FASTQ iterator + division + packing)"] = round(lib.pgrc_ref_division_seconds(), 3)
        print(json.dumps({leg: out["legs"][leg]}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
