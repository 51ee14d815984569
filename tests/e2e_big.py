#!/usr/bin/env python3
"""The reference's whole encoder at a size where stage 4 takes the dual kernel by default and matters for the wall clock
(BASELINE configs[4], second half: "full pgrc-encoder end-to-end archive size + wall-clock"): a synthetic FASTQ of
N x 150 bp reads (10x coverage of a random genome with a few repeats; 50 % reverse strand; 0-4 substitutions; every
400th read holds an N), written with numpy (no per-read Python loop), then

  cpu leg : PgRCEncoder untouched (oracle/_ref) at --threads
  gpu leg : the same encoder with HipDividedReadsSets (stage 1), HipReadsMatcher (stage 4) and HipTextMatcher (stage 7)

per leg: wall clock, seconds inside stages 1 / 4 / 7, archive bytes; the GPU leg's archive is decoded with the reference's
decoder and compared with the input as a multiset of reads (64-bit row hashes, sorted).  --identity: both legs at -t 1 and
the two archives compared byte for byte (identity is a -t 1 property: at -t > 1 the reference's own archive differs from run
to run -- racy index build, parallel Pg generator -- SURVEY 8c).  Test infrastructure (uses oracle/_ref).

usage: python tests/e2e_big.py WORKDIR [--reads N] [--threads T] [--identity]"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def row_hashes(rows):
    """64-bit hash per row of a (n, L) uint8 matrix (polynomial over fixed odd weights, then a splitmix finaliser)"""
    L = rows.shape[1]
    w = (np.arange(1, L + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    out = np.empty(rows.shape[0], dtype=np.uint64)
    step = 1 << 20
    with np.errstate(over="ignore"):
        for lo in range(0, rows.shape[0], step):
            h = (rows[lo:lo + step].astype(np.uint64) * w[None, :]).sum(axis=1, dtype=np.uint64)
            h ^= h >> np.uint64(30); h *= np.uint64(0xBF58476D1CE4E5B9)
            h ^= h >> np.uint64(27); h *= np.uint64(0x94D049BB133111EB)
            h ^= h >> np.uint64(31)
            out[lo:lo + step] = h
    return out


def write_fastq(path, genome, n, L, seed):
    """-> sorted row hashes of the reads written (ASCII rows)"""
    rng = np.random.default_rng(seed)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    hashes = np.empty(n, dtype=np.uint64)
    hdr_w = 10                                             # "@r%09d"-style fixed-width names: constant record size
    rec = 1 + hdr_w + 1 + L + 1 + 2 + L + 1
    step = 1 << 20
    with open(path, "wb") as f:
        for lo in range(0, n, step):
            m = min(step, n - lo)
            starts = rng.integers(0, genome.size - L, size=m)
            r = genome[starts[:, None] + np.arange(L)[None, :]]
            rc = rng.random(m) < 0.5
            r[rc] = comp[r[rc][:, ::-1]]
            nsub = rng.choice(np.array([0, 0, 0, 0, 0, 0, 1, 1, 2, 4]), size=m)
            for k in range(4):
                sel = np.flatnonzero(nsub > k)
                p = rng.integers(0, L, size=sel.size)
                r[sel, p] = (r[sel, p] + rng.integers(1, 4, size=sel.size).astype(np.uint8)) & 3
            idx = np.arange(lo, lo + m)
            withn = np.flatnonzero(idx % 400 == 7)
            r[withn, rng.integers(0, L, size=withn.size)] = 4
            rows = lut[r]
            hashes[lo:lo + m] = row_hashes(rows)
            buf = np.empty((m, rec), dtype=np.uint8)
            buf[:, 0] = ord("@")
            digits = (idx[:, None] // (10 ** np.arange(hdr_w - 1, -1, -1))[None, :]) % 10
            buf[:, 1:1 + hdr_w] = digits.astype(np.uint8) + ord("0")
            at = 1 + hdr_w
            buf[:, at] = ord("\n"); at += 1
            buf[:, at:at + L] = rows; at += L
            buf[:, at] = ord("\n"); buf[:, at + 1] = ord("+"); buf[:, at + 2] = ord("\n"); at += 3
            buf[:, at:at + L] = ord("I"); at += L
            buf[:, at] = ord("\n")
            f.write(buf.tobytes())
    hashes.sort()
    return hashes


def decoded_hashes(path, L):
    raw = np.fromfile(path, dtype=np.uint8)
    assert raw.size % (L + 1) == 0, "decoded file is not made of %d-symbol lines" % L
    rows = raw.reshape(-1, L + 1)
    assert (rows[:, L] == ord("\n")).all()
    h = row_hashes(rows[:, :L])
    h.sort()
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workdir")
    ap.add_argument("--reads", type=int, default=25_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--coverage", type=float, default=10.0)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--identity", action="store_true", help="both legs at -t 1, archives compared byte for byte")
    ap.add_argument("--cpu-only", action="store_true", help="the harness itself, no GPU: only the untouched encoder, decoded and compared")
    a = ap.parse_args()
    import pgrc_amd  # noqa: F401
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpgrc_ref.so"))
    lib.pgrc_ref_encode.argtypes = [C.c_char_p] * 3 + [C.c_int] * 3 + [C.c_char, C.c_int, C.c_int, C.c_char, C.c_int]
    lib.pgrc_ref_decode.argtypes = [C.c_char_p, C.c_int]
    lib.pgrc_ref_division_seconds.restype = C.c_double
    lib.pgrc_ref_dual_runs.restype = C.c_uint64
    lib.pgrc_ref_streamed_runs.restype = C.c_uint64
    lib.pgrc_ref_device_exports.restype = C.c_uint64
    threads = 1 if a.identity else a.threads
    L, n = a.read_len, a.reads
    G = int(n * L / a.coverage)
    os.makedirs(a.workdir, exist_ok=True)
    rng = np.random.default_rng(99)
    genome = rng.integers(0, 4, G, dtype=np.uint8)
    for _ in range(8):                                     # a few repeats: crowded buckets, reads matching at several places
        s, d = rng.integers(0, G - 3000, 2)
        genome[d:d + 3000] = genome[s:s + 3000]
    fq = os.path.join(os.path.abspath(a.workdir), "in.fastq")
    t = time.time()
    want = write_fastq(fq, genome, n, L, 7)
    del genome
    out = {"reads": n, "read_len": L, "genome": G, "threads": threads, "fastq_bytes": os.path.getsize(fq),
           "fastq_s": round(time.time() - t, 1), "legs": {}}
    print(json.dumps({"fastq_s": out["fastq_s"]}), flush=True)
    digests = {}
    for leg, use_gpu in ((("cpu", 0),) if a.cpu_only else (("cpu", 0), ("gpu", 7))):          # GPU leg: stages 1 (read sets), 4 and 7
        d = os.path.join(os.path.abspath(a.workdir), leg)
        os.makedirs(d, exist_ok=True)
        os.chdir(d)
        t = time.time()
        lib.pgrc_ref_encode(fq.encode(), b"", b"out.pgrc", threads, use_gpu, 0, b"\0", 0, 0, b"\0", 0)
        wall = time.time() - t
        s4, s7 = C.c_double(0), C.c_double(0)
        lib.pgrc_ref_stage_seconds(C.byref(s4), C.byref(s7))
        h = hashlib.sha256()
        with open("out.pgrc", "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        digests[leg] = h.hexdigest()
        out["legs"][leg] = {"encode_wall_s": round(wall, 2), "archive_bytes": os.path.getsize("out.pgrc"),
                            "stage1_read_sets_s": round(lib.pgrc_ref_division_seconds(), 3),
                            "stage4_mapReadsIntoPg_s (incl. export and the reference's stream compression)": round(s4.value, 3),
                            "stage7_text_matcher_s": round(s7.value, 3), "sha256": digests[leg]}
        print(json.dumps({leg: out["legs"][leg]}), flush=True)
    out["gpu_dual_runs"] = int(lib.pgrc_ref_dual_runs())
    out["gpu_streamed_runs"] = int(lib.pgrc_ref_streamed_runs())
    out["gpu_device_exports"] = int(lib.pgrc_ref_device_exports())
    out["archives_identical"] = digests["cpu"] == digests.get("gpu")
    out["identity_note"] = ("byte identity of the two archives is a -t 1 property: at -t > 1 the reference's own archive differs from "
                            "run to run (racy copMEM index build, parallel Pg generator, SURVEY 8c), and HipReadsMatcher then orders "
                            "the matched reads on the device (ties by read index)")
    t = time.time()
    lib.pgrc_ref_decode(b"out.pgrc", threads)
    got = decoded_hashes("out.pgrc_out", L)
    out["decode_s"] = round(time.time() - t, 1)
    out["decoded_reads"] = int(got.size)
    out["roundtrip"] = bool(got.size == want.size and np.array_equal(got, want))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
