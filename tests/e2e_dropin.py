"""End-to-end drop-in check, run as a child process by tests/test_gpu_e2e.py (the reference's encoder calls exit() on
errors, so it must not run inside pytest).

Drives the reference's WHOLE encoder (PgRCEncoder::executePgRCChain, pgrc/pgrc-encoder.cpp) compiled into
oracle/_ref/libpgrc_ref.so twice on the same synthetic FASTQ: once on the CPU at -t 1, once with
PgTools::mapReadsIntoPg (matching/ReadsMatchers.cpp:693-796) replaced by the patched version of INTEGRATION.md
section 1, i.e. with integration/HipReadsMatcher -> libpgrc_match.so in the matcher seam (stage 4), and with
SimplePgMatcher holding integration/HipTextMatcher instead of its CopMEMMatcher (stage 7, Pg-vs-Pg matching).  Then decodes the second
archive with the reference's decoder.  Prints one JSON line: archive sizes/digests, whether they are byte-identical,
how often the GPU path ran, and whether the decoded reads equal the input.

usage: python tests/e2e_dropin.py WORKDIR CASE        (CASE: one of CASES below)
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    #            paired, preserve_order, mode, seed, M, pre_mode, pre_seed
    "se":        (False, False, None, 0, 0, None, 0),
    "se_order":  (False, True, None, 0, 0, None, 0),
    "pe":        (True, False, None, 0, 0, None, 0),
    "pe_order":  (True, True, None, 0, 0, None, 0),
    "se_pre":    (False, False, "c", 32, 4, "c", 64),   # two-phase flow (ReadsMatchers.cpp:749-779)
    "se_modeD":  (False, False, "d", 36, 3, None, 0),   # read-side seed index instead of copMEM
    "se_modeI":  (False, False, "I", 34, 5, None, 0),   # interleaved + "shortcut" (upper case) rule
    "se_exact":  (False, False, "d", 100, 3, None, 0),  # seed == read length: DefaultReadsExactMatcher's seat (kind 'e')
}


def write_fastq(path, reads, quality_seed=None):
    """quality_seed None: every base 'I' (Phred 40); else varied quality rows -- good reads, reads with a bad tail, reads
    noisy all over -- so that PgRC's quality-based division (-q) has something to cut"""
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    L = reads.shape[1]
    qual = b"I" * L
    rng = np.random.default_rng(quality_seed) if quality_seed is not None else None
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            if rng is not None:
                q = rng.integers(30, 42, size=L)
                u = rng.random()
                if u < 0.15:
                    k = int(rng.integers(1, L))
                    q[k:] = rng.integers(2, 12, size=L - k)
                elif u < 0.3:
                    q = rng.integers(2, 42, size=L)
                qual = (q + 33).astype(np.uint8).tobytes()
            f.write(b"@r%d\n" % i + lut[r].tobytes() + b"\n+\n" + qual + b"\n")


def make_reads(seed, G, L, n, paired):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, G, dtype=np.uint8)
    # a few repeats so that buckets fill up and ties exist
    for _ in range(6):
        s, d = rng.integers(0, G - 3000, 2)
        g[d:d + 3000] = g[s:s + 3000]
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    reads = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        if paired and (i & 1):
            s = min(G - L, s + int(rng.integers(150, 400)))
            rc = not rc
        else:
            s = int(rng.integers(0, G - L))
            rc = rng.random() < 0.5
        rd = g[s:s + L].copy()
        if rc:
            rd = comp[rd[::-1]]
        for _ in range(int(rng.choice([0, 0, 0, 0, 1, 1, 2, 3, 5, 9]))):
            p = int(rng.integers(0, L))
            rd[p] = (rd[p] + int(rng.integers(1, 4))) & 3
        if i % 400 == 7:
            rd[int(rng.integers(0, L))] = 4
        reads[i] = rd
    return reads


def read_seqs(path):
    with open(path, "rb") as f:
        return [ln.strip() for ln in f if ln.strip()]


def main():
    work, case = sys.argv[1], sys.argv[2]
    paired, order, mode, seed, mcpm, pre_mode, pre_seed = CASES[case]
    import pgrc_amd  # noqa: F401  (loads libpgrc_match.so and the one HIP runtime first)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpgrc_ref.so"))
    lib.pgrc_ref_encode.argtypes = [C.c_char_p] * 3 + [C.c_int] * 3 + [C.c_char, C.c_int, C.c_int, C.c_char, C.c_int]
    lib.pgrc_ref_decode.argtypes = [C.c_char_p, C.c_int]
    lib.pgrc_ref_bulk_updates.restype = C.c_uint64
    lib.pgrc_ref_text_match_calls.restype = C.c_uint64
    lib.pgrc_ref_device_exports.restype = C.c_uint64
    lib.pgrc_ref_dual_runs.restype = C.c_uint64
    lib.pgrc_ref_streamed_runs.restype = C.c_uint64
    lib.pgrc_ref_division_calls.restype = C.c_int

    # default: small enough for the test suite; PGRC_E2E_READS / PGRC_E2E_GENOME scale it up for a one-off check
    n = int(os.environ.get("PGRC_E2E_READS", "60000"))
    G, L = int(os.environ.get("PGRC_E2E_GENOME", str(5 * n))), int(os.environ.get("PGRC_E2E_READ_LEN", "100"))
    reads = make_reads(11, G, L, n, paired)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    out = {"case": case, "reads": n, "read_len": L}
    digests = {}
    # PGRC_E2E_CPU_ONLY=1: both legs on the CPU (no GPU needed) -- checks the harness itself: the reference encoder is
    # deterministic at -t 1 and its archive decodes to the input
    # GPU leg: bit 0 = stage 4 (reads -> Pg, HipReadsMatcher), bit 1 = stage 7 (Pg -> Pg, HipTextMatcher), bit 2 = stage 1 (the read
    # sets: division + packing, HipDividedReadsSets)
    gpu_leg = 0 if os.environ.get("PGRC_E2E_CPU_ONLY") == "1" else int(os.environ.get("PGRC_E2E_GPU_STAGES", "3"))
    exports_seen = 0
    for leg, use_gpu in (("cpu", 0), ("gpu", gpu_leg)):
        d = os.path.join(work, case, leg)
        os.makedirs(d, exist_ok=True)
        os.chdir(d)       # same relative archive name in both legs: the archive embeds a name derived from it
        qseed = 5 if os.environ.get("PGRC_REF_Q_PROMILS") else None    # (quality division asked for: varied quality rows)
        if paired:
            write_fastq("in_1.fastq", reads[0::2], qseed)
            write_fastq("in_2.fastq", reads[1::2], None if qseed is None else qseed + 1)
            src, pair = b"in_1.fastq", b"in_2.fastq"
        else:
            write_fastq("in.fastq", reads, qseed)
            src, pair = b"in.fastq", b""
        calls = lib.pgrc_ref_encode(src, pair, b"out.pgrc", int(os.environ.get("PGRC_E2E_THREADS", "1")), use_gpu, 1 if order else 0,
                                    (mode or "\0").encode(), seed, mcpm, (pre_mode or "\0").encode(), pre_seed)
        blob = open("out.pgrc", "rb").read()
        digests[leg] = hashlib.sha256(blob).hexdigest()
        out[leg + "_bytes"] = len(blob)
        out[leg + "_gpu_calls"] = calls
        out[leg + "_bulk_updates"] = int(lib.pgrc_ref_bulk_updates())   # entries served by the device extraction
        out[leg + "_text_match_calls"] = int(lib.pgrc_ref_text_match_calls())   # matchTexts calls served by HipTextMatcher
        out[leg + "_device_exports"] = int(lib.pgrc_ref_device_exports()) - exports_seen   # exports whose streams came from the device
        exports_seen = int(lib.pgrc_ref_device_exports())
    out["gpu_division_calls"] = int(lib.pgrc_ref_division_calls())   # read-set divisions that ran on the device (bit 2 of the GPU stages)
    out["gpu_dual_runs"] = int(lib.pgrc_ref_dual_runs())     # device runs that took the dual kernel (GPU leg only: the CPU leg has none)
    out["gpu_streamed_runs"] = int(lib.pgrc_ref_streamed_runs())   # ... that overlapped hand-over, matching and result fetch
    out["identical"] = digests["cpu"] == digests["gpu"]
    out["sha256"] = digests

    # decode the GPU-leg archive with the reference's decoder and compare with the input
    lib.pgrc_ref_decode(b"out.pgrc", 1)
    if paired:
        got1, got2 = read_seqs("out.pgrc_out_1"), read_seqs("out.pgrc_out_2")
        want = [(lut[a].tobytes(), lut[b].tobytes()) for a, b in zip(reads[0::2], reads[1::2])]
        got = list(zip(got1, got2))
    else:
        got = read_seqs("out.pgrc_out")
        want = [lut[r].tobytes() for r in reads]
    out["decoded"] = len(got)
    out["roundtrip"] = (got == want) if order else (sorted(got) == sorted(want))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
