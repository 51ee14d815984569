#!/usr/bin/env python3
"""Stage 4 of the reference's encoder at the size of BASELINE config C3, inside the compiled reference (oracle/_ref):
PgTools::mapReadsIntoPg's matcher on the encoder's LQ + N sum set (the reference's own PackedConstantLengthReadsSets, filled
from packed rows) against a 1.875 Gbp pseudogenome -- HipReadsMatcher (hand-over, run and result fetch overlapped:
matchStreamed; PGRC_NO_STREAM=1: in turn) and the reference's CopMEMReadsApproxMatcher on the host cores -- plus the adapter's
Pg-order export up to the builder's own stream compression (position sort on the host, streams from the device).
usage (GPU box, needs oracle/_ref): python tools/stage4_c3.py [--reads N --pg G --cpu-reads M] > profiles/rNN_stage4_c3.json"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--pg", type=int, default=1_875_000_000)
    ap.add_argument("--L", type=int, default=150)
    ap.add_argument("--nfrac", type=float, default=0.02)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--cpu-reads", type=int, default=5_000_000, help="reads of the CPU leg (0 = skip, -1 = all): the reference does ~1 M reads/s")
    ap.add_argument("--list-frac", type=float, default=0.5, help="entries of the reads list already on the Pg, as a fraction of the reads")
    a = ap.parse_args()
    import numpy as np
    import pgrc_amd  # noqa: F401  (loads libpgrc_match.so and the HIP runtime first)
    import boundary_c3
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpgrc_ref.so"))
    lib.pgrc_ref_stage4.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint8,
                                    C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pgrc_ref_phase_seconds.restype = C.c_double
    lib.pgrc_ref_phase_seconds.argtypes = [C.c_char_p]
    lib.pgrc_ref_streamed_runs.restype = C.c_uint64
    n, L, G = a.reads, a.L, a.pg
    t0 = time.perf_counter()
    pg, lq_rows, n_rows, n_lq, n_n = boundary_c3.make_host_inputs(n, L, G, a.nfrac)
    out = {"reads": n, "L": L, "pg": G, "n_set_reads": n_n, "prep_s": time.perf_counter() - t0, "threads": a.threads}
    phases = ["streamed hand-over + matching + results", "hand-over of the pseudogenome and the reads", "device run (both strands)", "result fetch",
              "export: position sort", "export: streams from the device",
              "  host: result vectors (the wait for them)", "  library: pseudogenome up, index builds started", "  library: reads up, matching, results down (+ strand flags)",
              "  sort: (position, read) pairs", "  sort: the reference's sort", "  sort: order array",
              "  streams: original indexes", "  streams: library call", "  streams: append to the builder"]

    def run(use_adapter, lq, nl, nr, nn, with_export, list_count):
        secs = (C.c_double * 4)()
        matched, fnv = C.c_uint64(), C.c_uint64()
        before = {p: lib.pgrc_ref_phase_seconds(p.encode()) for p in phases}
        t = time.perf_counter()
        rc = lib.pgrc_ref_stage4(use_adapter, pg.ctypes.data, G, lq.ctypes.data, nl, nr.ctypes.data if nn else None, nn, L, 38, L // 50,
                                 a.threads, a.threads, with_export, list_count, secs, C.byref(matched), C.byref(fnv))
        assert rc == 0
        r = {"wall_s": time.perf_counter() - t, "sets_built_s": secs[0], "matcher_s": secs[1], "export_streams_s": secs[2], "matched": matched.value, "fnv": hex(fnv.value)}
        r["adapter_phases_s"] = {p: lib.pgrc_ref_phase_seconds(p.encode()) - before[p] for p in phases if lib.pgrc_ref_phase_seconds(p.encode()) - before[p] > 0}
        return r

    list_count = int(n * a.list_frac)
    out["hip_streamed"] = [run(1, lq_rows, n_lq, n_rows, n_n, 1, list_count) for _ in range(2)]
    out["streamed_runs"] = int(lib.pgrc_ref_streamed_runs())
    os.environ["PGRC_NO_STREAM"] = "1"
    out["hip_in_turn"] = [run(1, lq_rows, n_lq, n_rows, n_n, 0, 0)]
    del os.environ["PGRC_NO_STREAM"]
    for k in ("hip_streamed", "hip_in_turn"):
        for r in out[k]:
            r["reads_per_s_matcher"] = n / r["matcher_s"]
    out["results_equal"] = len({r["fnv"] for k in ("hip_streamed", "hip_in_turn") for r in out[k]}) == 1
    if a.cpu_reads:
        # the reference on the host cores, on the FIRST reads of each set (whole text, so its index builds are the real ones)
        m = n if a.cpu_reads < 0 else min(a.cpu_reads, n)
        ml, mn = (n_lq, n_n) if m == n else (m - int(m * a.nfrac), int(m * a.nfrac))
        cpu = run(0, lq_rows, ml, n_rows, mn, 0, 0)
        cpu["reads"] = ml + mn
        if m != n:
            hip = run(1, lq_rows, ml, n_rows, mn, 0, 0)          # the same sample through the adapter: results must agree
            cpu["hip_same_sample_fnv"] = hip["fnv"]
            cpu["equal_to_hip_on_the_sample"] = hip["fnv"] == cpu["fnv"]
        else:
            cpu["equal_to_hip"] = cpu["fnv"] == out["hip_streamed"][0]["fnv"]
        cpu["note"] = ("reference as shipped at -t %d: its multithreaded index build is racy (DESIGN.md section 6.2), so a few reads in repeats may differ from the serial-index results" % a.threads)
        out["reference_cpu"] = cpu
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
