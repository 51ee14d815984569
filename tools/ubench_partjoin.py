#!/usr/bin/env python3
"""partitioned probing against random probing of the bucket heads (tools/ubench/partjoin.hip; round 4: no-go,
profiles/r04_ubench_partjoin.txt).  The kernel is not part of the product library: `make -C pgrc_amd/csrc partjoin` links the product's
objects with it into tools/ubench/libpgrc_partjoin.so, which this script loads.
1 G probe records (round 1 of C3: 5 seeds x 2 strands x 100 M reads) against a table of 2^29 16-byte heads.
usage: python tools/ubench_partjoin.py [n_probes] [hash_bits] > profiles/rNN_ubench_partjoin.txt"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
lib = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libpgrc_partjoin.so"))
lib.pgrc_match_ubench_partjoin.restype = C.c_int
lib.pgrc_match_ubench_partjoin.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_float * 4), C.POINTER(C.c_uint64 * 2)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
hb = int(sys.argv[2]) if len(sys.argv) > 2 else 29
for rep in range(2):
    ms, sums = (C.c_float * 4)(), (C.c_uint64 * 2)()
    e = lib.pgrc_match_ubench_partjoin(n, hb, C.byref(ms), C.byref(sums))
    assert e == 0, e
    print(f"run {rep}: {n} probes, table of 2^{hb} heads: records {ms[0]:.2f} ms | random gather {ms[1]:.2f} ms ({n / ms[1] / 1e6:.1f} G/s) | "
          f"partition by the top {hb - 13} bucket bits (two stable scatter passes, radix.hip) {ms[2]:.2f} ms | join (8192 heads in LDS per partition) {ms[3]:.2f} ms | "
          f"partition + join {ms[2] + ms[3]:.2f} ms | checksums {'agree' if sums[0] == sums[1] else 'DIFFER'}", flush=True)
