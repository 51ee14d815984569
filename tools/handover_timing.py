#!/usr/bin/env python3
"""Where the adapter's hand-over time goes at the e2e timing size (40 Mbp Pg, 1.2 M x 100 bp packed reads): first and
second context of a process (the first pays the runtime's one-time costs: code object load, first allocations)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pgrc_amd import MatchContext, synth
sys.path.insert(0, os.path.join(ROOT, "tools"))
from boundary_rate import pack_acgt
g = synth.pg_params(40_000_000, seed=1); pg = synth.pg_host(g)
rs = synth.reads_params(1_200_000, 100, seed=1); reads = synth.reads_host(g, pg, rs); rows = pack_acgt(reads)
for rep in range(3):
    t0 = time.perf_counter(); ctx = MatchContext(100, 38, 33, 0, "c")
    t1 = time.perf_counter(); ctx.set_pg_ascii(pg)
    t2 = time.perf_counter(); ctx.set_reads_packed(rows, reads.shape[0])
    t3 = time.perf_counter(); ctx.init_results(); ctx.run(True)
    t4 = time.perf_counter(); r = ctx.get_results()
    t5 = time.perf_counter()
    print(f"context {rep}: create {1e3*(t1-t0):.1f} ms, set_pg {1e3*(t2-t1):.1f}, set_reads {1e3*(t3-t2):.1f}, run {1e3*(t4-t3):.1f}, get_results {1e3*(t5-t4):.1f}")
    ctx.close()
