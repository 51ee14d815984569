#!/usr/bin/env python3
"""Row f2 beyond 4 Gi symbols (the 64-bit branches): a 4.4 Gbp synthetic Pg against itself (forward): every sampled match
must be a real, right-maximal exact match of at least 45 symbols.  One-off check (needs ~20 GB of host memory)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from pgrc_amd import CopMEMMatcher, synth  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4_400_000_000
g = synth.pg_params(G, seed=12345)
d_pg = torch.zeros((G + 15) // 16 + 64, dtype=torch.int32, device="cuda")
synth.pg_device(g, d_pg.data_ptr())
torch.cuda.synchronize()
src = bench.unpack_pg_to_ascii(d_pg.cpu().numpy().view(np.uint32)[: (G + 15) // 16])[:G].copy()
del d_pg
m = CopMEMMatcher(src, 45)
mt = m.matchTexts(src, True, False)
ctr = m.counters()
ps, ln, pd = mt[:, 0].astype(np.int64), mt[:, 1].astype(np.int64), mt[:, 2].astype(np.int64)
assert len(mt) > 100000 and ln.min() >= 45 and (pd < ps).all() and (ps + ln).max() <= G
assert int(ps.max()) > (1 << 32), "no match beyond 4 Gi: the 64-bit paths were not exercised"
rng = np.random.default_rng(2)
hi = np.flatnonzero(ps > (1 << 32))
for k in np.concatenate([rng.choice(len(mt), 2000), rng.choice(hi, min(2000, len(hi)))]):
    a, b, n = ps[k], pd[k], ln[k]
    assert np.array_equal(src[a:a + n], src[b:b + n]), k
    assert a + n == G or src[a + n] != src[b + n], k
print({"pg_len": G, "matches": len(mt), "beyond_4Gi": int(len(hi)), "events": ctr["events"], "probes": ctr["probes"], "ok": True})
