#!/usr/bin/env python
"""BASELINE.json configs[4]: dense-city tiles (5000 polygons + 4000 polylines = 20000 segments per tile, z=17)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from osm_renderer_amd import synth, abi
from osm_renderer_amd.renderer import Context
from oracle import oracle_py as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dl = synth.config5(n)
ctx = Context(0)
sc = ctx.upload(dl)
out = ctx.render(sc)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
ev0.record()
for _ in range(reps):
    ctx.render(sc, out=out)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
print(f"config5: {n} tiles, {ms:.2f} ms/batch, {n / ms * 1e3:.1f} tiles/s, alg bytes/tile {dl.algorithmic_bytes() / n:.0f}")
t = time.perf_counter()
want = O.render_batch(dl.subset([0, n - 1]), threads=2)
print(f"oracle 2 tiles {time.perf_counter() - t:.2f} s; parity:", bool(np.array_equal(out[[0, n - 1]].cpu().numpy(), want)))
