#!/usr/bin/env python
"""Real-data stress that the named configs do not contain: a few polygons / polylines with thousands of
vertices, mostly outside the tile (landuse multipolygons, long ways)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from osm_renderer_amd import display_list
from osm_renderer_amd.display_list import TileBuilder
from osm_renderer_amd.renderer import Context
from oracle import oracle_py as O
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ntiles = 64
rnd = np.random.default_rng(3)
tiles = []
for t in range(ntiles):
    tb = TileBuilder()
    for k in range(5):
        c = rnd.integers(-500, 756, size=2); R = rnd.integers(300, 3000)
        ang = np.sort(rnd.uniform(0, 2 * np.pi, nv)); r = R * (1 + 0.1 * np.sin(7 * ang + rnd.uniform(0, 6)) + 0.0005 * rnd.standard_normal(nv))  # smooth outline, small jitter
        pts = np.stack([c[0] + r * np.cos(ang), c[1] + r * np.sin(ang)], 1).round().astype(int).tolist()
        pts.append(pts[0])
        tb.fill(pts, tuple(rnd.integers(0, 256, size=3)), 0.6)
        tb.stroke(pts, 2.0, tuple(rnd.integers(0, 256, size=3)), 0.8, dashes=[6, 3])
    tiles.append(tb.build())
dl = display_list.concat(tiles)
ctx = Context(0); sc = ctx.upload(dl); out = ctx.render(sc); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.render(sc, out=out); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"{ntiles} tiles x 5 polygons+outlines of {nv} vertices: {ms:.2f} ms/batch = {ntiles/ms*1e3:.0f} tiles/s")
t = time.perf_counter(); want = O.render_batch(dl.subset([0, 1]), threads=2); print(f"oracle 2 tiles: {time.perf_counter()-t:.2f} s; parity", bool(np.array_equal(out[:2].cpu().numpy(), want)))
