#!/usr/bin/env python
"""osmt_render_batch_png: the one-piece call against the begin / end pair with two jobs in flight, one caller thread (GPU).

    python tools/bench_png_begin_end.py [tiles=1024] [batches=16]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osm_renderer_amd import synth  # noqa: E402
from osm_renderer_amd.renderer import Context  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_pipe = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = Context(0)
dl = synth.config2(tiles)
pb = [ctx.host_alloc((dl.n_jobs * 96 * 1024,)) for _ in range(2)]
for _ in range(3):
    ctx.render_batch_png(dl, out=pb[0]) if "out" in ctx.render_batch_png.__code__.co_varnames else ctx.png_end(ctx.png_begin(dl), pb[0])
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ctx.png_end(ctx.png_begin(dl), pb[0])
    best = min(best, time.perf_counter() - t0)
print(f"one piece (begin + end back to back): {best * 1e3:.3f} ms per batch = {tiles / best:,.0f} tiles/s")
for rep in range(3):
    t0 = time.perf_counter()
    prev = ctx.png_begin(dl)
    for k in range(1, n_pipe):
        cur = ctx.png_begin(dl)
        ctx.png_end(prev, pb[(k - 1) & 1])
        prev = cur
    ctx.png_end(prev, pb[(n_pipe - 1) & 1])
    dt = (time.perf_counter() - t0) / n_pipe
    print(f"begin(k+1) before end(k), {n_pipe} batches: {dt * 1e3:.3f} ms per batch = {tiles / dt:,.0f} tiles/s")
