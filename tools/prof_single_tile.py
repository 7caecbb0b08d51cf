#!/usr/bin/env python
"""One-tile requests (the reference's server shape: one tile per request per worker, src/http_server.rs:134-181) through
osmt_render_batch_rgb, back to back on one thread: wall clock per request, for `rocprofv3 --kernel-trace --stats` to say how
much of it the kernels are (run on a GPU box):

    rocprofv3 --kernel-trace --stats -d /tmp/st -o st -- python tools/prof_single_tile.py [requests]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = Context(0)
big = synth.config2(8)
pool = [big.subset([i]) for i in range(8)]
out = ctx.host_alloc((1, 256, 256, 3))
for dl in pool:
    ctx.render_batch_rgb(dl, out=out)
ts = []
for i in range(n):
    t = time.perf_counter()
    ctx.render_batch_rgb(pool[i % len(pool)], out=out)
    ts.append(time.perf_counter() - t)
ts = np.array(ts) * 1e6
print("one-tile requests: %d, p50 %.1f us, p99 %.1f us, mean %.1f us" % (n, np.percentile(ts, 50), np.percentile(ts, 99), ts.mean()))
