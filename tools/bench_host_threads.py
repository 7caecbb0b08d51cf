#!/usr/bin/env python
"""The reference's server shape (http_server.rs:50-83: N worker threads, one tile per request): T host threads
calling osmt_render_batch concurrently on ONE context, n tiles per call."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context

ctx = Context(0)
for n in (1, 16):
    dls = [synth.make_tiles(synth.config_tiles(n * 64)[i * n:(i + 1) * n]) for i in range(64)]
    ctx.render_batch_host(dls[0])
    for T in (1, 4, 16, 64):
        calls = max(4, 400 // T) if n == 1 else max(2, 100 // T)
        def work(t):
            for c in range(calls):
                ctx.render_batch_host(dls[(t + c) % 64])
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.perf_counter() - t0
        print(f"n={n:3d} tiles/call, {T:3d} threads: {T * calls * n / dt:9.0f} tiles/s  ({dt / calls * 1e3:.3f} ms per call per thread)")
