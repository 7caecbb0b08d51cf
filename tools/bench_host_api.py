#!/usr/bin/env python
"""Latency / throughput of the host-buffer entry point osmt_render_batch (upload + 3 kernels + readback),
the call a per-request server loop would make."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context
ctx = Context(0)
for n in (1, 16, 256, 1024):
    dl = synth.config2(n)
    ctx.render_batch_host(dl)
    reps = max(3, 200 // n)
    t = time.perf_counter()
    for _ in range(reps):
        ctx.render_batch_host(dl)
    dt = (time.perf_counter() - t) / reps
    print(f"osmt_render_batch n={n:5d}: {dt*1e3:8.3f} ms/call  {n/dt:10.0f} tiles/s (PCIe-inclusive, pageable host buffers)")
    pin = ctx.host_alloc((n, dl.dim, dl.dim, 4))
    ctx.render_batch_host(dl, out=pin)
    t = time.perf_counter()
    for _ in range(reps):
        ctx.render_batch_host(dl, out=pin)
    dt = (time.perf_counter() - t) / reps
    print(f"osmt_render_batch n={n:5d}: {dt*1e3:8.3f} ms/call  {n/dt:10.0f} tiles/s (PCIe-inclusive, pinned output, chunks overlapped when n >= 256)")
    ctx.host_free(pin)
