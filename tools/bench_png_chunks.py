#!/usr/bin/env python
"""osmt_render_batch_png wall clock for different chunk counts (OSMT_PNG_CHUNKS, diagnostic) — run on a GPU box."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import torch
from osm_renderer_amd import synth
from osm_renderer_amd.lib import load
from osm_renderer_amd.renderer import Context
ctx = Context(0)
dl = synth.config2(1024)
pin = ctx.host_alloc((1024 * load().osmt_png_device_bound(256, 256),))
ts = []
for i in range(9):
    torch.cuda.synchronize(); t = time.perf_counter(); ctx.render_batch_png(dl, out=pin, as_bytes=False); ts.append(time.perf_counter() - t)
print("%%.3f ms best, %%.3f median" %% (min(ts[2:]) * 1e3, sorted(ts[2:])[len(ts[2:]) // 2] * 1e3))
'''
for k in sys.argv[1:] or ["1", "2", "4", "8"]:
    env = dict(os.environ, OSMT_PNG_CHUNKS=k)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True)
    print("chunks", k, r.stdout.strip() or r.stderr[-300:])
