#!/usr/bin/env python
"""Where the wall clock of one osmt_render_batch_png call goes (1024 config-2 tiles): validation, upload (host
tables + H2D), kernels, PNG encode, read-back.  Run on a GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from osm_renderer_amd import synth
from osm_renderer_amd.lib import load, check
from osm_renderer_amd.renderer import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = Context(0)
dl = synth.config2(n)
b = dl.as_batch()

def best(f, reps=7):
    f()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return min(ts) * 1e3

print(f"tiles {n}: ops {len(dl.ops)}")
print("osmt_validate_batch        %.3f ms" % best(lambda: check(load().osmt_validate_batch(C.byref(b)))))
sc = [None]
def up():
    if sc[0] is not None:
        sc[0].free()
    sc[0] = ctx.upload(dl)
print("upload (+ free of the last) %.3f ms" % best(up))
out = torch.empty((n, 256, 256, 4), dtype=torch.uint8, device=ctx.device)
print("render (resident)          %.3f ms" % best(lambda: ctx.render(sc[0], out)))
print("png encode (device)        %.3f ms" % best(lambda: ctx.encode_png_device(out)))
pin = ctx.host_alloc((n * load().osmt_png_device_bound(256, 256),))
print("osmt_render_batch_png      %.3f ms" % best(lambda: ctx.render_batch_png(dl, out=pin, as_bytes=False)))
pin2 = ctx.host_alloc((n, 256, 256, 4))
print("osmt_render_batch (pinned) %.3f ms" % best(lambda: ctx.render_batch_host(dl, out=pin2)))
