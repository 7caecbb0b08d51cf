#!/usr/bin/env python
"""Composite-pass bandwidth vs tile dimension / layer count (stride aliasing check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context
ctx = Context(0)
canvas = [0.9, 0.9, 0.9, 1.0]
for dim, L, n in [(512, 8, 64), (496, 8, 68), (504, 8, 66), (512, 7, 72), (512, 8, 256), (256, 8, 256), (1024, 8, 16)]:
    planes = synth.composite_planes(n, L=L, dim=dim, device=ctx.device)
    out = torch.empty((n, dim, dim, 4), dtype=torch.uint8, device=ctx.device)
    for _ in range(3):
        ctx.composite(planes, canvas, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ctx.composite(planes, canvas, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    b = n * (L * dim * dim * 32 + dim * dim * 4)
    print(f"dim={dim:5d} L={L} n={n:4d}: {ms:7.3f} ms  {b/ms/1e6:8.1f} GB/s  ({b/ms/1e6/8000*100:5.1f} % of 8 TB/s)")
    del planes, out
