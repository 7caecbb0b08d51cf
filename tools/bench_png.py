#!/usr/bin/env python
"""GPU PNG encoder (k_png_encode): kernel time on resident framebuffers, compression ratio, and the PCIe-inclusive
rate of osmt_render_batch_png next to osmt_render_batch (raw RGBA8 into pinned memory) and the host zlib encoder."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context, encode_png

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dl = synth.config2(n)
scene = ctx.upload(dl)
fb = ctx.render(scene)
torch.cuda.synchronize()
slots, lens = ctx.encode_png_device(fb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    slots, lens = ctx.encode_png_device(fb)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
tot = int(lens.sum().item())
print(f"k_png_encode: {ms:.3f} ms per {n} tiles = {n / ms * 1e3:.0f} tiles/s; reads {fb.numel() / ms / 1e6:.1f} GB/s of framebuffers; "
      f"{tot / n / 1024:.1f} KiB per PNG ({fb.numel() / tot:.2f}x smaller than RGBA8, {fb.numel() * 0.75 / tot:.2f}x than RGB8)")
pin_png = ctx.host_alloc((n * 96 * 1024,))
for fn, name in ((lambda: ctx.render_batch_png(dl), "osmt_render_batch_png (PNG files to pageable host memory)"),
                 (lambda: ctx.render_batch_png(dl, out=pin_png, as_bytes=False), "osmt_render_batch_png (PNG files to pinned host memory)")):
    fn()
    t = time.perf_counter()
    for _ in range(3):
        fn()
    dt = (time.perf_counter() - t) / 3
    print(f"{name}: {dt * 1e3:.2f} ms per {n} tiles = {n / dt:.0f} tiles/s PCIe-inclusive")
pin = ctx.host_alloc((n, 256, 256, 4))
ctx.render_batch_host(dl, out=pin)
t = time.perf_counter()
for _ in range(3):
    ctx.render_batch_host(dl, out=pin)
dt = (time.perf_counter() - t) / 3
print(f"osmt_render_batch (raw RGBA8 to pinned host): {dt * 1e3:.2f} ms per {n} tiles = {n / dt:.0f} tiles/s PCIe-inclusive")
host = fb[:16].cpu().numpy()
t = time.perf_counter()
sz = sum(len(encode_png(host[i], 6)) for i in range(16))
dt = (time.perf_counter() - t) / 16
print(f"host zlib level 6 (osmt_encode_png): {dt * 1e3:.2f} ms per tile on one core = {1 / dt:.0f} tiles/s/core, {sz / 16 / 1024:.1f} KiB per PNG")
