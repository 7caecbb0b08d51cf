#!/usr/bin/env python
"""Composite pass (64 x 512^2, L = 8) of several builds of the library (OSMT_LIB), one process each: ms per launch, TB/s, checksum.

    python tools/time_composite.py base v1 v2 ...        ("base" = libosmtile.so, others libosmtile_<v>.so)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import torch
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context
ctx = Context(0)
n, L, dim = 64, 8, 512
planes = synth.composite_planes(n, L=L, dim=dim, device=ctx.device)
out = torch.empty((n, dim, dim, 4), dtype=torch.uint8, device=ctx.device)
canvas = [0xFC / 255.0, 0xF8 / 255.0, 0xE4 / 255.0, 1.0]
for _ in range(3):
    ctx.composite(planes, canvas, out=out)
best = 1e9
for rep in range(3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); ctx.composite(planes, canvas, out=out); b.record()
    torch.cuda.synchronize()
    best = min(best, sum(a.elapsed_time(b) for a, b in ev) / len(ev))
byts = n * (L * dim * dim * 32 + dim * dim * 4)
print(json.dumps({"ms": best, "tbs": byts / best / 1e9, "frac": byts / best / 1e9 / 8.0, "sum": int(out.to(torch.int64).sum().item())}))
'''
for v in sys.argv[1:]:
    lib = os.path.join(ROOT, "osm_renderer_amd", "libosmtile.so" if v == "base" else f"libosmtile_{v}.so")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ, OSMT_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    try:
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        print(f"{v:12s} {d['ms']:.4f} ms  {d['tbs']:.3f} TB/s  {d['frac']:.4f} of 8 TB/s  checksum {d['sum']}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(v, "FAILED", e, r.stderr.decode(errors="replace")[-300:], flush=True)
