#!/usr/bin/env python
"""Per-section wave-time of k_raster (build variant -DOSMT_PROF: s_memtime deltas summed over all waves)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OSMT_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "osm_renderer_amd", "libosmtile_prof.so")
import torch
from osm_renderer_amd import synth, abi
from osm_renderer_amd.renderer import Context
from osm_renderer_amd.lib import load
names = ["init", "cull/compact", "op header", "stroke: records+scan / wait-for-blend", "stroke: items (walks)", "stroke: blend",
         "fill A (extents)", "fill B (sort+mask)", "fill C (blend)", "loop tail", "output"]
ctx = Context(0)
L = load()
buf = (C.c_ulonglong * 16)()
for label, kw in (("config2", {}), ("fills only", dict(n_line=0)), ("strokes only", dict(n_poly=0))):
    dl = synth.make_tiles(synth.config_tiles(1024), **kw)
    sc = ctx.upload(dl)
    out = ctx.render(sc)
    L.osmt_prof_read(buf, 1)
    ctx.render(sc, out=out)
    L.osmt_prof_read(buf, 1)
    tot = sum(buf[:11])
    print(f"== {label}: total wave-time {tot/1e9:.2f} G ticks")
    for i, n in enumerate(names):
        print(f"   {n:42s} {100*buf[i]/tot:6.2f} %")
