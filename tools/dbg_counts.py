"""Diagnostic: per-wave work counters of k_raster from a library built with -DOSMT_ABL=5 (the counters replace the
first 8 pixels of every sub-tile's first row).  OSMT_LIB=.../libosmtile_dbg.so python tools/dbg_counts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context

ctx = Context(0)
dl = synth.config2(64)
out = ctx.render(ctx.upload(dl)).cpu().numpy().view(np.uint32).reshape(64, 256, 256)
c = out[:, ::16, :].reshape(64, 16, 8, 32)[:, :, :, :8].reshape(-1, 8).astype(np.int64)
names = ["stroke_visits", "passes", "items", "lane_iters", "-", "fill_visits", "set_pixels", "max_iters_seen"]
print("waves", len(c))
for i, n in enumerate(names):
    print(f"{n:16s} per wave {c[:, i].mean():10.2f}   per tile {c[:, i].sum() / 64:12.1f}")
print("items per pass", c[:, 2].sum() / max(c[:, 1].sum(), 1), " iters per item", c[:, 3].sum() / max(c[:, 2].sum(), 1),
      " set pixels per item", c[:, 6].sum() / max(c[:, 2].sum(), 1))
