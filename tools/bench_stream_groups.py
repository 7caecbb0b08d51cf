#!/usr/bin/env python
"""Experiment (round 3): the 1024-tile batch as G separate resident scenes rendered on G streams.  0.98 -> 0.91 ms per
step with G = 4 — but the gain is CROSS-STEP overlap (scene g's next step starts while scene g+1 is still in this one):
the same grouping INSIDE one step of one scene (tile groups on four streams joined back into the caller's stream, built
and tested bit-exact, then removed) was slower: 0.978 -> 1.007 / 1.043 / 1.070 ms for 2 / 3 / 4 groups.  What this
measures is what worker threads with their own scenes get (end_to_end.png_files_worker_threads_tiles_per_s)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context
ctx = Context(0)
def bench(groups, cfg="config2", n=1024, reps=20):
    per = n // groups
    tiles = synth.config_tiles(n)
    if cfg == "config2":
        dls = [synth.make_tiles(tiles[g * per:(g + 1) * per], zoom=15, scale=1, n_poly=50, n_line=40) for g in range(groups)]
    else:
        dls = [synth.make_tiles(synth.config_tiles(n, x0=79000, y0=40000)[g * per:(g + 1) * per], zoom=17, scale=1, n_poly=5000, n_line=4000, radius=(2.0, 12.0), step=12.0) for g in range(groups)]
    scenes = [ctx.upload(d) for d in dls]
    outs = [torch.empty((per, 256, 256, 4), dtype=torch.uint8, device=ctx.device) for _ in range(groups)]
    streams = [torch.cuda.Stream() for _ in range(groups)]
    def step():
        for g in range(groups):
            ctx.render(scenes[g], outs[g], stream=streams[g])
    for _ in range(3): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    for s in scenes: s.free()
    return dt * 1e3
for g in (1, 2, 4, 8):
    print("config2 groups", g, "ms/step %.3f" % bench(g))
for g in (1, 2, 4):
    print("config5(64) groups", g, "ms/step %.3f" % bench(g, "config5", 64, 5))
