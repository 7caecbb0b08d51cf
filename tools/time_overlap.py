#!/usr/bin/env python
"""How a caller that renders batch after batch should queue the stages (GPU): config-2 steps over S resident copies of the
batch, (a) one stream per copy (bench.py's --streams), (b) the pre-pass stages of every step on ONE high-priority stream
and the raster stages on a normal one, chained by events.  Prints ms per step for each arrangement.

    python tools/time_overlap.py [tiles=1024] [steps=200]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osm_renderer_amd import abi, synth  # noqa: E402
from osm_renderer_amd.renderer import Context  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = Context(0)
dl = synth.config2(tiles)
PRE = abi.STAGE_PROJECT | abi.STAGE_OPINFO


def measure(name, slots, body):
    scenes = [ctx.upload(dl) for _ in range(slots)]
    outs = [torch.empty((dl.n_jobs, dl.dim, dl.dim, 4), dtype=torch.uint8, device=ctx.device) for _ in range(slots)]
    state = body(scenes, outs)
    for i in range(20):
        state(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        state(20 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{name:56s} {ms:.4f} ms/step  {dl.n_jobs / ms * 1e3:,.0f} tiles/s  sum {int(outs[0].to(torch.int64).sum().item())}", flush=True)
    for s in scenes:
        s.free()


def per_copy_streams(slots):
    def body(scenes, outs):
        lanes = [torch.cuda.Stream() for _ in range(slots)]

        def step(i):
            k = i % slots
            with torch.cuda.stream(lanes[k]):
                ctx.render_stages(scenes[k], PRE)
                ctx.render_stages(scenes[k], abi.STAGE_RASTER, outs[k])
        return step
    return body


def split_streams(slots, pre_priority, ras_priority):
    def body(scenes, outs):
        s_pre = torch.cuda.Stream(priority=pre_priority)
        s_ras = torch.cuda.Stream(priority=ras_priority)
        done_pre = [torch.cuda.Event() for _ in range(slots)]
        done_ras = [torch.cuda.Event() for _ in range(slots)]
        used = [False] * slots

        def step(i):
            k = i % slots
            with torch.cuda.stream(s_pre):
                if used[k]:
                    s_pre.wait_event(done_ras[k])  # the copy's lists are rebuilt: its previous raster stage must be over
                ctx.render_stages(scenes[k], PRE)
                done_pre[k].record(s_pre)
            with torch.cuda.stream(s_ras):
                s_ras.wait_event(done_pre[k])
                ctx.render_stages(scenes[k], abi.STAGE_RASTER, outs[k])
                done_ras[k].record(s_ras)
            used[k] = True
        return step
    return body


measure("1 copy, 1 stream", 1, per_copy_streams(1))
measure("2 copies, one stream each", 2, per_copy_streams(2))
measure("2 copies, pre-pass stream / raster stream, same priority", 2, split_streams(2, 0, 0))
measure("2 copies, pre-pass stream HIGH priority", 2, split_streams(2, -1, 0))
measure("3 copies, pre-pass stream HIGH priority", 3, split_streams(3, -1, 0))
measure("2 copies, raster stream HIGH priority", 2, split_streams(2, 0, -1))
