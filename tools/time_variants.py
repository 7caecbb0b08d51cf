#!/usr/bin/env python
"""Times the stages of a step for several builds of the library in ONE process launch each (kernel variants built with
osm_renderer_amd.build.build_variant, picked through OSMT_LIB): per-stage HIP-event times on config 2 / config 5 / @2x.

    python tools/time_variants.py base v1 v2 ...          ("base" = libosmtile.so)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys, time
import torch
sys.path.insert(0, %r)
from osm_renderer_amd import abi, synth
from osm_renderer_amd.renderer import Context
ctx = Context(0)
res = {}
def run(name, dl, reps):
    scene = ctx.upload(dl)
    out = torch.empty((dl.n_jobs, dl.dim, dl.dim, 4), dtype=torch.uint8, device=ctx.device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    pre = ras = 0.0
    for i in range(reps + 2):
        ev[0].record()
        ctx.render_stages(scene, abi.STAGE_PROJECT | abi.STAGE_OPINFO)
        ev[1].record()
        ctx.render_stages(scene, abi.STAGE_RASTER, out)
        ev[2].record()
        torch.cuda.synchronize()
        if i >= 2:
            pre += ev[0].elapsed_time(ev[1]); ras += ev[1].elapsed_time(ev[2])
    res[name] = {"prepass_ms": pre / reps, "raster_ms": ras / reps, "tiles_per_s": dl.n_jobs / ((pre + ras) / reps) * 1e3,
                 "checksum": int(out.to(torch.int64).sum().item())}
    scene.free()
run("config2", synth.config2(1024), 10)
run("raster_2x", synth.config3(256), 5)
run("config5", synth.config5(64), 3)
if %r:
    run("config5_256", synth.config5(256), 3)
print(json.dumps(res))
'''


BIG = bool(os.environ.get("OSMT_TIME_BIG"))


def main():
    for v in sys.argv[1:]:
        lib = os.path.join(ROOT, "osm_renderer_amd", "libosmtile.so" if v == "base" else f"libosmtile_{v}.so")
        if not os.path.exists(lib):
            print(v, "MISSING", lib)
            continue
        env = dict(os.environ, OSMT_LIB=lib)
        try:
            r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, BIG)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            line = r.stdout.decode().strip().splitlines()[-1] if r.stdout.strip() else ""
            d = json.loads(line)
            print(v, " ".join(f"{k}: pre {x['prepass_ms']:.3f} ras {x['raster_ms']:.3f} ms ({x['tiles_per_s']:.0f} t/s, sum {x['checksum']})" for k, x in d.items()), flush=True)
        except Exception as e:  # noqa: BLE001
            print(v, "FAILED", type(e).__name__, e, r.stderr.decode(errors="replace")[-300:] if "r" in dir() else "", flush=True)


if __name__ == "__main__":
    main()
