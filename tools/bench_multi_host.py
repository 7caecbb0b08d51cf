#!/usr/bin/env python
"""Host side of osmt_render_batch_multi without eight GPUs: G contexts on the devices that ARE visible (all on device 0
of a one-GPU box), the 10 000-tile config-4 batch, and the share of the call that runs on ONE thread.

    python tools/bench_multi_host.py [tiles] [G ...]

The library reports (OSMT_TRACE_MULTI=1) the serial head of a call — the O(n_jobs log n_jobs) partition check — the
parallel part (per shard: validation of its own jobs, packing, upload, kernels, read-back) and the serial tail (the tile-count
sum).  With G contexts on one device the parallel part does not speed up (one GPU does all the kernels), but the serial
share of the call is what Amdahl's law needs: predicted speed-up at G real GPUs = 1 / (s + (1 - s) / G) with s measured at
G = 1.  One JSON line per G."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np, torch
from osm_renderer_amd import shard, synth
from osm_renderer_amd.renderer import Context
n_tiles, G = %d, %d
n_dev = torch.cuda.device_count()
ctxs = [Context(d %% n_dev) for d in range(G)]
dl = synth.make_tiles(synth.config_tiles(n_tiles))
pin = ctxs[0].host_alloc((dl.n_jobs, dl.dim, dl.dim, 4))
shard.render_batch_multi(ctxs, dl, out=pin)  # warm-up: buffers, streams
t = []
for _ in range(3):
    t0 = time.perf_counter()
    _, cnt = shard.render_batch_multi(ctxs, dl, out=pin)
    t.append(time.perf_counter() - t0)
    assert cnt == dl.n_jobs
print(json.dumps({"wall_ms": [x * 1e3 for x in t], "devices": n_dev}))
'''


def run(n_tiles, G):
    env = dict(os.environ, OSMT_TRACE_MULTI="1")
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, n_tiles, G)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode().strip().splitlines()
    if r.returncode != 0 or not out:
        return {"contexts": G, "error": r.stderr.decode(errors="replace")[-400:]}
    res = json.loads(out[-1])
    lines = re.findall(r"osmt multi: (\d+) contexts, (\d+) tiles: serial head ([0-9.]+) us, parallel ([0-9.]+) us, serial tail ([0-9.]+) us",
                       r.stderr.decode(errors="replace"))
    last = lines[-3:]  # the three timed calls
    head = sum(float(x[2]) for x in last) / len(last)
    par = sum(float(x[3]) for x in last) / len(last)
    tail = sum(float(x[4]) for x in last) / len(last)
    total = head + par + tail
    return {"contexts": G, "devices": res["devices"], "tiles": n_tiles, "call_ms": min(res["wall_ms"]), "serial_head_us": head, "parallel_us": par,
            "serial_tail_us": tail, "serial_fraction": (head + tail) / total, "tiles_per_s": n_tiles / (min(res["wall_ms"]) * 1e-3)}


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    gs = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
    base = None
    for G in gs:
        r = run(n_tiles, G)
        if "error" not in r:
            if base is None:
                base = r
            s = base["serial_fraction"]
            r["predicted_speedup_at_G_gpus"] = 1.0 / (s + (1.0 - s) / G)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
