#!/bin/bash
# round-4 run g: composite variants; adaptive arena sizing (test + PNG begin/end rate)
O=gpurun_out/r04_g; mkdir -p $O
timeout 600 python tools/time_composite.py base comppipe comppipe4 comp32 comp8 base > $O/composite_variants.txt 2>&1; cat $O/composite_variants.txt
for v in comppipe comppipe4; do OSMT_LIB=$PWD/osm_renderer_amd/libosmtile_$v.so timeout 200 python -m pytest tests/test_gpu_projection_composite.py -q -x 2>&1 | tail -1; done
timeout 400 python -m pytest tests/test_gpu_fullsize_and_errors.py tests/test_gpu_png_device.py -x -q 2>&1 | tail -2
timeout 300 python - <<'PY' 2>&1 | tail -5
import time, numpy as np
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context
ctx = Context(0)
dl = synth.config2(1024)
pb = [ctx.host_alloc((dl.n_jobs * 96 * 1024,)) for _ in range(2)]
for _ in range(2):
    ctx.png_end(ctx.png_begin(dl), pb[0])
for rep in range(2):
    n_pipe = 10
    t0 = time.perf_counter()
    prev = ctx.png_begin(dl)
    for k in range(1, n_pipe):
        cur = ctx.png_begin(dl)
        ctx.png_end(prev, pb[(k - 1) & 1])
        prev = cur
    ctx.png_end(prev, pb[(n_pipe - 1) & 1])
    dt = (time.perf_counter() - t0) / n_pipe
    print("png begin/end: %.3f ms per batch, %.0f tiles/s" % (dt * 1e3, 1024 / dt))
ts = []
for _ in range(4):
    t0 = time.perf_counter(); ctx.render_batch_png(dl, out=pb[0], as_bytes=False); ts.append(time.perf_counter() - t0)
print("png one call: %.3f ms, %.0f tiles/s" % (min(ts) * 1e3, 1024 / min(ts)))
pin3 = ctx.host_alloc((dl.n_jobs, dl.dim * dl.dim * 3))
ts = []
for _ in range(4):
    t0 = time.perf_counter(); ctx.render_batch_rgb(dl, out=pin3); ts.append(time.perf_counter() - t0)
print("rgb8 one call: %.3f ms, %.0f tiles/s" % (min(ts) * 1e3, 1024 / min(ts)))
PY
