#!/usr/bin/env python
"""Label pass cost on top of the config-2 area workload (run on a GPU box).

    python tools/bench_labels.py [tiles] [labels_per_tile] [steps]

Synthetic labels (osm_renderer_amd.labels.make_labels: TrueType-like quadratic outlines flattened exactly like
Rasterizer::draw_quad, ~1000 draw_line calls per label, 40 % with an icon, 30 % rotated) are generated for a pool of 64
tiles and repeated; prints tiles/s without and with the label pass and the labels/s of the label kernels alone."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from osm_renderer_amd import labels, synth
from osm_renderer_amd.renderer import Context


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    per_tile = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    ctx = Context(0)
    rng = np.random.default_rng(1)
    sizes = [(16, 16), (12, 20), (20, 20)]
    for h, w in sizes:
        img = rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8)
        ctx.register_image(img)
    dl = synth.config2(n_tiles)
    pool = min(64, n_tiles)
    t0 = time.time()
    base = labels.make_labels(pool, labels_per_tile=per_tile, n_images=3, image_sizes=sizes, seed=2)
    ll = labels.concat_labels([base.subset([i % pool]) for i in range(n_tiles)]) if n_tiles != pool else base
    gen_s = time.time() - t0
    scene = ctx.upload(dl)
    out = torch.empty((n_tiles, 256, 256, 4), dtype=torch.uint8, device=ctx.device)

    def timed():
        for _ in range(2):
            ctx.render(scene, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            ctx.render(scene, out)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    ms_plain = timed()
    t0 = time.time()
    scene.set_labels(ll)
    set_s = time.time() - t0
    ms_lab = timed()
    ok = scene.label_status()
    print(json.dumps({
        "tiles": n_tiles, "labels": int(len(ll.labels)), "draw_line_calls": int(len(ll.segs)),
        "labels_succeeded": int(ok.sum()), "ms_areas_only": round(ms_plain, 3), "ms_with_labels": round(ms_lab, 3),
        "tiles_per_s_areas_only": round(n_tiles / ms_plain * 1e3), "tiles_per_s_with_labels": round(n_tiles / ms_lab * 1e3),
        "label_pass_ms": round(ms_lab - ms_plain, 3), "labels_per_s": round(len(ll.labels) / max(ms_lab - ms_plain, 1e-9) * 1e3),
        "label_input_GBps": round(ll.algorithmic_bytes() / max(ms_lab - ms_plain, 1e-9) / 1e6, 2),
        "host_generate_s": round(gen_s, 2), "host_set_labels_s": round(set_s, 3),
    }))


if __name__ == "__main__":
    main()
