#!/usr/bin/env python
"""Randomised GPU-vs-oracle differential fuzzing beyond the pytest suite (run on a GPU box).

    python tools/fuzz_parity.py [seconds] [seed] [labels]

Random tiles with adversarial parameters: widths 0..40, all directions, tiny / huge dashes, every
cap, use_caps_for_dashes, scales 1..3, far-away and huge coordinates, self-intersecting and
multi-ring fills, many ops.  Any differing pixel is printed with the op that was drawn last.

Round 5: the batch size cycles through 1, 12, 64, 65 and 130 tiles, a tile has 0 .. 300 ops (one tile in ten has none), so
every run drives BOTH raster instantiations — k_raster<FOLD> (batches of at most 64 tiles: tiles of at most 128 ops build
their lists in the raster kernel, bigger ones get them from k_sublist) and k_sublist -> k_raster<false> (batches of more
than 64 tiles) — and it runs on poisoned device memory (OSMT_POISON_ALLOC=1 unless the environment says otherwise)."""
import os, sys, time
os.environ.setdefault("OSMT_POISON_ALLOC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from osm_renderer_amd import abi, display_list
from osm_renderer_amd.display_list import TileBuilder
from osm_renderer_amd.renderer import Context
from oracle import oracle_py as O

CAPS = [abi.CAP_NONE, abi.CAP_BUTT, abi.CAP_ROUND, abi.CAP_SQUARE]
rnd = None


def rand_pts(n, W, spread):
    mode = rnd.integers(0, 5)
    p0 = rnd.integers(-40, W + 40, size=2)
    if mode == 0:  # far away / huge
        p0 = rnd.integers(-200000, 200000, size=2)
        spread = int(rnd.choice([50, 3000, 400000]))
    steps = rnd.integers(-spread, spread + 1, size=(n, 2))
    if mode == 1:  # axis aligned
        steps[:, rnd.integers(0, 2)] = 0
    if mode == 2:  # exact diagonals / repeats
        steps[:, 1] = steps[:, 0] * rnd.choice([-1, 1])
        steps[rnd.integers(0, n)] = 0
    return (p0 + np.cumsum(steps, axis=0)).tolist()


BATCH_SIZES = (1, 12, 64, 65, 130)


def rand_op_count():
    u = rnd.random()
    if u < 0.10:
        return 0  # plain canvas (drawer.rs:60-131 with no areas)
    if u < 0.80:
        return int(rnd.integers(1, 40))
    if u < 0.95:
        return int(rnd.integers(40, 141))  # both sides of OSMT_FOLD_MAX_OPS = 128
    return int(rnd.integers(141, 301))


def make_tile(scale, image_ids=(), n_ops=None):
    W = 256 * scale
    tb = TileBuilder(scale=scale, canvas=None if rnd.random() < 0.2 else tuple(rnd.integers(0, 256, size=3)))
    for _ in range(int(rnd.integers(1, 40)) if n_ops is None else n_ops):
        kind = rnd.random()
        col = tuple(rnd.integers(0, 256, size=3))
        op = float(rnd.choice([1.0, 1.0, 0.5, 0.25, 0.9, 0.0, 1.0 / 3.0]))
        if kind < 0.4:
            rings = [rand_pts(int(rnd.integers(2, 14)), W, int(rnd.choice([8, 40, 120]))) for _ in range(int(rnd.integers(1, 4)))]
            for r in rings:
                if rnd.random() < 0.8:
                    r.append(r[0])
            if image_ids and rnd.random() < 0.15:  # Filler::Image; sometimes an id nobody registered (draws nothing)
                tb.fill_image(rings, int(rnd.choice(image_ids)) if rnd.random() < 0.9 else 99999, op)
            else:
                tb.fill(rings, col, op)
        elif kind < 0.45:
            tb.nop()
        else:
            d = None
            if rnd.random() < 0.5:
                nd = int(rnd.integers(1, 7))
                d = [float(rnd.choice([0.3, 1.0, 2.0, 3.0, 5.5, 8.0, 13.0, 40.0])) * scale for _ in range(nd)]
            w = float(rnd.choice([0.0, 0.05, 0.2, 0.5, 0.99, 1.0, 1.01, 1.5, 2.0, 2.5, 3.0, 4.0, 7.0, 12.5, 25.0, 40.0, -3.0, 1.3, 2e-101, 0.7])) * scale
            tb.stroke(rand_pts(int(rnd.integers(2, 9)), W, int(rnd.choice([3, 30, 90, 300]))), w, col, op, dashes=d,
                      cap=CAPS[int(rnd.integers(0, 4))], use_caps_for_dashes=bool(rnd.integers(0, 2)))
            if rnd.random() < 0.2:  # a second stroke op over the SAME rings (casing + stroke of one way share their points)
                tb.stroke_again(max(0.0, abs(w) - float(rnd.choice([0.5, 1.0, 2.0]))), tuple(rnd.integers(0, 256, size=3)), op,
                                dashes=d if rnd.random() < 0.5 else None, cap=CAPS[int(rnd.integers(0, 4))])
    return tb.build()


def rand_label_text(scale):
    """adversarial draw_line calls: random polygons with float / integer / half-integer vertices, long and tiny edges,
    horizontal and degenerate ones, overlapping contours, both orientations, occasionally a contour far to the side"""
    from osm_renderer_amd import labels as L

    W = 256 * scale
    mode = int(rnd.integers(0, 6))
    if mode == 0:  # synthetic glyph run, maybe rotated
        segs, _ = L.synth_text(rnd, float(rnd.uniform(-W / 2, 1.5 * W)), float(rnd.uniform(-W / 2, 1.5 * W)),
                               float(rnd.choice([7.0, 10.0, 13.0, 22.0])) * scale, int(rnd.integers(1, 9)),
                               float(rnd.choice([0.0, 0.0, rnd.uniform(-3.1, 3.1)])))
        return segs
    out = []
    for _ in range(int(rnd.integers(1, 5))):
        n = int(rnd.integers(2, 9))
        c = rnd.uniform(-W / 2, 1.5 * W, size=2)
        r = float(rnd.choice([0.3, 2.0, 9.0, 40.0, 300.0]))
        pts = c + rnd.uniform(-r, r, size=(n, 2))
        q = int(rnd.integers(0, 4))
        if q == 0:
            pts = np.round(pts)
        elif q == 1:
            pts = np.round(pts * 2) / 2
        if mode == 5 and rnd.random() < 0.3:
            pts[:, 0] -= 3000.0  # far to the side, same stripes: the wide-window path
        if rnd.random() < 0.3:
            pts[int(rnd.integers(0, n))] = pts[int(rnd.integers(0, n))]  # zero-length edge
        if rnd.random() < 0.3:
            pts[1, 1] = pts[0, 1]  # horizontal edge
        closed = np.concatenate([pts, pts[:1]])
        if rnd.random() < 0.5:
            closed = closed[::-1]
        for a, b in zip(closed[:-1], closed[1:]):
            out.append((a[0], a[1], b[0], b[1]))
    return np.array(out, dtype=np.float64).reshape(-1, 4)


def make_labels(n_tiles, scale, image_ids, image_sizes):
    from osm_renderer_amd import labels as L

    W = 256 * scale
    out = []
    for _ in range(n_tiles):
        tl = L.TileLabels()
        for _ in range(int(rnd.integers(0, 25))):
            icon = text = None
            if image_ids and rnd.random() < 0.4:
                k = int(rnd.integers(0, len(image_ids)))
                icon = (image_ids[k] if rnd.random() < 0.95 else 99999, float(rnd.uniform(-1.2 * W, 2.2 * W)) + float(rnd.choice([0.0, 0.5])),
                        float(rnd.uniform(-1.2 * W, 2.2 * W)))
            if rnd.random() < 0.8:
                text = (tuple(int(v) for v in rnd.integers(0, 256, size=3)), rand_label_text(scale))
            tl.label(icon=icon, text=text)
        out.append(tl.build())
    return L.concat_labels(out)


def run(budget=60.0, seed=1, ctx=None, dump=True, with_labels=False, min_iters=0):
    """Fuzz for `budget` seconds — and for at least `min_iters` iterations whatever the budget (a test that asserts on WHAT was
    covered asks for one full cycle of BATCH_SIZES instead of hoping that a time budget reached it on a slow or loaded host);
    returns (tiles rendered, mismatching tiles)."""
    global rnd
    rnd = np.random.default_rng(seed)
    ctx = ctx or Context(0)
    images, image_ids, sizes = [], [], ((16, 16), (9, 9), (5, 23))
    if True:  # icons for image fills (area runs) and label icons (label runs)
        arrays = []
        for h, w in sizes:
            img = rnd.integers(0, 256, size=(h, w, 4)).astype(np.uint8)
            img[: h // 2, :, 3] = 255
            arrays.append(img)
            image_ids.append(ctx.register_image(img))
        images = [np.zeros((1, 1, 4), dtype=np.uint8) for _ in range(max(image_ids) + 1)]  # ids are registry positions
        for i, img in zip(image_ids, arrays):
            images[i] = img
    t0 = time.time()
    n_tiles = n_bad = 0
    stats = {"batches": {}, "empty_tiles": 0, "folded_tiles": 0, "listed_tiles": 0}
    it = 0
    while time.time() - t0 < budget or it < min_iters:
        sizes_now = BATCH_SIZES[:4] if with_labels else BATCH_SIZES  # (the label oracle is the slow side: no 130-tile label batches)
        bs = sizes_now[it % len(sizes_now)]
        it += 1
        scale = int(rnd.choice([1, 1, 2, 3])) if bs <= 12 else int(rnd.choice([1, 1, 1, 2]))
        counts = [rand_op_count() for _ in range(bs)]
        if bs > 12:  # the oracle renders these too: mostly short lists in the big batches, a few long ones
            counts = [c if (c <= 40 or k % 9 == 0) else c % 40 for k, c in enumerate(counts)]
        tiles = [make_tile(scale, image_ids, c) for c in counts]
        stats["batches"][bs] = stats["batches"].get(bs, 0) + 1
        stats["empty_tiles"] += sum(1 for c in counts if c == 0)
        stats["folded_tiles"] += sum(1 for c in counts if bs <= 64 and c <= 128)
        stats["listed_tiles"] += sum(1 for c in counts if bs > 64 or c > 128)
        dl = display_list.concat(tiles)
        ll = make_labels(len(tiles), scale, image_ids, sizes) if with_labels else None
        if ll is not None:
            scene = ctx.upload(dl, ll)
            got = ctx.render(scene).cpu().numpy()
            st = scene.label_status()
            scene.free()
            want, wst = O.render_batch(dl, threads=12, images=images, labels=ll, want_status=True)
            if not np.array_equal(st, wst):
                n_bad += 1
                print(f"LABEL STATUS MISMATCH seed={seed} after {n_tiles} tiles: labels {np.nonzero(st != wst)[0][:8].tolist()}")
        else:
            got = ctx.render_batch_host(dl)
            want = O.render_batch(dl, threads=12, images=images)
        n_tiles += len(tiles)
        if not np.array_equal(got, want):
            for i in range(len(tiles)):
                if not np.array_equal(got[i], want[i]):
                    n_bad += 1
                    ys, xs = np.nonzero((got[i] != want[i]).any(-1))
                    print(f"MISMATCH seed={seed} tile#{n_tiles - len(tiles) + i} scale={scale}: {len(ys)} px, first at x={xs[0]} y={ys[0]} "
                          f"gpu={got[i][ys[0], xs[0]].tolist()} oracle={want[i][ys[0], xs[0]].tolist()}")
                    if not dump:
                        continue
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.save(f"gpurun_out/fuzz_bad_{seed}_{n_tiles}_{i}_ops.npy", tiles[i].ops)
                    np.save(f"gpurun_out/fuzz_bad_{seed}_{n_tiles}_{i}_coords.npy", tiles[i].coords)
                    np.save(f"gpurun_out/fuzz_bad_{seed}_{n_tiles}_{i}_dashes.npy", tiles[i].dashes)
                    np.save(f"gpurun_out/fuzz_bad_{seed}_{n_tiles}_{i}_rings.npy", tiles[i].rings)
                    if ll is not None:
                        one = ll.subset([i])
                        np.save(f"gpurun_out/fuzz_bad_{seed}_{n_tiles}_{i}_labels.npy", one.labels)
                        np.save(f"gpurun_out/fuzz_bad_{seed}_{n_tiles}_{i}_segs.npy", one.segs)
    from osm_renderer_amd.lib import load as _load

    print(f"fuzz: {n_tiles} tiles in {time.time() - t0:.0f} s, {n_bad} mismatching tiles (seed {seed}, labels {bool(with_labels)}, "
          f"poison {_load().osmt_debug_poison_enabled()}); batches by size {dict(sorted(stats['batches'].items()))}, "
          f"{stats['empty_tiles']} empty tiles, {stats['folded_tiles']} tiles through k_raster<FOLD>, {stats['listed_tiles']} through k_sublist")
    run.last_stats = stats
    return n_tiles, n_bad


if __name__ == "__main__":
    _, bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                 with_labels=len(sys.argv) > 3 and sys.argv[3] == "labels")
    sys.exit(1 if bad else 0)
