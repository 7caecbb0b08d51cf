"""osmt_worker_render — the per-request entry of the reference's server loop (src/http_server.rs:50-83,105-108,134-181:
one tile per request, available_parallelism() worker threads) — gathers the requests of concurrent threads into shared
launches.  Whatever group a request lands in, its pixels must be the ones the oracle computes for its tile alone."""
import threading

import numpy as np
import pytest

from osm_renderer_amd import labels, synth

pytestmark = pytest.mark.gpu


def _rgb(rgba):
    return np.ascontiguousarray(rgba[..., :3]).reshape(rgba.shape[0], -1)


def _hammer(gpu_ctx, singles, want, n_threads, calls, label_singles=None):
    errors, counts = [], [0] * n_threads
    start = threading.Barrier(n_threads)

    def body(t):
        try:
            w = gpu_ctx.worker()
            start.wait()
            for c in range(calls):
                i = (t * 7 + c * 13) % len(singles)
                got = w.render(singles[i], None if label_singles is None else label_singles[i])
                if not np.array_equal(got, want[i]):
                    bad = np.nonzero(got != want[i])
                    raise AssertionError(f"thread {t} call {c} tile {i}: {len(bad[0])} bytes differ, first at {int(bad[1][0])}")
                counts[t] += 1
            w.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=body, args=(t,)) for t in range(n_threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:4]
    assert counts == [calls] * n_threads


def test_sixteen_threads_of_one_tile_requests_match_the_oracle(gpu_ctx, oracle):
    dl = synth.make_tiles(synth.config_tiles(64, x0=19100, y0=10050))
    want_rgba = oracle.render_batch(dl, threads=8)
    singles = [dl.subset([i]) for i in range(dl.n_jobs)]
    want = [_rgb(want_rgba[i : i + 1]) for i in range(dl.n_jobs)]
    _hammer(gpu_ctx, singles, want, n_threads=16, calls=200)
    # one thread alone takes the direct path (a group of one): same pixels
    _hammer(gpu_ctx, singles, want, n_threads=1, calls=20)


def test_requests_of_different_shapes_and_labels_share_groups(gpu_ctx, oracle):
    """Requests with 1-3 tiles, with and without labels, NODE_REF beside nothing else of its kind, @2x beside @1x:
    compatible ones are merged (label lists and node tables re-based), the others wait for the next group."""
    rng = np.random.default_rng(5)
    sizes = [(16, 16), (12, 20)]
    imgs = [rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8) for h, w in sizes]
    ids = [gpu_ctx.register_image(i) for i in imgs]
    all_imgs = [np.zeros((1, 1, 4), dtype=np.uint8)] * (max(ids) + 1)
    for i, img in zip(ids, imgs):
        all_imgs[i] = img
    base = synth.make_tiles(synth.config_tiles(12, x0=19300, y0=10070))
    ll = labels.make_labels(12, labels_per_tile=6, n_images=len(ids), image_sizes=sizes, seed=3)
    ll.labels["image_id"] = np.asarray(ids, dtype=np.uint32)[ll.labels["image_id"] % len(ids)]
    hi = synth.make_tiles(synth.config_tiles(4, x0=19400, y0=10080), scale=2)
    refs = synth.make_tiles(synth.config_tiles(4, x0=19500, y0=10090)).with_node_refs()
    reqs, lab, want = [], [], []
    for i in range(0, 12, 3):  # three tiles per request, labelled
        idx = [i, i + 1, i + 2]
        reqs.append(base.subset(idx))
        lab.append(ll.subset(idx))
        want.append(_rgb(oracle.render_batch(reqs[-1], images=all_imgs, threads=3, labels=lab[-1])))
    for i in range(12):  # one tile, no labels
        reqs.append(base.subset([i]))
        lab.append(None)
        want.append(_rgb(oracle.render_batch(reqs[-1])))
    for i in range(4):
        reqs.append(hi.subset([i]))
        lab.append(None)
        want.append(_rgb(oracle.render_batch(reqs[-1])))
        reqs.append(refs.subset([i]))
        lab.append(None)
        want.append(_rgb(oracle.render_batch(reqs[-1])))
    _hammer(gpu_ctx, reqs, want, n_threads=12, calls=40, label_singles=lab)


def test_worker_errors_stay_with_their_request(gpu_ctx, oracle):
    """A request the validation refuses fails alone (OSMT_INVALID_ARG, like osmt_render_batch_rgb); the others go on."""
    from osm_renderer_amd.lib import OsmtError

    dl = synth.make_tiles(synth.config_tiles(8, x0=19600, y0=10100))
    want_rgba = oracle.render_batch(dl, threads=8)
    good = [dl.subset([i]) for i in range(8)]
    bad = dl.subset([0])
    bad.ops = bad.ops.copy()
    bad.ops["kind"][0] = 9  # unknown kind
    errors, seen_bad = [], []
    start = threading.Barrier(8)

    def body(t):
        try:
            w = gpu_ctx.worker()
            start.wait()
            for c in range(30):
                if t == 0 and c % 3 == 0:
                    try:
                        w.render(bad)
                        raise AssertionError("the malformed request was rendered")
                    except OsmtError as e:
                        seen_bad.append(e.code)
                else:
                    got = w.render(good[(t + c) % 8])
                    assert np.array_equal(got, _rgb(want_rgba[(t + c) % 8 : (t + c) % 8 + 1]))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=body, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:4]
    assert len(seen_bad) == 10 and all(c < 0 for c in seen_bad)
