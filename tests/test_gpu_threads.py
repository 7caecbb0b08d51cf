"""The reference renders from N worker threads at once (src/http_server.rs:50-83), each with its own
TilePixels.  The C ABI promises the same: any number of host threads may use one osmt_ctx."""
import threading

import numpy as np
import pytest

from osm_renderer_amd import synth

pytestmark = pytest.mark.gpu


def test_concurrent_host_threads_share_one_context(gpu_ctx, oracle):
    import torch

    n_threads, per = 6, 5
    lists = [synth.make_tiles(synth.config_tiles(per, x0=19000 + 37 * t, y0=10000 + t), scale=1 + (t % 2)) for t in range(n_threads)]
    want = [oracle.render_batch(dl, threads=4) for dl in lists]
    got = [None] * n_threads
    errors = []

    def worker(t):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            for rep in range(4):
                if rep % 2 == 0:  # HBM-resident path on a private stream
                    with torch.cuda.stream(stream):
                        scene = gpu_ctx.upload(lists[t])
                        out = gpu_ctx.render(scene, stream=stream)
                        stream.synchronize()
                        got[t] = out.cpu().numpy()
                        scene.free()
                else:  # host-buffer path
                    got[t] = gpu_ctx.render_batch_host(lists[t])
                assert np.array_equal(got[t], want[t]), f"thread {t} rep {rep}"
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(n_threads):
        np.testing.assert_array_equal(got[t], want[t])
