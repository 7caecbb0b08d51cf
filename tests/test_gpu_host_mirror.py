"""The C++ host mirror of the reference's draw interface (osm_renderer_amd/host/osmt_draw.hpp)
driven like Drawer::draw_to_pixels, compared with the oracle driven by the same calls."""
import os
import subprocess

import numpy as np
import pytest

from osm_renderer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "host_mirror_demo")


def _oracle_tiles(oracle):
    from osm_renderer_amd import labels
    from osm_renderer_amd.display_list import TileBuilder

    tb = TileBuilder(canvas=(241, 238, 232))
    tb.fill([[(10, 10), (200, 30), (150, 220), (20, 180), (10, 10)]], (200, 40, 40), 0.6)
    tb.nop()
    tb.fill([[(60, 60), (120, 70), (100, 130), (60, 60)], [(300, 300), (310, 300), (305, 320), (300, 300)]], (20, 40, 220), 1.0)
    tb.stroke([(5, 250), (90, 120), (180, 200), (250, 20)], 6.0, (10, 120, 10), 0.8, dashes=[9.0, 4.0], cap=abi.CAP_ROUND)
    tb.stroke([(0, 0), (255, 255)], 1.5, (0, 0, 0), 1.0)

    def square(x0, y0, x1, y1):
        return [(x0, y0, x0, y1), (x0, y1, x1, y1), (x1, y1, x1, y0), (x1, y0, x0, y0)]

    tl = labels.TileLabels()
    tl.label(text=((102, 102, 255), square(40.25, 40.5, 70.75, 52.125)))
    tl.label(text=((255, 0, 0), square(60, 45, 90, 60)))  # collides with the first one
    curve = [(150, 150, 150, 180)]
    labels.flatten_quad(150, 180, 190, 165, 150, 150, curve)
    tl.label(text=((0, 0, 0), curve))
    ta, st = oracle.render_job(tb.build(), 0, labels=tl.build(), want_status=True)
    assert st.tolist() == [1, 0, 1]
    b = oracle.Pixels(1)
    b.reset(None)
    b.draw_lines(oracle.ring_to_pairs([(30, 30), (220, 60)]), 12.0, (255, 200, 0), 0.5, cap=abi.CAP_SQUARE,
                 use_caps_for_dashes=True)
    b.bump_generation()
    b.blend_unfinished_pixels()
    return ta[..., :3], b.to_rgb()


def test_host_mirror_builds():
    """CPU-side: the mirror header compiles and links against the C ABI (no GPU call)."""
    from tests._hostdemo import build_demo

    assert os.path.exists(build_demo())


@pytest.mark.gpu
def test_host_mirror_matches_oracle(gpu_ctx, oracle, tmp_path):
    from tests._hostdemo import build_demo

    out = tmp_path / "mirror.rgb"
    env = dict(os.environ)
    import torch

    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env["LD_LIBRARY_PATH"] = torch_lib + ":" + env.get("LD_LIBRARY_PATH", "")
    png = tmp_path / "a.png"
    subprocess.check_call([build_demo(), str(out), str(png)], env=env)
    raw = np.fromfile(out, dtype=np.uint8).reshape(3, 256, 256, 3)
    ta, tb = _oracle_tiles(oracle)
    np.testing.assert_array_equal(raw[0], ta)
    from PIL import Image

    np.testing.assert_array_equal(np.array(Image.open(png).convert("RGB")), ta)  # TileBatch::render_png
    np.testing.assert_array_equal(raw[1], ta)
    np.testing.assert_array_equal(raw[2], tb)
