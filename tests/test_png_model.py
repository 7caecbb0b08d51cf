"""The CPU model of the GPU PNG encoder (tests/_png_model.py) writes valid PNG files under the shipped prefix code:
PIL and zlib decode them to the input.  (The GPU kernel is compared with the model byte for byte in
tests/test_gpu_png_device.py; this is the half of that argument that needs no GPU.)"""
import io
import json
import os
import re
import zlib

import numpy as np

from tests import _png_model


def _images():
    rng = np.random.default_rng(5)
    imgs = np.zeros((4, 64, 96, 4), dtype=np.uint8)
    imgs[0] = rng.integers(0, 256, size=(64, 96, 4))      # every literal, the long codes included
    imgs[1, :, :] = (241, 238, 232, 255)                  # flat: 258-byte matches
    imgs[2, :, :, :3] = (np.arange(96)[None, :, None] // 3).astype(np.uint8)
    imgs[3, 10:40, 20:70, :3] = rng.integers(0, 4, size=(30, 50, 3)) * 70
    imgs[..., 3] = 255
    return imgs


def test_model_files_decode_to_the_input():
    from PIL import Image

    for img in _images():
        png = _png_model.encode(img)
        assert np.array_equal(np.array(Image.open(io.BytesIO(png)).convert("RGB")), img[..., :3])
        idat_len = int.from_bytes(png[33:37], "big")
        assert len(zlib.decompress(png[41 : 41 + idat_len])) == img.shape[0] * (3 * img.shape[1] + 1)


def test_model_with_hash_matches_decodes_to_the_input():
    """256-wide images take the kernel's fast path, the one that searches for matches: repeated structure a few rows apart
    (what a map tile is made of), noise and a flat tile"""
    from PIL import Image

    rng = np.random.default_rng(11)
    imgs = np.zeros((4, 64, 256, 4), dtype=np.uint8)
    imgs[..., :3] = (241, 238, 232)
    for y in range(64):  # slanted anti-aliased lines: the same residuals come back shifted, row after row
        for x0, sl in ((10, 0.5), (90, 1.5), (170, -0.75)):
            x = int(x0 + sl * y) % 250
            imgs[0, y, x : x + 3, :3] = ((200, 120, 40), (150, 90, 30), (220, 200, 190))
    imgs[1, :, :, :3] = rng.integers(0, 256, size=(64, 256, 3))
    imgs[2, 8:40, 30:200, :3] = rng.integers(0, 3, size=(32, 170, 1)) * 90
    imgs[..., 3] = 255
    n_match = 0
    for img in imgs:
        png = _png_model.encode(img)
        assert np.array_equal(np.array(Image.open(io.BytesIO(png)).convert("RGB")), img[..., :3])
        toks, _ = _png_model.tile_tokens(img)
        n_match += sum(1 for t in toks if t[0] == "match" and t[2] != 1)
        assert all(t[0] == "lit" or (3 <= t[1] <= 258 and 1 <= t[2] <= 32768) for t in toks)
    assert n_match > 50
    assert len(_png_model.encode(imgs[0])) < len(_png_model.encode(imgs[0], lz=False))


def test_code_is_complete_and_bounded_and_matches_the_header_the_kernel_ships():
    t = _png_model._T
    lens = t["litlen_lengths"]
    assert len(lens) == 286 and min(lens) >= 1 and max(lens) <= t["lmax"]
    assert sum(2.0 ** -l for l in lens) == 1.0  # Kraft: a complete prefix code
    # csrc/osmt_png_table.h carries the same code (bit-reversed code | length << 16) and the same header bits
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "osm_renderer_amd", "csrc", "osmt_png_table.h")).read()
    table = [int(x, 16) for x in re.findall(r"0x([0-9A-F]{8})u", src.split("png_code_table[286]")[1].split("png_dist_table")[0])]
    assert len(table) == 286
    for s in range(286):
        code, n = _png_model.lit_token(s) if s < 257 else (_png_model.rev(_png_model.CODES[s], lens[s]), lens[s])
        assert table[s] == (code | (n << 16))
    dl = t["dist_lengths"]
    assert len(dl) == 30 and min(dl) >= 1 and max(dl) <= t["lmax"] and sum(2.0 ** -l for l in dl) == 1.0
    dtable = [int(x, 16) for x in re.findall(r"0x([0-9A-F]{8})u", src.split("png_dist_table[30]")[1])]
    assert len(dtable) == 30
    for s in range(30):
        assert dtable[s] == (_png_model.rev(_png_model.DCODES[s], dl[s]) | (dl[s] << 16))
    head = [int(x, 16) for x in re.findall(r"0x([0-9A-F]{8})u", src.split("png_head_words[PNG_HEAD_WORDS]")[1].split(";")[0])]
    acc = 0
    for k, w in enumerate(head):
        acc |= w << (32 * k)
    assert acc & 0xFFFFFF == 0x017854  # 'T', then the zlib header 78 01
    assert (acc >> 24) == int(t["block_header_hex"], 16)
    assert int(re.search(r"PNG_BLOCK_HDR_BITS (\d+)u", src).group(1)) == t["block_header_bits"]
