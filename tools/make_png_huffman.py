#!/usr/bin/env python
"""Builds the ONE prefix code the GPU PNG encoder uses for every tile (csrc/osmt_png_table.h, tests/golden/png_huffman.json).

k_png_encode writes a single deflate block per tile whose tokens are literals and distance-1 runs of the Paeth-filtered
scanlines.  With the fixed Huffman code of RFC 1951 a literal costs 8-9 bits; a code fitted to the tile's own histogram
(a second pass on the GPU) was sized at -20 % in round 2 and not built.  This script measures that ONE code, fitted to
a corpus of map tiles and shipped as a constant "dynamic" block header, gets the same saving without a second pass:
the filtered bytes of map tiles look alike (zeros, small residuals of anti-aliased edges, a few flat colours).

corpus = config-2 tiles rendered by the oracle (the bench workload) + 256x256 crops of the reference's golden renders
(tests/rendered/1{4..8}_expected.png: real map tiles; only their byte statistics enter the table), half the weight each.
Code lengths are limited to LMAX bits so that a filtered byte never costs more than LMAX bits (osmt_png_device_bound).

Run in the build container (reads /root/reference for the real tiles; without it the synthetic half alone is used):
    python tools/make_png_huffman.py            # rewrites the header and the JSON, prints sizes on held-out tiles
"""
import heapq
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _png_model as png_model  # noqa: E402  (the tokeniser; its code tables are what this script replaces)
LMAX = 12
LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def paeth_filter(rgb):
    H, W, _ = rgb.shape
    raw = rgb.reshape(H, W * 3).astype(np.int32)
    a = np.zeros_like(raw); a[:, 3:] = raw[:, :-3]
    b = np.zeros_like(raw); b[1:] = raw[:-1]
    c = np.zeros_like(raw); c[1:, 3:] = raw[:-1, :-3]
    p = a + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
    return ((raw - pred) & 255).astype(np.uint8)


def histogram(rgb):
    """token histograms of one tile, tokenised by the model of the kernel (tests/_png_model.py: literals, distance-1 runs and the
    hash matches of round 6): (literal/length counts, distance counts, extra bits, tokens, filtered rows)"""
    toks, rows = png_model.tile_tokens(rgb)
    h = np.zeros(286, np.int64)
    hd = np.zeros(30, np.int64)
    extra = 0
    for t in toks:
        if t[0] == "lit":
            h[t[1]] += 1
        else:
            idx = max(i for i in range(29) if LEN_BASE[i] <= t[1])
            di = max(i for i in range(30) if png_model.DIST_BASE[i] <= t[2])
            h[257 + idx] += 1; hd[di] += 1
            extra += LEN_EXTRA[idx] + png_model.DIST_EXTRA[di]
    h[256] += 1
    return h, hd, extra, toks, np.frombuffer(b"".join(rows), dtype=np.uint8)


def limited_lengths(freq, lmax):
    """Huffman code lengths <= lmax for every symbol (all get a code): plain Huffman on frequencies lifted by a floor that
    is doubled until the longest code fits (a floor of total / 2^k bounds the depth)."""
    freq = np.asarray(freq, dtype=np.float64)
    floor = 0.0
    while True:
        f = np.maximum(freq, floor) + 1e-9
        heap = [(float(x), i, [i]) for i, x in enumerate(f)]
        heapq.heapify(heap)
        depth = [0] * len(f)
        nid = len(f)
        while len(heap) > 1:
            f1, _, a = heapq.heappop(heap); f2, _, b = heapq.heappop(heap)
            for s in a + b: depth[s] += 1
            heapq.heappush(heap, (f1 + f2, nid, a + b)); nid += 1
        if max(depth) <= lmax:
            return depth
        floor = max(floor * 2.0, freq.sum() / 2.0 ** (lmax + 4))


def canonical(lengths):
    """RFC 1951 3.2.2: codes from lengths (MSB-first integers)"""
    maxl = max(lengths)
    bl_count = [0] * (maxl + 1)
    for l in lengths:
        if l: bl_count[l] += 1
    code = 0; next_code = [0] * (maxl + 2)
    for bits in range(1, maxl + 1):
        code = (code + bl_count[bits - 1]) << 1
        next_code[bits] = code
    out = []
    for l in lengths:
        if l: out.append(next_code[l]); next_code[l] += 1
        else: out.append(0)
    return out


def rev(code, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (code & 1); code >>= 1
    return r


class Bits:
    def __init__(self): self.acc = 0; self.n = 0
    def put(self, v, nb): self.acc |= v << self.n; self.n += nb


def dynamic_header(litlen, dist):
    """bits of BFINAL=1, BTYPE=10 and the code description (RFC 1951 3.2.7); Huffman codes go out MSB first = reversed"""
    seq = list(litlen) + list(dist)
    rle = []  # (symbol, extra value, extra bits)
    i = 0
    while i < len(seq):
        v = seq[i]; j = i
        while j < len(seq) and seq[j] == v: j += 1
        n = j - i
        if v == 0:
            while n >= 11:
                m = min(n, 138); rle.append((18, m - 11, 7)); n -= m
            if n >= 3: rle.append((17, n - 3, 3)); n = 0
            rle += [(0, 0, 0)] * n
        else:
            rle.append((v, 0, 0)); n -= 1
            while n >= 3:
                m = min(n, 6); rle.append((16, m - 3, 2)); n -= m
            rle += [(v, 0, 0)] * n
        i = j
    clf = [0] * 19
    for s, _, _ in rle: clf[s] += 1
    cl_len = [0] * 19
    used = [s for s in range(19) if clf[s]]
    lens = limited_lengths([clf[s] for s in used], 7) if len(used) > 1 else [1]
    for s, l in zip(used, lens): cl_len[s] = l
    cl_code = canonical(cl_len)
    hclen = 19
    while hclen > 4 and cl_len[CL_ORDER[hclen - 1]] == 0: hclen -= 1
    b = Bits()
    b.put(1, 1); b.put(2, 2)
    b.put(len(litlen) - 257, 5); b.put(len(dist) - 1, 5); b.put(hclen - 4, 4)
    for k in range(hclen): b.put(cl_len[CL_ORDER[k]], 3)
    for s, ev, eb in rle:
        b.put(rev(cl_code[s], cl_len[s]), cl_len[s])
        if eb: b.put(ev, eb)
    return b


def encode_stream(toks, litlen, codes, dlen, dcodes, hdr):
    """the deflate stream of a tile's tokens under the table: for the self-check"""
    b = Bits(); b.put(hdr.acc, hdr.n)
    for t in toks:
        if t[0] == "lit":
            b.put(rev(codes[t[1]], litlen[t[1]]), litlen[t[1]])
            continue
        idx = max(k for k in range(29) if LEN_BASE[k] <= t[1])
        s = 257 + idx
        b.put(rev(codes[s], litlen[s]), litlen[s])
        if LEN_EXTRA[idx]: b.put(t[1] - LEN_BASE[idx], LEN_EXTRA[idx])
        di = max(k for k in range(30) if png_model.DIST_BASE[k] <= t[2])
        b.put(rev(dcodes[di], dlen[di]), dlen[di])
        if png_model.DIST_EXTRA[di]: b.put(t[2] - png_model.DIST_BASE[di], png_model.DIST_EXTRA[di])
    b.put(rev(codes[256], litlen[256]), litlen[256])
    return b.acc.to_bytes((b.n + 7) // 8, "little"), b.n


def main():
    from osm_renderer_amd import synth
    from oracle import oracle_py

    syn = oracle_py.render_batch(synth.config2(16), threads=8)[..., :3]
    real = []
    ref = "/root/reference/tests/rendered"
    if os.path.isdir(ref):
        from PIL import Image
        for z in (14, 15, 16, 17, 18):
            im = np.array(Image.open(f"{ref}/{z}_expected.png").convert("RGB"))
            for ty in range(im.shape[0] // 256):
                for tx in range(im.shape[1] // 256):
                    real.append(im[ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256])
    hs = [histogram(t) for t in syn]
    hr = [histogram(t) for t in real]
    train_s, test_s = hs[:12], hs[12:]
    train_r, test_r = hr[::2], hr[1::2]
    fs = sum(h[0] for h in train_s).astype(np.float64)
    mix = fs / fs.sum()
    if train_r:
        fr = sum(h[0] for h in train_r).astype(np.float64)
        mix = 0.5 * mix + 0.5 * fr / fr.sum()
    litlen = limited_lengths(mix, LMAX)
    assert min(litlen) >= 1 and max(litlen) <= LMAX
    assert abs(sum(2.0 ** -l for l in litlen) - 1.0) < 1e-12, "the code must be complete"
    codes = canonical(litlen)
    # the distance code, fitted the same way: every one of the 30 symbols gets a code (a match may sit anywhere within 32 KiB)
    fd = sum(h[1] for h in train_s).astype(np.float64)
    mixd = fd / fd.sum()
    if train_r:
        frd = sum(h[1] for h in train_r).astype(np.float64)
        mixd = 0.5 * mixd + 0.5 * frd / frd.sum()
    dlen = limited_lengths(mixd, LMAX)
    assert min(dlen) >= 1 and max(dlen) <= LMAX and abs(sum(2.0 ** -l for l in dlen) - 1.0) < 1e-12
    dcodes = canonical(dlen)
    hdr = dynamic_header(litlen, dlen)

    def fixed_bits(h, hd, extra):
        b = 3
        for s, c in enumerate(h):
            b += int(c) * (8 if s < 144 else 9 if s < 256 else 7 if s < 280 else 8)
        return b + extra + 5 * int(hd.sum())
    def table_bits(h, hd, extra):
        return hdr.n + sum(int(c) * litlen[s] for s, c in enumerate(h)) + sum(int(c) * dlen[s] for s, c in enumerate(hd)) + extra
    for name, test in (("config-2 tiles (held out)", test_s), ("reference golden tiles (held out)", test_r)):
        if not test: continue
        fb = np.mean([fixed_bits(h, hd, e) for h, hd, e, _, _ in test]) / 8 / 1000
        tb = np.mean([table_bits(h, hd, e) for h, hd, e, _, _ in test]) / 8 / 1000
        zb = np.mean([len(zlib.compress(r.tobytes(), 6)) for _, _, _, _, r in test]) / 1000
        nm = np.mean([sum(1 for t in toks if t[0] == "match" and t[2] != 1) for _, _, _, toks, _ in test])
        print(f"{name}: fixed code {fb:.1f} kB, this table {tb:.1f} kB ({nm:.0f} hash matches per tile), zlib -6 on the same filtered bytes {zb:.1f} kB")
    # self-check: zlib inflates a stream written with the table to the filtered bytes
    for h, hd, e, toks, rows in (test_s[:1] + test_r[:1]):
        data, nbits = encode_stream(toks, litlen, codes, dlen, dcodes, hdr)
        assert nbits == table_bits(h, hd, e), (nbits, table_bits(h, hd, e))
        got = zlib.decompressobj(-15).decompress(data)
        assert got == rows.tobytes(), "zlib does not inflate the stream to the filtered bytes"
    print(f"header {hdr.n} bits, longest code {max(litlen)} bits, code of 0: {litlen[0]} bits, end of block: {litlen[256]} bits")

    entries = [rev(codes[s], litlen[s]) | (litlen[s] << 16) for s in range(286)]
    dentries = [rev(dcodes[s], dlen[s]) | (dlen[s] << 16) for s in range(30)]
    # the file from byte 40 on: 'T' of "IDAT", the zlib header 78 01, then the block header bits
    head = Bits(); head.put(0x54, 8); head.put(0x78, 8); head.put(0x01, 8); head.put(hdr.acc, hdr.n)
    nwords = (head.n + 31) // 32
    words = [(head.acc >> (32 * k)) & 0xFFFFFFFF for k in range(nwords)]
    with open(os.path.join(ROOT, "osm_renderer_amd", "csrc", "osmt_png_table.h"), "w") as f:
        f.write("/* GENERATED by tools/make_png_huffman.py — do not edit.  The one prefix code of the GPU PNG encoder (a constant\n"
                " * \"dynamic Huffman\" block header, RFC 1951 3.2.7) fitted to map tiles; entry = bit-reversed code | length << 16. */\n")
        f.write("#pragma once\n#include <stdint.h>\n#ifndef PNG_TABLE_QUAL\n#define PNG_TABLE_QUAL static const /* the kernels' file: __device__ __constant__ */\n#endif\n")
        f.write(f"#define PNG_LMAX {LMAX}u /* longest literal/length code: no filtered byte costs more bits than this */\n")
        f.write(f"#define PNG_BLOCK_HDR_BITS {hdr.n}u /* BFINAL, BTYPE and the code description */\n")
        f.write(f"#define PNG_HEAD_WORDS {nwords}u /* words 10 .. of the file: 'T', the zlib header, the block header bits */\n")
        f.write("PNG_TABLE_QUAL uint32_t png_head_words[PNG_HEAD_WORDS] = {" + ", ".join(f"0x{w:08X}u" for w in words) + "};\n")
        f.write("PNG_TABLE_QUAL uint32_t png_code_table[286] = {\n")
        for k in range(0, 286, 8):
            f.write("    " + ", ".join(f"0x{e:08X}u" for e in entries[k:k + 8]) + ",\n")
        f.write("};\n")
        f.write("/* the distance code: entry = bit-reversed code | length << 16 of distance symbol 0 .. 29 (RFC 1951 3.2.5) */\n")
        f.write("PNG_TABLE_QUAL uint32_t png_dist_table[30] = {\n")
        for k in range(0, 30, 8):
            f.write("    " + ", ".join(f"0x{e:08X}u" for e in dentries[k:k + 8]) + ",\n")
        f.write("};\n")
    with open(os.path.join(ROOT, "tests", "golden", "png_huffman.json"), "w") as f:
        json.dump({"what": "the GPU PNG encoder's prefix code (tools/make_png_huffman.py): code lengths (tests/_png_model.py "
                           "derives the canonical codes itself) and the block header bits, LSB first, as a hex integer",
                   "lmax": LMAX, "litlen_lengths": litlen, "dist_lengths": dlen, "block_header_bits": hdr.n,
                   "block_header_hex": "%x" % hdr.acc}, f)
    print("wrote osm_renderer_amd/csrc/osmt_png_table.h and tests/golden/png_huffman.json")


if __name__ == "__main__":
    main()
