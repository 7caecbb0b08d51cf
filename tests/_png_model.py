"""Byte-exact CPU model of the GPU PNG encoder k_png_encode (ONE deflate block under the constant prefix code of
tests/golden/png_huffman.json, distance-1 runs, Paeth filter): test infrastructure for tests/test_gpu_png_device.py.
The canonical codes are derived here from the code lengths (RFC 1951 3.2.2); the block header bits come with the lengths
(they describe them: any inflater — zlib, PIL in the tests — checks that the two agree)."""
import json, os, struct, zlib
import numpy as np

LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_EXTRA = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png_huffman.json")) as _f:
    _T = json.load(_f)
LITLEN = _T["litlen_lengths"]
HDR_BITS, HDR_N = int(_T["block_header_hex"], 16), _T["block_header_bits"]

def rev(code, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (code & 1); code >>= 1
    return r

def _canonical(lengths):
    maxl = max(lengths)
    count = [0] * (maxl + 1)
    for l in lengths:
        if l: count[l] += 1
    code = 0; nxt = [0] * (maxl + 2)
    for bits in range(1, maxl + 1):
        code = (code + count[bits - 1]) << 1
        nxt[bits] = code
    out = []
    for l in lengths:
        out.append(nxt[l] if l else 0)
        if l: nxt[l] += 1
    return out

CODES = _canonical(LITLEN)

def lit_token(v):
    return rev(CODES[v], LITLEN[v]), LITLEN[v]

def len_token(L):
    # the one distance code (distance 1) is a single 0 bit
    idx = max(i for i in range(29) if LEN_BASE[i] <= L)
    s = 257 + idx; eb = LEN_EXTRA[idx]; ev = L - LEN_BASE[idx]
    hb, hn = rev(CODES[s], LITLEN[s]), LITLEN[s]
    return hb | (ev << hn), hn + eb + 1

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

class Bits:
    def __init__(self): self.acc = 0; self.n = 0
    def put(self, v, nb): self.acc |= v << self.n; self.n += nb
    def bytes(self):
        nb = (self.n + 7) // 8
        return self.acc.to_bytes(nb, 'little')

def encode(rgba):
    H, W, _ = rgba.shape
    f = paeth_filter(rgba[..., :3])
    bw = Bits(); bw.put(HDR_BITS, HDR_N)
    A, B = 1, 0
    for y in range(H):
        row = f[y]
        lt, ln = lit_token(4); bw.put(lt, ln)
        n = len(row); i = 0
        while i < n:
            v = int(row[i]); j = i
            while j + 1 < n and row[j + 1] == v: j += 1
            L = j - i + 1
            t, tn = lit_token(v); bw.put(t, tn)
            R = L - 1
            while R >= 3:
                m = min(R, 258); t2, n2 = len_token(m); bw.put(t2, n2); R -= m
            for _ in range(R): bw.put(t, tn)
            i = j + 1
        stream = np.concatenate([[4], row]).astype(np.int64)
        nn = len(stream)
        B = (B + nn * A + int(((nn - np.arange(nn)) * stream).sum())) % 65521
        A = (A + int(stream.sum())) % 65521
    t, tn = lit_token(256); bw.put(t, tn)
    deflate = bw.bytes()
    idat = b'\x78\x01' + deflate + struct.pack('>I', (B << 16) | A)
    def chunk(t, d): return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xFFFFFFFF)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, 2, 0, 0, 0)) + chunk(b'IDAT', idat) + chunk(b'IEND', b'')
