"""Byte-exact CPU model of the GPU PNG encoder k_png_encode (fixed-Huffman deflate, distance-1 runs, Paeth
filter): test infrastructure for tests/test_gpu_png_device.py."""
import struct, zlib, io
import numpy as np

LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_EXTRA = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]

def rev(code, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (code & 1); code >>= 1
    return r

def lit_token(v):
    if v < 144: return rev(0x30 + v, 8), 8
    return rev(0x190 + (v - 144), 9), 9

def len_token(L):
    # distance 1 appended (5 zero bits)
    idx = max(i for i in range(29) if LEN_BASE[i] <= L)
    code = 257 + idx; eb = LEN_EXTRA[idx]; ev = L - LEN_BASE[idx]
    if code <= 279: hb, hn = rev(code - 256, 7), 7
    else: hb, hn = rev(0xC0 + (code - 280), 8), 8
    return hb | (ev << hn), hn + eb + 5

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
    bw = Bits(); bw.put(3, 3)
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
    bw.put(0, 7)
    deflate = bw.bytes()
    idat = b'\x78\x01' + deflate + struct.pack('>I', (B << 16) | A)
    def chunk(t, d): return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xFFFFFFFF)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, 2, 0, 0, 0)) + chunk(b'IDAT', idat) + chunk(b'IEND', b'')

