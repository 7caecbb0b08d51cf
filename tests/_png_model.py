"""Byte-exact CPU model of the GPU PNG encoder (csrc/osmt_pngenc.hip): ONE deflate block under the constant prefix codes of
tests/golden/png_huffman.json (literal/length and distance), Paeth filter, distance-1 runs and — on the kernel's fast path,
256- or 512-pixel-wide images whose height splits into eight bands — LZ77 matches found through a small hash table per band
(round 6).  Test infrastructure for tests/test_gpu_png_device.py; the model IS the specification of the kernel's choices:

  row            f[0] = 4 (Paeth), f[1 .. 3W] the filtered bytes; NB = 3W + 1
  lane           l = 0 .. 63 owns the bytes [1 + l * NBY, 1 + (l + 1) * NBY), NBY = 3W / 64
  burst          the FIRST byte k of a lane's span with f[k] != 0 and (k == 1 or f[k - 1] == 0), if k + 4 <= NB
  hash           ((f[k] | f[k+1] << 8 | f[k+2] << 16 | f[k+3] << 24) * 2654435761 mod 2^32) >> 24: 256 slots per band, holding
                 1 + (row in band << 11 | k) of the LATEST burst with that hash (rows above only: a row inserts its bursts
                 after it has looked its own up)
  candidate      the slot's burst, if it lies at most 7 rows above (the kernel keeps a ring of eight rows in LDS) and at least
                 4 bytes agree; the match runs to the first differing byte, at most 258 bytes, and stays inside both rows;
                 distance = rows * NB + k - k'
  selection      lanes in order: a candidate is taken if it starts at or behind the end of the last taken one
  tokens         the bytes outside the taken matches are coded as before (literal, distance-1 matches of <= 258 for the rest
                 of a run, <= 2 trailing literals), a run ending where a taken match or the row ends

The canonical codes are derived here from the code lengths (RFC 1951 3.2.2); the block header bits come with the lengths
(they describe them: any inflater — zlib, PIL in the tests — checks that the two agree)."""
import json, os, struct, zlib
import numpy as np

LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_EXTRA = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DIST_BASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DIST_EXTRA = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
LZ_HASH_BITS = 8
LZ_MIN_MATCH = 4
LZ_WINDOW_ROWS = 7
LZ_BANDS = 8  # row bands of a tile (PNG_BANDS: the waves of the kernel's workgroup), each with a table and a history of its own

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png_huffman.json")) as _f:
    _T = json.load(_f)
LITLEN = _T["litlen_lengths"]
DISTLEN = _T["dist_lengths"] if len(_T["dist_lengths"]) == 30 else [5] * 30  # (a table from before round 6: the generator is about to replace it)
LMAX = _T["lmax"]
HDR_BITS, HDR_N = int(_T["block_header_hex"], 16), _T["block_header_bits"]
TOKENS_BIT = 43 * 8 + HDR_N           # PNG_TOKENS_BIT: file bit the first token starts at
HEAD_WORDS = (24 + HDR_N + 31) // 32  # PNG_HEAD_WORDS

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
DCODES = _canonical(DISTLEN)

def lit_token(v):
    return rev(CODES[v], LITLEN[v]), LITLEN[v]

def dist_token(d):
    idx = max(i for i in range(30) if DIST_BASE[i] <= d)
    eb = DIST_EXTRA[idx]; ev = d - DIST_BASE[idx]
    hb, hn = rev(DCODES[idx], DISTLEN[idx]), DISTLEN[idx]
    return hb | (ev << hn), hn + eb

def len_token(L, d=1):
    """length code + extra bits, then the distance code + extra bits"""
    idx = max(i for i in range(29) if LEN_BASE[i] <= L)
    s = 257 + idx; eb = LEN_EXTRA[idx]; ev = L - LEN_BASE[idx]
    hb, hn = rev(CODES[s], LITLEN[s]), LITLEN[s]
    db, dn = dist_token(d)
    return hb | (ev << hn) | (db << (hn + eb)), hn + eb + dn

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

def lz_applies(W, H):
    """the kernel's fast path (k_png_encode_fast): the only one that searches for matches"""
    return W in (256, 512) and H % LZ_BANDS == 0 and H >= LZ_BANDS

def hash4(b0, b1, b2, b3):
    w = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24)
    return ((w * 2654435761) & 0xFFFFFFFF) >> (32 - LZ_HASH_BITS)

def row_matches(row, NB, NBY, r, table, hist):
    """the matches the kernel takes in one row (list of (k, length, distance)), then the row's own insertions"""
    bursts = []
    for l in range(64):
        s0 = 1 + l * NBY
        for k in range(s0, s0 + NBY):
            if row[k] != 0 and (k == 1 or row[k - 1] == 0):
                if k + 4 <= NB:
                    bursts.append(k)
                break
    taken = []
    cover = 0
    hs = []
    for k in bursts:
        h = hash4(row[k], row[k + 1], row[k + 2], row[k + 3])
        hs.append(h)
        v = table[h]
        if not v:
            continue
        cr, ck = (v - 1) >> 11, (v - 1) & 2047
        if r - cr > LZ_WINDOW_ROWS:
            continue
        d = (r - cr) * NB + (k - ck)
        src = hist[cr]
        mx = min(258, NB - k, NB - ck)
        n = 0
        while n < mx and src[ck + n] == row[k + n]:
            n += 1
        if n >= LZ_MIN_MATCH and k >= cover:
            taken.append((k, n, d))
            cover = k + n
    for k, h in zip(bursts, hs):
        table[h] = max(table[h], 1 + ((r << 11) | k))
    return taken

def tile_tokens(rgba, lz=None):
    """(tokens, filtered rows): ('lit', v) and ('match', length, distance) in stream order"""
    H, W, _ = rgba.shape
    f = paeth_filter(rgba[..., :3])
    NB = 3 * W + 1
    use_lz = lz_applies(W, H) if lz is None else lz
    rows = [bytes([4]) + f[y].tobytes() for y in range(H)]
    out = []
    if use_lz:
        NBY = 3 * W // 64
        rpb = H // LZ_BANDS
    for y in range(H):
        row = rows[y]
        taken = []
        if use_lz:
            b, r = divmod(y, rpb)
            if r == 0:
                table = [0] * (1 << LZ_HASH_BITS)
                hist = []
            taken = row_matches(row, NB, NBY, r, table, hist)
            hist.append(row)
        row_toks = [('lit', 4)]
        i = 1; ti = 0
        while i < NB:
            if ti < len(taken) and taken[ti][0] == i:
                row_toks.append(('match', taken[ti][1], taken[ti][2]))
                i += taken[ti][1]; ti += 1
                continue
            v = row[i]
            seg_end = taken[ti][0] if ti < len(taken) else NB
            j = i + 1
            while j < seg_end and row[j] == v: j += 1
            row_toks.append(('lit', v))
            R = j - i - 1
            while R >= 3:
                m = min(R, 258); row_toks.append(('match', m, 1)); R -= m
            row_toks += [('lit', v)] * R
            i = j
        out += row_toks
    return out, rows

def _tok_bits(t):
    return lit_token(t[1])[1] if t[0] == 'lit' else len_token(t[1], t[2])[1]

def encode(rgba, lz=None):
    H, W, _ = rgba.shape
    toks, rows = tile_tokens(rgba, lz=lz)
    bw = Bits(); bw.put(HDR_BITS, HDR_N)
    for t in toks:
        if t[0] == 'lit':
            b, n = lit_token(t[1])
        else:
            b, n = len_token(t[1], t[2])
        bw.put(b, n)
    A, B = 1, 0
    for row in rows:
        stream = np.frombuffer(row, dtype=np.uint8).astype(np.int64)
        nn = len(stream)
        B = (B + nn * A + int(((nn - np.arange(nn)) * stream).sum())) % 65521
        A = (A + int(stream.sum())) % 65521
    t, tn = lit_token(256); bw.put(t, tn)
    deflate = bw.bytes()
    idat = b'\x78\x01' + deflate + struct.pack('>I', (B << 16) | A)
    def chunk(t, d): return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xFFFFFFFF)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, 2, 0, 0, 0)) + chunk(b'IDAT', idat) + chunk(b'IEND', b'')
