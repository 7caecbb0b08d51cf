/*
 * osmt_pngenc.hip — PNG files written by the GPU (rgb_triples_to_png, png_writer.rs:4-21).
 * gfx950 only.
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdlib.h>

#include "osmt_internal.h"
#define PNG_TABLE_QUAL static __device__ __constant__
#include "osmt_png_table.h"

/* ------------------------------------------------------------------------- */
/* PNG encoding on the GPU (rgb_triples_to_png, png_writer.rs:4-21): one workgroup per tile
 * turns an RGBA8 framebuffer into a complete RGB8 PNG file — Paeth-filtered rows, ONE deflate block whose only
 * matches are distance-1 runs, Adler-32, chunk CRCs — so that a serving pipeline moves ~45 KB per tile over PCIe
 * instead of 256 KB and the host does no zlib work.  The reference's tests compare decoded pixels only
 * (tests/test_rendering.rs:15-23), so the encoder is free to differ from the png crate.
 * The block is a "dynamic Huffman" one whose code is the SAME for every tile (osmt_png_table.h, fitted to map tiles by
 * tools/make_png_huffman.py: -20 % against the fixed code of RFC 1951, what a per-tile code would give, without a
 * second pass): the 521 header bits are a constant, a token is a table look-up (LDS copy of the table).
 *
 * Per row (3W+1 filtered bytes in LDS, one wave): lanes own contiguous byte spans; a byte starts a run when it differs
 * from its predecessor; a run (v, L) becomes literal(v), matches(len <= 258, dist 1) for the other L-1 bytes,
 * and at most two trailing literals.  Bit counts are prefix-summed across lanes and tokens are OR-ed into an LDS
 * bit buffer; rows are sized first so that every row knows its bit position in the file (see k_png_encode). */
#define PNG_HDR_BYTES 43u /* 8 signature + 25 IHDR + 4 IDAT length + 4 "IDAT" + 2 zlib header */
#define PNG_TOKENS_BIT (PNG_HDR_BYTES * 8u + PNG_BLOCK_HDR_BITS) /* file bit the first token starts at */
#define PNG_HEAD_FULL_WORDS ((PNG_TOKENS_BIT >> 5) - 10u)          /* words 10 .. that hold header bits only */
static_assert(PNG_HEAD_WORDS == PNG_HEAD_FULL_WORDS + ((PNG_TOKENS_BIT & 31u) ? 1u : 0u), "osmt_png_table.h: head words");
static_assert(PNG_LMAX + 5u + PNG_LMAX <= 32u && PNG_LMAX + 13u <= 32u, "a run token, and either half of a match token, fits one png_put");
static_assert(2u * PNG_LMAX + 5u + 13u <= 4u * PNG_LMAX, "a match of four bytes costs no more than four literals: PNG_LMAX bits per filtered byte bound every row");
#define PNG_TAB_WORDS (286u + 30u)
#define PNG_DTAB 286u /* tab[PNG_DTAB + s]: distance symbol s */

/* `tab`: the LDS copy of png_code_table and, behind it, png_dist_table (bit-reversed code | length << 16) */
__device__ __forceinline__ void png_tab_load(uint32_t* tab, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < PNG_TAB_WORDS; i += nthreads) tab[i] = i < 286u ? png_code_table[i] : png_dist_table[i - 286u];
}
__device__ __forceinline__ void png_lit(const uint32_t* tab, uint32_t v, uint32_t& bits, uint32_t& n) {
    const uint32_t e = tab[v];
    bits = e & 0xFFFFu;
    n = e >> 16;
}
/* the length half of a match token: length code + extra bits */
__device__ __forceinline__ void png_len(const uint32_t* tab, uint32_t L, uint32_t& bits, uint32_t& n) {
    uint32_t idx, eb = 0u, ev = 0u;
    if (L == 258u) {
        idx = 28u;
    } else if (L <= 10u) {
        idx = L - 3u;
    } else {
        const uint32_t l = L - 3u;
        eb = (31u - (uint32_t)__clz((int)l)) - 2u;
        idx = 4u + 4u * eb + ((l >> eb) & 3u);
        ev = l & ((1u << eb) - 1u);
    }
    const uint32_t e = tab[257u + idx];
    const uint32_t hn = e >> 16;
    bits = (e & 0xFFFFu) | (ev << hn);
    n = hn + eb;
}
/* the distance half: distance code + extra bits (RFC 1951 3.2.5: symbols 0..3 are the distances 1..4, then two symbols per
 * power of two of d - 1) */
__device__ __forceinline__ void png_dist(const uint32_t* tab, uint32_t d, uint32_t& bits, uint32_t& n) {
    uint32_t idx, eb = 0u, ev = 0u;
    if (d <= 4u) {
        idx = d - 1u;
    } else {
        const uint32_t t = d - 1u;
        const uint32_t hb = 31u - (uint32_t)__clz((int)t);
        eb = hb - 1u;
        idx = 2u * hb + ((t >> eb) & 1u);
        ev = t & ((1u << eb) - 1u);
    }
    const uint32_t e = tab[PNG_DTAB + idx];
    const uint32_t hn = e >> 16;
    bits = (e & 0xFFFFu) | (ev << hn);
    n = hn + eb;
}
/* match of length L (3..258) at distance 1: length code + extra bits + the code of distance symbol 0 */
__device__ __forceinline__ void png_run(const uint32_t* tab, uint32_t L, uint32_t& bits, uint32_t& n) {
    uint32_t idx, eb = 0u, ev = 0u;
    if (L == 258u) {
        idx = 28u;
    } else if (L <= 10u) {
        idx = L - 3u;
    } else {
        const uint32_t l = L - 3u;
        eb = (31u - (uint32_t)__clz((int)l)) - 2u;
        idx = 4u + 4u * eb + ((l >> eb) & 3u);
        ev = l & ((1u << eb) - 1u);
    }
    const uint32_t e = tab[257u + idx];
    const uint32_t hn = e >> 16;
    const uint32_t d0 = tab[PNG_DTAB];
    bits = (e & 0xFFFFu) | (ev << hn) | ((d0 & 0xFFFFu) << (hn + eb));
    n = hn + eb + (d0 >> 16);
}
/* bits of the tokens of run (v, L) */
__device__ __forceinline__ uint32_t png_run_bits(const uint32_t* tab, uint32_t v, uint32_t L) {
    const uint32_t ln = tab[v] >> 16;
    uint32_t total = ln, R = L - 1u;
    const uint32_t n258 = (tab[285] >> 16) + (tab[PNG_DTAB] >> 16); /* code 285: no extra bits */
    while (R >= 258u) { /* at most 11 rounds per 1024-px row (no integer division) */
        total += n258;
        R -= 258u;
    }
    if (R >= 3u) {
        uint32_t b, n;
        png_run(tab, R, b, n);
        total += n;
    } else {
        total += R * ln;
    }
    return total;
}

__device__ __forceinline__ void png_put(uint32_t* buf, uint32_t& pos, uint32_t bits, uint32_t n) {
    const uint32_t w = pos >> 5, sh = pos & 31u;
    atomicOr(buf + w, bits << sh);
    if (sh + n > 32u) atomicOr(buf + w + 1u, bits >> (32u - sh));
    pos += n;
}

#define PNG_MAX_W 1024u
#ifndef OSMT_V_PNG_WAVES
#define OSMT_V_PNG_WAVES 4
#endif
#define PNG_WAVES ((uint32_t)OSMT_V_PNG_WAVES)

/* filtered row y (Paeth, type 4) of the tile into f[0 .. 3W]; executed by one wave */
__device__ __forceinline__ void png_filter_row(const uint8_t* __restrict__ src, uint32_t W, uint32_t y, uint32_t lane, uint8_t* f) {
    const uint32_t* __restrict__ row = reinterpret_cast<const uint32_t*>(src + (size_t)y * W * 4u);
    const uint32_t* __restrict__ up = reinterpret_cast<const uint32_t*>(src + (size_t)(y ? y - 1u : 0u) * W * 4u);
    if (lane == 0) f[0] = 4u;
    for (uint32_t p = lane; p < W; p += 64u) {
        const uint32_t cur = row[p];
        const uint32_t a4 = p ? row[p - 1u] : 0u;
        const uint32_t b4 = y ? up[p] : 0u;
        const uint32_t c4 = (p && y) ? up[p - 1u] : 0u;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int a = (int)((a4 >> (8 * ch)) & 0xFFu), b = (int)((b4 >> (8 * ch)) & 0xFFu), c = (int)((c4 >> (8 * ch)) & 0xFFu);
            const int pp = a + b - c;
            const int pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
            const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            f[1u + 3u * p + (uint32_t)ch] = (uint8_t)((int)((cur >> (8 * ch)) & 0xFFu) - pred);
        }
    }
}

/* One row of the filtered stream, one wave.  EMIT = false: returns the row's bit count (lane-uniform) and its
 * Adler partial sums; EMIT = true: ORs the tokens into `bits` (zeroed, LDS) starting at bit 0. */
template <bool EMIT>
__device__ __forceinline__ uint32_t png_row_tokens(const uint32_t* tab, const uint8_t* f, uint32_t NB, uint32_t lane, uint32_t* bits,
                                                   uint32_t& adler1, uint32_t& adler2) {
    const uint32_t span = (NB - 1u + 63u) / 64u;
    const uint32_t s0 = min(NB, 1u + lane * span), s1 = min(NB, s0 + span);
    uint32_t first_start = 0xFFFFFFFFu;
    uint32_t a1 = 0u, a2 = 0u;
    for (uint32_t k = s0; k < s1; ++k) {
        const uint32_t v = f[k];
        if (first_start == 0xFFFFFFFFu && (k == 1u || v != f[k - 1u])) first_start = k;
        if (!EMIT) {
            a1 += v;
            a2 += (NB - k) * v;
        }
    }
    const unsigned long long has = __ballot(first_start != 0xFFFFFFFFu);
    const unsigned long long later = lane < 63u ? (has >> (lane + 1u)) : 0ull;
    const int nxt_lane = later ? (int)lane + 1 + __builtin_ctzll(later) : (int)lane;
    const uint32_t nxt_pos_raw = (uint32_t)__shfl((int)first_start, nxt_lane);
    const uint32_t nxt_pos = later ? nxt_pos_raw : NB; /* where the run that leaves this span ends */
    uint32_t my_bits = lane == 0 ? (tab[4] >> 16) : 0u; /* the filter-type byte: literal(4) */
    for (uint32_t k = s0; k < s1;) {
        const uint32_t v = f[k];
        const bool is_start = k == 1u || v != f[k - 1u];
        uint32_t e = k + 1u;
        while (e < s1 && f[e] == v) ++e;
        if (is_start) my_bits += png_run_bits(tab, v, ((e == s1) ? nxt_pos : e) - k);
        k = e;
    }
    uint32_t incl = my_bits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if ((int)lane >= d) incl += t;
    }
    const uint32_t row_bits = (uint32_t)__shfl((int)incl, 63);
    if (!EMIT) {
        if (lane == 0) {
            a1 += 4u;
            a2 += NB * 4u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            a1 += (uint32_t)__shfl_xor((int)a1, d);
            a2 += (uint32_t)__shfl_xor((int)a2, d);
        }
        adler1 = a1;
        adler2 = a2;
        return row_bits;
    }
    uint32_t pos = incl - my_bits;
    if (lane == 0) {
        uint32_t b, n;
        png_lit(tab, 4u, b, n);
        png_put(bits, pos, b, n);
    }
    for (uint32_t k = s0; k < s1;) {
        const uint32_t v = f[k];
        const bool is_start = k == 1u || v != f[k - 1u];
        uint32_t e = k + 1u;
        while (e < s1 && f[e] == v) ++e;
        if (is_start) {
            const uint32_t end = (e == s1) ? nxt_pos : e;
            uint32_t lb, ln;
            png_lit(tab, v, lb, ln);
            png_put(bits, pos, lb, ln);
            uint32_t R = end - k - 1u;
            while (R >= 3u) {
                const uint32_t m = min(R, 258u);
                uint32_t b, n;
                png_run(tab, m, b, n);
                png_put(bits, pos, b, n);
                R -= m;
            }
            for (; R; --R) png_put(bits, pos, lb, ln);
        }
        k = e;
    }
    return row_bits;
}

/* One workgroup (4 waves) per tile.  Pass 1: every wave sizes its rows (bits + Adler sums); a scan gives each
 * row its bit position in the file; pass 2: every wave re-filters its rows, builds the row's bits in LDS and
 * stores them shifted to that position — interior words plainly, the first and last word of a row (shared with
 * its neighbours) with atomicOr into words zeroed between the passes. */
__global__ __launch_bounds__(64 * PNG_WAVES) void k_png_encode(const uint8_t* __restrict__ g_rgba, size_t tile_stride, uint32_t n_tiles,
                                                              uint32_t W, uint32_t H, uint32_t ihdr_crc, uint8_t* g_out,
                                                              size_t out_stride, uint32_t* __restrict__ g_len) {
    __shared__ uint8_t sh_f[PNG_WAVES][3u * PNG_MAX_W + 4u];
    __shared__ uint32_t sh_bits[PNG_WAVES][(PNG_LMAX * (3u * PNG_MAX_W + 1u)) / 32u + 4u];
    __shared__ uint32_t sh_tab[PNG_TAB_WORDS];
    __shared__ uint32_t sh_rowpos[PNG_MAX_W + 1u]; /* pass 1: bits of row y; after the scan: its absolute bit position */
    __shared__ uint32_t sh_a1[PNG_MAX_W], sh_a2[PNG_MAX_W];
    __shared__ uint32_t sh_crc_tab[256];
    __shared__ uint32_t sh_col[32];
    __shared__ uint32_t sh_raw[64 * PNG_WAVES];
    __shared__ uint32_t sh_adler;
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint8_t* __restrict__ src = g_rgba + (size_t)tile * tile_stride;
    uint8_t* out = g_out + (size_t)tile * out_stride;
    uint32_t* out_w = reinterpret_cast<uint32_t*>(out);
    const uint32_t NB = 3u * W + 1u; /* bytes of one filtered row incl. the filter-type byte */

    for (uint32_t i = tid; i < 256u; i += 64u * PNG_WAVES) { /* CRC-32 (reflected 0xEDB88320) byte table */
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        sh_crc_tab[i] = c;
    }
    png_tab_load(sh_tab, tid, 64u * PNG_WAVES);
    __syncthreads();
    /* ---- pass 1: size every row ---- */
    for (uint32_t y = wave; y < H; y += PNG_WAVES) {
        png_filter_row(src, W, y, lane, sh_f[wave]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t a1, a2;
        const uint32_t rb = png_row_tokens<false>(sh_tab, sh_f[wave], NB, lane, nullptr, a1, a2);
        if (lane == 0) {
            sh_rowpos[y] = rb;
            sh_a1[y] = a1;
            sh_a2[y] = a2;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    /* ---- row positions (bit 0 of the deflate stream = byte 43; the block header comes first) + Adler-32 ---- */
    if (tid == 0) {
        uint32_t pos = PNG_TOKENS_BIT;
        uint32_t A = 1u, B = 0u;
        for (uint32_t y = 0; y < H; ++y) {
            const uint32_t rb = sh_rowpos[y];
            sh_rowpos[y] = pos;
            pos += rb;
            B = (uint32_t)(((unsigned long long)B + (unsigned long long)NB * A + sh_a2[y]) % 65521ull);
            A = (A + sh_a1[y]) % 65521u;
        }
        sh_rowpos[H] = pos; /* end-of-block code goes here */
        sh_adler = (B << 16) | A;
        /* signature, IHDR, IDAT length placeholder, "IDAT", zlib header (0x78 0x01), block header bits 1,1,0 */
        out_w[0] = 0x474E5089u;
        out_w[1] = 0x0A1A0A0Du;
        out_w[2] = 0x0D000000u;
        out_w[3] = 0x52444849u;
        out_w[4] = __builtin_bswap32(W);
        out_w[5] = __builtin_bswap32(H);
        out_w[6] = 0x00000208u;
        out_w[7] = (ihdr_crc >> 24 << 8) | (((ihdr_crc >> 16) & 0xFFu) << 16) | (((ihdr_crc >> 8) & 0xFFu) << 24);
        out_w[8] = (ihdr_crc & 0xFFu);
        out_w[9] = 0x41444900u;
    }
    __syncthreads();
    /* words shared by two rows (and the word the stream ends in) start from zero; words 10 .. carry 'T', the zlib
     * header and the block header (the last of them may be where row 0 starts) */
    for (uint32_t y = tid; y <= H; y += 64u * PNG_WAVES) {
        const uint32_t w = sh_rowpos[y] >> 5;
        if (w >= 10u + PNG_HEAD_WORDS) out_w[w] = 0u;
        if (y == H) out_w[w + 1u] = 0u; /* the end-of-block code may spill into the next word */
    }
    if (tid < PNG_HEAD_WORDS) out_w[10u + tid] = png_head_words[tid];
    __threadfence_block();
    __syncthreads();
    /* ---- pass 2: emit ---- */
    const uint32_t nwords_row = (PNG_LMAX * NB) / 32u + 2u;
    for (uint32_t y = wave; y < H; y += PNG_WAVES) {
        png_filter_row(src, W, y, lane, sh_f[wave]);
        for (uint32_t i = lane; i < nwords_row; i += 64u) sh_bits[wave][i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t a1, a2;
        const uint32_t row_bits = png_row_tokens<true>(sh_tab, sh_f[wave], NB, lane, sh_bits[wave], a1, a2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t gbit = sh_rowpos[y];
        const uint32_t sh = gbit & 31u, wb = gbit >> 5;
        const uint32_t n_out = ((gbit + row_bits - 1u) >> 5) - wb + 1u; /* words holding bits of this row */
        for (uint32_t k = lane; k < n_out; k += 64u) {
            uint32_t w = sh ? (sh_bits[wave][k] << sh) : sh_bits[wave][k];
            if (k && sh) w |= sh_bits[wave][k - 1u] >> (32u - sh);
            /* the row's first word, and its last one unless the row ends exactly on a word boundary, are shared
             * with the neighbouring rows (pre-zeroed above); everything else is this row's alone */
            const bool shared = k == 0u || (k + 1u == n_out && ((gbit + row_bits) & 31u) != 0u);
            if (shared)
                atomicOr(out_w + wb + k, w);
            else
                out_w[wb + k] = w;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __threadfence_block();
    __syncthreads();
    /* end of block, pad to a byte, Adler-32, IDAT length */
    const uint32_t eob = sh_tab[256];
    const uint32_t gend = sh_rowpos[H] + (eob >> 16);
    const uint32_t endb = (gend + 7u) >> 3; /* first byte after the deflate stream */
    if (tid == 0) {
        { /* the words the code lands in were zeroed above; the last row's bits are in (barrier) */
            const uint32_t pos = sh_rowpos[H], w = pos >> 5, sh = pos & 31u, eb = eob & 0xFFFFu;
            out_w[w] |= eb << sh;
            if (sh + (eob >> 16) > 32u) out_w[w + 1u] |= eb >> (32u - sh);
        }
        const uint32_t adler = sh_adler;
        out[endb + 0u] = (uint8_t)(adler >> 24);
        out[endb + 1u] = (uint8_t)(adler >> 16);
        out[endb + 2u] = (uint8_t)(adler >> 8);
        out[endb + 3u] = (uint8_t)adler;
        const uint32_t idat_len = 2u + (endb - PNG_HDR_BYTES) + 4u;
        out[33] = (uint8_t)(idat_len >> 24);
        out[34] = (uint8_t)(idat_len >> 16);
        out[35] = (uint8_t)(idat_len >> 8);
        out[36] = (uint8_t)idat_len;
    }
    __threadfence_block();
    __syncthreads();
    /* CRC-32 of "IDAT" + data = bytes [37, endb + 4): per-thread raw CRCs of equal blocks, then combined */
    const uint32_t NT = 64u * PNG_WAVES;
    const uint32_t c0 = 37u, c1 = endb + 4u;
    const uint32_t blk = (c1 - c0 + NT - 1u) / NT;
    {
        const uint32_t b0 = min(c1, c0 + tid * blk), b1 = min(c1, b0 + blk);
        uint32_t s = 0u;
        for (uint32_t k = b0; k < b1; ++k) s = sh_crc_tab[(s ^ out[k]) & 0xFFu] ^ (s >> 8);
        sh_raw[tid] = s;
        if (tid < 32u) { /* column `tid` of the operator "advance the CRC register over blk zero bytes" */
            uint32_t c = 1u << tid;
            for (uint32_t k = 0; k < blk; ++k) c = sh_crc_tab[c & 0xFFu] ^ (c >> 8);
            sh_col[tid] = c;
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t s = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < NT; ++i) {
            const uint32_t b0 = min(c1, c0 + i * blk), b1 = min(c1, b0 + blk);
            const uint32_t len = b1 - b0;
            if (!len) break;
            if (len == blk) {
                uint32_t t = 0u;
                for (uint32_t b = 0; b < 32u; ++b)
                    if ((s >> b) & 1u) t ^= sh_col[b];
                s = t;
            } else {
                for (uint32_t k = 0; k < len; ++k) s = sh_crc_tab[s & 0xFFu] ^ (s >> 8);
            }
            s ^= sh_raw[i];
        }
        const uint32_t crc = ~s;
        uint32_t o = c1;
        out[o++] = (uint8_t)(crc >> 24);
        out[o++] = (uint8_t)(crc >> 16);
        out[o++] = (uint8_t)(crc >> 8);
        out[o++] = (uint8_t)crc;
        const uint8_t iend[12] = {0, 0, 0, 0, 0x49, 0x45, 0x4E, 0x44, 0xAE, 0x42, 0x60, 0x82};
        for (int k = 0; k < 12; ++k) out[o++] = iend[k];
        g_len[tile] = o;
    }
}

/* Fast path of k_png_encode for W = 64 * PX (PX = 4: 256-px tiles, PX = 8: 512): ONE tokenisation pass.
 * Wave w owns the band of rows [w*H/PNG_BANDS, (w+1)*H/PNG_BANDS) and walks it top to bottom; a lane owns PX consecutive pixels,
 * whose raw values, the row above (carried in registers from the previous iteration) and the 3*PX filtered bytes
 * all live in registers — run starts, run ends inside the lane and token sizes are straight-line code, the next
 * row's pixels are fetched while the current one is tokenised.  Band 0 appends its rows directly behind the
 * file header; bands 1.. append into staging areas further up the tile's slot (bit 0 of a word), and once the
 * band lengths are known they are moved down, bit-shifted, behind their predecessors (dst <= src, chunked
 * read-then-write).  Adler-32 per band, combined like zlib's adler32_combine. */
/* Round 6: LZ77 matches beyond the distance-1 runs (tests/_png_model.py is the specification, byte for byte).  79 % of a map
 * tile's filtered bytes are zeros (runs already), the rest are short bursts of anti-aliasing residuals that cost ~8 bits each —
 * and the same burst, zeros and next burst come back a few rows further down wherever a line keeps its slope.  Per band, all in
 * LDS: a hash table of 256 slots (the 4 bytes at a burst's start -> the latest burst with that hash in the rows above) and a
 * ring of the band's last eight filtered rows (a window of seven rows above buys all but 0.7 % of what the whole band would:
 * 44.5 against 44.2 kB per config-2 tile); per row: every lane looks its first burst up, the candidates are extended 16 bytes
 * per step, taken greedily in lane order, and the tokeniser treats a taken match as one token and its end as a run start.
 * ~1300 matches per config-2 tile: 48.2 -> 44.5 kB (zlib -6 on the same bytes: 42.2).  (First version of the round: the whole
 * band's history in the tile's output slot — global stores and loads behind a fence per row; at agent scope that fence wrote the
 * XCD's L2 back once per row: 7.9 ms per 1024 tiles instead of 1.7.) */
#define PNG_LZ_SLOTS 256u /* (1024 slots: 44.51 kB per config-2 tile, 256: 44.74 — and a fourth workgroup fits a CU's LDS) */
#define PNG_LZ_HASH_SHIFT 24u
#define PNG_LZ_RING 8u /* rows kept: the current one and the seven above it */
#define PNG_LZ_MIN 4u
/* row bands of a tile = waves of its workgroup (round 6: eight, was four: a tile's latency halves — it is what a chunk of
 * the PNG call, or a single-tile request, waits for — and a full batch still fills the machine, two workgroups per CU) */
#define PNG_BANDS 8u
#define PNG_NT (64u * PNG_BANDS)
template <int PX>
__global__ __launch_bounds__(PNG_NT) __attribute__((amdgpu_waves_per_eu(PX == 4 ? 4 : 2, PX == 4 ? 4 : 2))) void k_png_encode_fast(const uint8_t* __restrict__ g_rgba, size_t tile_stride, uint32_t n_tiles,
                                                         uint32_t H, uint32_t ihdr_crc, uint8_t* g_out, size_t out_stride,
                                                         uint32_t band_cap_words, uint32_t* __restrict__ g_len, uint32_t lz_enable) {
    constexpr uint32_t W = 64u * PX, NB = 3u * W + 1u, NBY = 3u * PX; /* bytes per lane */
    constexpr uint32_t ROWW = (PNG_LMAX * NB) / 32u + 3u;
    constexpr uint32_t ROWBW = (NB + 3u + 20u + 3u) / 4u; /* a filtered row in LDS: byte k at byte k + 3 (a lane's span is whole dwords); a compare reads 20 bytes ahead */
    static_assert(NB - 1u < 2048u && NBY % 4u == 0u && NBY <= 24u, "slot packing, lane spans");
    __shared__ uint32_t sh_lz_tab[PNG_BANDS][PNG_LZ_SLOTS];
    __shared__ uint32_t sh_hist[PNG_BANDS][PNG_LZ_RING][ROWBW];
    __shared__ uint32_t sh_bits[PNG_BANDS][ROWW];
    __shared__ uint32_t sh_tab[PNG_TAB_WORDS];
    __shared__ uint32_t sh_crc_tab[256];
    __shared__ uint32_t sh_col[32];
    __shared__ uint32_t sh_raw[PNG_NT];
    __shared__ uint32_t sh_carry[PNG_BANDS];
    __shared__ uint32_t sh_band_bits[PNG_BANDS], sh_band_a[PNG_BANDS], sh_band_b[PNG_BANDS];
    __shared__ uint32_t sh_move[1];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint8_t* __restrict__ src = g_rgba + (size_t)tile * tile_stride;
    uint8_t* out = g_out + (size_t)tile * out_stride;
    uint32_t* out_w = reinterpret_cast<uint32_t*>(out);
    for (uint32_t i = tid; i < 256u; i += PNG_NT) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        sh_crc_tab[i] = c;
    }
    png_tab_load(sh_tab, tid, PNG_NT);
    for (uint32_t i = tid; i < PNG_BANDS * PNG_LZ_SLOTS; i += PNG_NT) (&sh_lz_tab[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t rows = H / PNG_BANDS, y_begin = wave * rows, y_end = y_begin + rows;
    /* where this band's bits go while it is being produced */
    const uint32_t stage_w = wave == 0u ? 0u : 11u + PNG_HEAD_WORDS + wave * band_cap_words; /* band 0: the file itself */
    const bool lz_on = lz_enable != 0u;
    uint32_t gbit = wave == 0u ? PNG_TOKENS_BIT : 0u;                        /* bit cursor relative to out_w[stage_w] */
    /* band 0 continues the last word of the block header */
    uint32_t carry = (wave == 0u && (PNG_TOKENS_BIT & 31u)) ? png_head_words[PNG_HEAD_WORDS - 1u] : 0u;
    if (tid < PNG_HEAD_FULL_WORDS) out_w[10u + tid] = png_head_words[tid]; /* 'T', zlib header, block header */
    if (tid == 0) {
        out_w[0] = 0x474E5089u;
        out_w[1] = 0x0A1A0A0Du;
        out_w[2] = 0x0D000000u;
        out_w[3] = 0x52444849u;
        out_w[4] = __builtin_bswap32(W);
        out_w[5] = __builtin_bswap32(H);
        out_w[6] = 0x00000208u;
        out_w[7] = (ihdr_crc >> 24 << 8) | (((ihdr_crc >> 16) & 0xFFu) << 16) | (((ihdr_crc >> 8) & 0xFFu) << 24);
        out_w[8] = (ihdr_crc & 0xFFu);
        out_w[9] = 0x41444900u;
    }
    uint32_t adler_a = 1u, adler_b = 0u;
    uint32_t cur[PX], prev[PX], nxt[PX];
    {
        const uint4* __restrict__ r = reinterpret_cast<const uint4*>(src + (size_t)y_begin * W * 4u) + lane * (PX / 4);
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
            const uint4 v = r[q];
            cur[4 * q] = v.x, cur[4 * q + 1] = v.y, cur[4 * q + 2] = v.z, cur[4 * q + 3] = v.w;
        }
        if (y_begin) {
            const uint4* __restrict__ u = reinterpret_cast<const uint4*>(src + (size_t)(y_begin - 1u) * W * 4u) + lane * (PX / 4);
#pragma unroll
            for (int q = 0; q < PX / 4; ++q) {
                const uint4 v = u[q];
                prev[4 * q] = v.x, prev[4 * q + 1] = v.y, prev[4 * q + 2] = v.z, prev[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < PX; ++j) prev[j] = 0u;
        }
    }
    const uint32_t base = 1u + lane * NBY; /* stream index of this lane's first filtered byte */
    for (uint32_t y = y_begin; y < y_end; ++y) {
        if (y + 1u < y_end) { /* fetch the next row now */
            const uint4* __restrict__ r = reinterpret_cast<const uint4*>(src + (size_t)(y + 1u) * W * 4u) + lane * (PX / 4);
#pragma unroll
            for (int q = 0; q < PX / 4; ++q) {
                const uint4 v = r[q];
                nxt[4 * q] = v.x, nxt[4 * q + 1] = v.y, nxt[4 * q + 2] = v.z, nxt[4 * q + 3] = v.w;
            }
        }
        /* ---- Paeth filter, bytes in registers ---- */
        uint32_t fb[NBY];
        {
            uint32_t la = (uint32_t)__shfl_up((int)cur[PX - 1], 1), lc = (uint32_t)__shfl_up((int)prev[PX - 1], 1);
            if (lane == 0) la = lc = 0u;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const uint32_t a4 = j ? cur[j - 1] : la, c4 = j ? prev[j - 1] : lc, b4 = prev[j], x4 = cur[j];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const int a = (int)((a4 >> (8 * ch)) & 0xFFu), b = (int)((b4 >> (8 * ch)) & 0xFFu), c = (int)((c4 >> (8 * ch)) & 0xFFu);
                    const int pp = a + b - c;
                    const int pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
                    const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    fb[3 * j + ch] = (uint32_t)((int)((x4 >> (8 * ch)) & 0xFFu) - pred) & 0xFFu;
                }
            }
        }
        /* ---- run starts and, for each byte, the next start inside the lane ---- */
        const uint32_t pbyte = (uint32_t)__shfl_up((int)fb[NBY - 1], 1);
        uint32_t startmask = 0u;
#pragma unroll
        for (int i = 0; i < (int)NBY; ++i) {
            const bool st = i ? (fb[i] != fb[i - 1]) : (lane == 0u || fb[0] != pbyte);
            startmask |= (st ? 1u : 0u) << i;
        }
        /* ---- matches (see above): `eff` = where a literal + run token starts, `bound` = where a run ends inside the lane ---- */
        uint32_t eff = startmask, bound = startmask;
        bool m_sel = false;
        uint32_t m_i = 0u, m_len = 0u, m_dist = 0u;
        if (lz_on) {
            const uint32_t r = y - y_begin;
            uint32_t* const rowb = sh_hist[wave][r & (PNG_LZ_RING - 1u)]; /* takes the place of the row eight above */
            uint32_t* const tab = sh_lz_tab[wave];
#pragma unroll
            for (int q = 0; q < (int)NBY / 4; ++q)
                rowb[1u + lane * (NBY / 4u) + (uint32_t)q] = fb[4 * q] | (fb[4 * q + 1] << 8) | (fb[4 * q + 2] << 16) | (fb[4 * q + 3] << 24);
            if (lane == 0u) rowb[0] = 0x04000000u; /* the filter-type byte at byte 3 */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            /* the lane's first burst: a non-zero byte behind a zero (or at the start of the row) */
            uint32_t burst = 0u;
#pragma unroll
            for (int i = 0; i < (int)NBY; ++i) {
                const bool pz = i ? (fb[i - 1] == 0u) : (lane == 0u || pbyte == 0u);
                burst |= ((fb[i] != 0u && pz) ? 1u : 0u) << i;
            }
            const uint32_t kb = burst ? (uint32_t)__builtin_ctz(burst) : 0u;
            const uint32_t k = base + kb;
            const bool has_b = burst != 0u && k + 4u <= NB;
            uint32_t hsh = 0u, cand = 0u;
            if (has_b) {
                const uint32_t a = k + 3u;
                const uint32_t w4 = __builtin_amdgcn_alignbyte(rowb[(a >> 2) + 1u], rowb[a >> 2], a & 3u);
                hsh = (w4 * 2654435761u) >> PNG_LZ_HASH_SHIFT;
                cand = tab[hsh];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier(); /* every lane has looked up before any lane inserts: a row never matches itself */
            if (has_b) atomicMax(&tab[hsh], 1u + ((r << 11) | k));
            uint32_t n = 0u, mx = 0u, ck = 0u, cr = 0u;
            bool going = false;
            if (cand) {
                cr = (cand - 1u) >> 11;
                ck = (cand - 1u) & 2047u;
                if (r - cr < PNG_LZ_RING) { /* still in the ring (the distance is below 8 rows: far inside deflate's 32 KiB) */
                    m_dist = (r - cr) * NB + (k - ck);
                    mx = min(258u, min(NB - k, NB - ck));
                    going = true;
                }
            }
            const uint32_t* const hsrc = sh_hist[wave][cr & (PNG_LZ_RING - 1u)];
            while (__ballot(going)) {
                if (going) {
                    const uint32_t ca = k + 3u + n, ha = ck + 3u + n;
                    const uint32_t* cp = rowb + (ca >> 2);
                    const uint32_t* hp = hsrc + (ha >> 2);
                    uint32_t cw[5], hw[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        cw[j] = cp[j];
                        hw[j] = hp[j];
                    }
                    uint32_t d = 16u;
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const uint32_t x = __builtin_amdgcn_alignbyte(cw[j + 1], cw[j], ca & 3u) ^ __builtin_amdgcn_alignbyte(hw[j + 1], hw[j], ha & 3u);
                        if (x) d = 4u * (uint32_t)j + ((uint32_t)__builtin_ctz(x) >> 3);
                    }
                    n += d;
                    if (d < 16u || n >= mx) going = false;
                }
            }
            n = min(n, mx);
            /* taken greedily, lanes (= positions) in order */
            unsigned long long cb = __ballot(n >= PNG_LZ_MIN);
            uint32_t cover = 0u;
            while (cb) {
                const uint32_t l = (uint32_t)__builtin_ctzll(cb);
                cb &= cb - 1ull;
                const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)k, (int)l), nn = (uint32_t)__builtin_amdgcn_readlane((int)n, (int)l);
                if (kk >= cover) {
                    if (lane == l) m_sel = true;
                    cover = kk + nn;
                }
            }
            /* where the last taken match of the lanes below ends */
            const unsigned long long below = __ballot(m_sel) & ((1ull << lane) - 1ull);
            const uint32_t endv = m_sel ? k + n : 0u;
            const uint32_t e2 = (uint32_t)__shfl((int)endv, below ? 63 - (int)__builtin_clzll(below) : 0);
            const uint32_t in_end = below ? e2 : 0u;
            /* the first byte behind the cover that comes in from the left, relative to this lane's span */
            const int32_t rel = below ? (int32_t)in_end - (int32_t)base : -1;
            const uint32_t in_cov = (uint32_t)min(max(rel, 0), (int32_t)NBY);
            uint32_t covered = (1u << in_cov) - 1u, forced = 0u;
            if (rel >= 0 && rel < (int32_t)NBY) forced |= 1u << (uint32_t)rel;
            if (m_sel) {
                m_i = kb;
                m_len = n;
                const uint32_t me = min(kb + n, NBY);
                covered |= ((1u << me) - 1u) & ~((1u << kb) - 1u);
                if (kb + n < NBY) forced |= 1u << (kb + n);
            }
            eff = (startmask | forced) & ~covered;
            bound = eff | (m_sel ? 1u << m_i : 0u);
        }
        const uint32_t first_start = bound ? base + (uint32_t)__builtin_ctz(bound) : 0xFFFFFFFFu;
        const unsigned long long has = __ballot(bound != 0u);
        const unsigned long long later = lane < 63u ? (has >> (lane + 1u)) : 0ull;
        const int nxt_lane = later ? (int)lane + 1 + __builtin_ctzll(later) : (int)lane;
        const uint32_t nxt_pos_raw = (uint32_t)__shfl((int)first_start, nxt_lane);
        const uint32_t nxt_pos = later ? nxt_pos_raw : NB;
        /* ---- size, prefix, emit ---- */
        uint32_t my_bits = lane == 0u ? (sh_tab[4] >> 16) : 0u, a1 = lane == 0u ? 4u : 0u, a2 = lane == 0u ? NB * 4u : 0u;
#pragma unroll
        for (int i = 0; i < (int)NBY; ++i) {
            a1 += fb[i];
            a2 += (NB - (base + (uint32_t)i)) * fb[i];
            if ((eff >> i) & 1u) {
                const uint32_t rest = bound >> (i + 1); /* i + 1 < 32 always: NBY <= 24 */
                const uint32_t end = rest ? base + (uint32_t)i + 1u + (uint32_t)__builtin_ctz(rest) : nxt_pos;
                my_bits += png_run_bits(sh_tab, fb[i], end - (base + (uint32_t)i));
            }
        }
        uint32_t ml_b = 0u, ml_n = 0u, md_b = 0u, md_n = 0u; /* the taken match's two halves */
        if (m_sel) {
            png_len(sh_tab, m_len, ml_b, ml_n);
            png_dist(sh_tab, m_dist, md_b, md_n);
            my_bits += ml_n + md_n;
        }
        uint32_t incl = my_bits;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
            if ((int)lane >= d) incl += t;
        }
        const uint32_t row_bits = (uint32_t)__shfl((int)incl, 63);
        uint32_t* bits = sh_bits[wave];
        for (uint32_t i = lane; i < ROWW; i += 64u) bits[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t pos = incl - my_bits;
        if (lane == 0u) {
            uint32_t b, n;
            png_lit(sh_tab, 4u, b, n);
            png_put(bits, pos, b, n);
        }
#pragma unroll
        for (int i = 0; i < (int)NBY; ++i) {
            if (m_sel && m_i == (uint32_t)i) {
                png_put(bits, pos, ml_b, ml_n);
                png_put(bits, pos, md_b, md_n);
            }
            if ((eff >> i) & 1u) {
                const uint32_t rest = bound >> (i + 1);
                const uint32_t end = rest ? base + (uint32_t)i + 1u + (uint32_t)__builtin_ctz(rest) : nxt_pos;
                uint32_t lb, ln;
                png_lit(sh_tab, fb[i], lb, ln);
                png_put(bits, pos, lb, ln);
                uint32_t R = end - (base + (uint32_t)i) - 1u;
                while (R >= 3u) {
                    const uint32_t m = min(R, 258u);
                    uint32_t b, n;
                    png_run(sh_tab, m, b, n);
                    png_put(bits, pos, b, n);
                    R -= m;
                }
                for (; R; --R) png_put(bits, pos, lb, ln);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            a1 += (uint32_t)__shfl_xor((int)a1, d);
            a2 += (uint32_t)__shfl_xor((int)a2, d);
        }
        adler_b = (uint32_t)(((unsigned long long)adler_b + (unsigned long long)NB * adler_a + a2) % 65521ull);
        adler_a = (adler_a + a1) % 65521u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        /* ---- append the row at the band's bit cursor ---- */
        {
            const uint32_t sh = gbit & 31u, wb = gbit >> 5;
            const uint32_t gend = gbit + row_bits;
            const uint32_t n_out = (gend >> 5) - wb + 1u; /* words touched; the last one is the new carry */
            for (uint32_t k = lane; k < n_out; k += 64u) {
                uint32_t w = sh ? (bits[k] << sh) : bits[k];
                if (k)
                    w |= sh ? (bits[k - 1u] >> (32u - sh)) : 0u;
                else
                    w |= carry;
                if (k + 1u < n_out)
                    out_w[stage_w + wb + k] = w;
                else
                    sh_carry[wave] = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            gbit = gend;
            carry = (gbit & 31u) ? sh_carry[wave] : 0u;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            prev[j] = cur[j];
            cur[j] = nxt[j];
        }
    }
    /* flush the band's partial word (upper bits zero) and publish its length and checksum */
    if (lane == 0u) {
        if (gbit & 31u) out_w[stage_w + (gbit >> 5)] = carry;
        sh_band_bits[wave] = wave == 0u ? gbit - PNG_TOKENS_BIT : gbit;
        sh_band_a[wave] = adler_a;
        sh_band_b[wave] = adler_b;
    }
    __threadfence_block();
    __syncthreads();
    /* ---- move bands 1.. down behind their predecessors ---- */
    uint32_t endpos = PNG_TOKENS_BIT + sh_band_bits[0];
    for (uint32_t b = 1; b < PNG_BANDS; ++b) {
        const uint32_t L = sh_band_bits[b];
        const uint32_t sw = 11u + PNG_HEAD_WORDS + b * band_cap_words; /* staging: bit 0 of out_w[sw] */
        const uint32_t sh = endpos & 31u, wb = endpos >> 5;
        const uint32_t n_src = (L + 31u) >> 5;
        const uint32_t n_dst = ((endpos + L + 31u) >> 5) - wb; /* destination words holding bits of this band */
        for (uint32_t c = 0; c < n_dst; c += PNG_NT) {
            const uint32_t k = c + tid;
            uint32_t w = 0u;
            if (k < n_dst) {
                const uint32_t s_cur = k < n_src ? out_w[sw + k] : 0u;
                const uint32_t s_prev = (k && k - 1u < n_src) ? out_w[sw + k - 1u] : 0u;
                w = sh ? ((s_cur << sh) | (s_prev >> (32u - sh))) : s_cur;
                if (k == 0u && sh) w |= out_w[wb]; /* the predecessor's partial last word */
            }
            __syncthreads(); /* every source word of this chunk is read before any destination word is written */
            if (k < n_dst) out_w[wb + k] = w;
            __threadfence_block();
            __syncthreads();
        }
        endpos += L;
    }
    /* end of block (the stream's last word has zeros above its bits; the code may spill into the next one) */
    const uint32_t eob = sh_tab[256];
    if (tid == 0) {
        const uint32_t w = endpos >> 5, sh = endpos & 31u, eb = eob & 0xFFFFu;
        if (sh)
            out_w[w] |= eb << sh;
        else
            out_w[w] = eb;
        if (sh + (eob >> 16) > 32u) out_w[w + 1u] = eb >> (32u - sh);
        /* Adler-32 of the concatenation (zlib's adler32_combine): A = A1 + A2 - 1, B = B1 + B2 + len2 * (A1 - 1) */
        unsigned long long A = sh_band_a[0], B = sh_band_b[0];
        const unsigned long long len2 = (unsigned long long)rows * NB;
        for (uint32_t b = 1; b < PNG_BANDS; ++b) {
            const unsigned long long A2 = sh_band_a[b], B2 = sh_band_b[b];
            B = (B + B2 + (len2 % 65521ull) * ((A + 65520ull) % 65521ull)) % 65521ull;
            A = (A + A2 + 65520ull) % 65521ull;
        }
        sh_move[0] = (uint32_t)((B << 16) | A);
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t gend = endpos + (eob >> 16);
    const uint32_t endb = (gend + 7u) >> 3;
    if (tid == 0) {
        const uint32_t adler = sh_move[0];
        out[endb + 0u] = (uint8_t)(adler >> 24);
        out[endb + 1u] = (uint8_t)(adler >> 16);
        out[endb + 2u] = (uint8_t)(adler >> 8);
        out[endb + 3u] = (uint8_t)adler;
        const uint32_t idat_len = 2u + (endb - PNG_HDR_BYTES) + 4u;
        out[33] = (uint8_t)(idat_len >> 24);
        out[34] = (uint8_t)(idat_len >> 16);
        out[35] = (uint8_t)(idat_len >> 8);
        out[36] = (uint8_t)idat_len;
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t c0 = 37u, c1 = endb + 4u;
    const uint32_t blk = (c1 - c0 + PNG_NT - 1u) / PNG_NT;
    {
        const uint32_t b0 = min(c1, c0 + tid * blk), b1 = min(c1, b0 + blk);
        uint32_t s = 0u;
        for (uint32_t k = b0; k < b1; ++k) s = sh_crc_tab[(s ^ out[k]) & 0xFFu] ^ (s >> 8);
        sh_raw[tid] = s;
        if (tid < 32u) {
            uint32_t c = 1u << tid;
            for (uint32_t k = 0; k < blk; ++k) c = sh_crc_tab[c & 0xFFu] ^ (c >> 8);
            sh_col[tid] = c;
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t s = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < PNG_NT; ++i) {
            const uint32_t b0 = min(c1, c0 + i * blk), b1 = min(c1, b0 + blk);
            const uint32_t len = b1 - b0;
            if (!len) break;
            if (len == blk) {
                uint32_t t = 0u;
                for (uint32_t b = 0; b < 32u; ++b)
                    if ((s >> b) & 1u) t ^= sh_col[b];
                s = t;
            } else {
                for (uint32_t k = 0; k < len; ++k) s = sh_crc_tab[s & 0xFFu] ^ (s >> 8);
            }
            s ^= sh_raw[i];
        }
        const uint32_t crc = ~s;
        uint32_t o = c1;
        out[o++] = (uint8_t)(crc >> 24);
        out[o++] = (uint8_t)(crc >> 16);
        out[o++] = (uint8_t)(crc >> 8);
        out[o++] = (uint8_t)crc;
        const uint8_t iend[12] = {0, 0, 0, 0, 0x49, 0x45, 0x4E, 0x44, 0xAE, 0x42, 0x60, 0x82};
        for (int k = 0; k < 12; ++k) out[o++] = iend[k];
        g_len[tile] = o;
    }
}

/* gathers the variable-length PNG files of a batch into one blob: tile i -> blob[off[i] .. off[i] + len[i]) */
__global__ __launch_bounds__(256) void k_png_compact(const uint8_t* __restrict__ slots, size_t slot_stride,
                                                     const uint32_t* __restrict__ len, const unsigned long long* __restrict__ off,
                                                     uint32_t n, uint8_t* __restrict__ blob) {
    const uint32_t tile = blockIdx.x;
    if (tile >= n) return;
    const uint8_t* __restrict__ src = slots + (size_t)tile * slot_stride;
    uint8_t* __restrict__ dst = blob + off[tile];
    const uint32_t L = len[tile];
    for (uint32_t i = threadIdx.x; i < L; i += 256u) dst[i] = src[i];
}

hipError_t osmt_launch_png_compact(const void* slots, size_t slot_stride, const uint32_t* len, const unsigned long long* off, uint32_t n,
                                   void* blob, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_png_compact, dim3(n), dim3(256), 0, st, reinterpret_cast<const uint8_t*>(slots), slot_stride, len, off, n,
                       reinterpret_cast<uint8_t*>(blob));
    return hipGetLastError();
}

hipError_t osmt_launch_png(const void* rgba, size_t tile_stride, uint32_t n, uint32_t W, uint32_t H, uint32_t ihdr_crc, void* out,
                           size_t out_stride, uint32_t* out_len, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (W > PNG_MAX_W) return hipErrorInvalidValue;
    /* OSMT_PNG_LZ=0: runs only, as before round 6 (bigger files, a shorter kernel): for measurements */
    static const uint32_t lz_enable = [] {
        const char* v = getenv("OSMT_PNG_LZ");
        return (uint32_t)((v && v[0] == '0') ? 0 : 1);
    }();
    if ((W == 256u || W == 512u) && (H % PNG_BANDS) == 0u && H >= PNG_BANDS) {
        /* staging capacity of one band: H / PNG_BANDS rows of at most PNG_LMAX bits per filtered byte */
        const uint32_t band_cap_words = (uint32_t)(((size_t)(H / PNG_BANDS) * (3u * W + 1u) * PNG_LMAX + 31u) / 32u + 2u);
        if ((size_t)(11u + PNG_HEAD_WORDS + PNG_BANDS * band_cap_words) * 4u + 64u <= out_stride) {
            if (W == 256u)
                hipLaunchKernelGGL((k_png_encode_fast<4>), dim3(n), dim3(PNG_NT), 0, st, reinterpret_cast<const uint8_t*>(rgba), tile_stride, n, H,
                                   ihdr_crc, reinterpret_cast<uint8_t*>(out), out_stride, band_cap_words, out_len, lz_enable);
            else
                hipLaunchKernelGGL((k_png_encode_fast<8>), dim3(n), dim3(PNG_NT), 0, st, reinterpret_cast<const uint8_t*>(rgba), tile_stride, n, H,
                                   ihdr_crc, reinterpret_cast<uint8_t*>(out), out_stride, band_cap_words, out_len, lz_enable);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(k_png_encode, dim3(n), dim3(64 * PNG_WAVES), 0, st, reinterpret_cast<const uint8_t*>(rgba), tile_stride, n, W, H, ihdr_crc,
                       reinterpret_cast<uint8_t*>(out), out_stride, out_len);
    return hipGetLastError();
}
