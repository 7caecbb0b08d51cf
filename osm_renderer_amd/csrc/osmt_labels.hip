/*
 * osmt_labels.hip — the label pass on the GPU: glyph coverage (font/rasterizer.rs), icon blits
 * (labeler.rs:91-106) and label collisions (tile_pixels.rs:131-162).  The survivors are blended by k_raster<LABELS>
 * (osmt_kernels.hip).  gfx950 only; -ffp-contract=off (the reference never fuses a*b+c).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "osmt_internal.h"

/* ------------------------------------------------------------------------- */
/* Label pass: font/rasterizer.rs + tile_pixels.rs:131-162 + labeler.rs:91-106.
 *
 * k_label_cover    one workgroup of TWO waves per band of a label's window (osmt_label_band: the stripes of a band
 *                  never share a cell with another band, and the list is ordered longest-first).  The per-key f64
 *                  sums of the reference (BTreeMap entry += ..., :77,:80) do not associate, so every cell must see
 *                  its addends in draw_line call order; everything else is free:
 *                    producer wave  lane = draw_line call, 64 at a time: the call's whole arithmetic (:27-80, two f64
 *                                   divisions) for its stripes in the band; each resulting sum is PARKED in the list
 *                                   of its CHANNEL (column parity, A/S kind, stripe parity) in LDS, lists compacted;
 *                    consumer wave  lane = channel: a cell belongs to exactly one channel, so the eight lists are
 *                                   independent chains walked side by side in call order, the running cell in a
 *                                   register, into the LDS-resident A / S accumulators (no atomics, no test for
 *                                   "same cell as the last sum": write back, read the next cell, add);
 *                  double-buffered hand-over (one barrier per batch).  Calls whose cells would collide in a channel
 *                  (three stripes tall, three cells wide) are replayed stripe by stripe by the consumer
 *                  (lane = stripe) at their place in the order.  Then
 *                  lane = stripe runs save_to_figure's scan over [x_min, x_max] (:121-143) and the band goes out
 *                  coalesced: total = min(a + s_acc, 1.0) per cell (0 where the stripe has no key) for k_raster, one
 *                  bit per cell (total > 0) for k_label_resolve.
 * k_label_resolve  one wave per tile, labels strictly in draw order: a label succeeds iff none of
 *                  the pixels it would set (icon rectangle, then cells with total > 0) inside labels_bb
 *                  belongs to an earlier SUCCEEDED label (set_label_pixel, tile_pixels.rs:131-148;
 *                  pixels of failed labels are overwritten freely); succeeded labels mark their pixels
 *                  in a (3W)^2-bit ownership map, 32 pixels per operation.  The early `return false` of draw_icon /
 *                  save_to_figure only skips pixels of a label that is not blended anyway.
 * k_raster<LABELS> blends the succeeded labels over the area canvas before to_rgb_triples. */
/* draw_line for stripe y (font/rasterizer.rs:46-80) into the stripe's own accumulator rows */
__device__ __forceinline__ bool label_stripe(const osmt_label_seg& sg, int32_t y, int32_t cx0, uint32_t cols, double* a_row,
                                             double* s_row, int32_t& x_min, int32_t& x_max) {
    const double x0 = sg.x0, y0 = sg.y0, slope = sg.slope, recip = sg.slope_recip, sign = sg.sign;
    const double y_bottom = fmax((double)y, sg.y_min);
    const double y_top = fmin((double)(y + 1), sg.y_max);
    const double y_delta = y_top - y_bottom;
    const double x_at_bottom = x0 + (y_bottom - y0) * slope;
    const double x_at_top = x0 + (y_top - y0) * slope;
    const bool flip_edge = !(x_at_bottom <= x_at_top);
    const double x_smallest = flip_edge ? x_at_top : x_at_bottom;
    const double x_largest = flip_edge ? x_at_bottom : x_at_top;
    const int32_t x_to = (int32_t)floor(x_largest);
    const int32_t x_from = (int32_t)floor(x_smallest);
    if (x_from < cx0 || x_to + 1 >= cx0 + (int32_t)cols) return false; /* cannot happen: the window is conservative */
    for (int32_t x = x_from; x <= x_to; ++x) {
        const double x_left = fmax((double)x, x_smallest);
        const double x_next = (double)(x + 1);
        const double x_right = fmin(x_next, x_largest);
        double pixel_area = (x_next - x_right) * y_delta;
        const double trapezoid_width = x_right - x_left;
        if (trapezoid_width > 0.0) {
            const double y_at_left = y0 + (x_left - x0) * recip;
            const double y_at_right = y0 + (x_right - x0) * recip;
            const double trapezoid_height = flip_edge ? (y_top - y_at_left) + (y_top - y_at_right)
                                                      : (y_at_left - y_bottom) + (y_at_right - y_bottom);
            pixel_area += trapezoid_width * trapezoid_height / 2.0;
        }
        a_row[x - cx0] += sign * pixel_area;
    }
    s_row[x_to + 1 - cx0] += sign * y_delta;
    x_min = min(x_min, x_from);
    x_max = max(x_max, x_to + 1);
    return true;
}

/* the y-independent part of draw_line (font/rasterizer.rs:27-41) */
__device__ __forceinline__ osmt_label_seg label_seg_prep(const double4 q) {
    const double x0 = q.x, y0 = q.y, x1 = q.z, y1 = q.w;
    osmt_label_seg r;
    const double delta = y1 - y0;
    r.x0 = x0;
    r.y0 = y0;
    r.sign = (y0 <= y1) ? 1.0 : -1.0;
    r.slope = (x1 - x0) / delta;
    r.slope_recip = 1.0 / r.slope;
    r.y_min = fmin(y0, y1);
    r.y_max = fmax(y0, y1);
    if (delta == 0.0) {
        r.yf = 1;
        r.yl = 0;
    } else {
        r.yf = (int32_t)floor(r.y_min);
        r.yl = (int32_t)floor(r.y_max);
    }
    return r;
}

#define LC_CELLS OSMT_LABEL_LDS_CELLS

/* A draw_line call parks its sums in LDS, one per CHANNEL = (column parity, A/S kind, stripe parity): the cells
 * one short call touches always fall into different channels, and a given cell always falls into the same one, so
 * every channel is a chain of sums that no other channel's cells take part in.  Calls whose cells collide in a
 * channel (three cells wide, three stripes tall, ...) are replayed stripe by stripe by the row owners instead. */
#define LC_CH 8
/* a parked sum names its cell by its byte offset in the band's accumulator array: A cells first, then the S cells, then
 * one cell that nobody reads (what a channel holds before its first sum and after a replay) */
static_assert((2 * LC_CELLS + 1) * 8 < 65536, "a parked sum's cell offset is a uint16");
#define LC_TRASH (2u * LC_CELLS * 8u)

/* channel list strides, skewed so that the eight channel lanes reading entry j of their lists hit different banks */
#define LC_VSTRIDE 65 /* doubles */
#define LC_KSTRIDE 66 /* uint16 */

/* what the producer wave hands to the consumer wave with one batch of 64 calls */
struct LcBatch {
    unsigned long long mine[LC_CH]; /* per channel: the calls that parked a sum in it */
    unsigned long long rest, slowm; /* calls that cross the band; those of them that have to be replayed */
    uint32_t base;                  /* index of the batch's first call */
    uint32_t done;                  /* no more batches */
};

__device__ __forceinline__ unsigned long long first_lane_u64(unsigned long long v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

/* one wave's LDS traffic stays in program order; this only keeps the compiler from moving it */
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(128) void k_label_cover(const osmt_labelinfo* __restrict__ g_lab,
                                                     const osmt_label_band* __restrict__ g_band, uint32_t n_bands,
                                                     const double4* __restrict__ g_seg, double* __restrict__ g_a,
                                                     unsigned long long* __restrict__ g_bits, uint32_t* g_err) {
    __shared__ double sh_acc[2 * LC_CELLS + 1];
    double* const sh_a = sh_acc;
    double* const sh_s = sh_acc + LC_CELLS;
    __shared__ double sh_ev_val[2][LC_CH * LC_VSTRIDE]; /* [buffer][channel][call of the batch]: the parked sum */
    __shared__ uint16_t sh_ev_key[2][LC_CH * LC_KSTRIDE]; /* its cell: byte offset in sh_acc */
    __shared__ LcBatch sh_batch[2];
    __shared__ uint32_t sh_cmin[64], sh_cmax[64]; /* per stripe: columns of its keys */
    if (blockIdx.x >= n_bands) return;
    const osmt_labelinfo* __restrict__ li = g_lab + g_band[blockIdx.x].label;
    const uint32_t lane = threadIdx.x & 63u;
    const bool producer = threadIdx.x < 64u;
    const int32_t ry0 = li->ry0, cx0 = li->cx0;
    const uint32_t R = (uint32_t)(li->ry1 - ry0 + 1), cols = li->cols;
    const uint32_t n_segs = li->n_segs;
    const double4* __restrict__ segs = g_seg + li->seg_off;
    double* __restrict__ A = g_a + li->plane_off;
    const uint32_t band_rows = min(64u, LC_CELLS / cols);
    const uint32_t rbase = g_band[blockIdx.x].rbase;
    const uint32_t nrow = min(band_rows, R - rbase);
    const uint32_t cnt = nrow * cols;
    const int32_t band0 = ry0 + (int32_t)rbase, band1 = band0 + (int32_t)nrow - 1;
    bool oob = false;
    for (uint32_t i = threadIdx.x; i < cnt; i += 128u) {
        sh_a[i] = 0.0;
        sh_s[i] = 0.0;
    }
    if (producer) {
        sh_cmin[lane] = 0xFFFFFFFFu;
        sh_cmax[lane] = 0u;
    }
    __syncthreads();
    if (producer) {
        /* ---- lane = draw_line call: all the f64 work of the call's stripes inside the band, parked per channel ---- */
        uint32_t k = 0; /* batches handed over so far */
        double4 seg_next = n_segs > lane ? segs[lane] : make_double4(0.0, 0.0, 0.0, 0.0);
        for (uint32_t base = 0; base < n_segs; base += 64u) {
            const uint32_t i = base + lane;
            const double4 seg_cur = seg_next;
            if (i + 64u < n_segs) seg_next = segs[i + 64u]; /* in flight while this batch is worked on */
            bool overlaps = false, slow = false;
            uint32_t chmask = 0u;
            if (i < n_segs) /* draw_line returns at once for delta == 0 (font/rasterizer.rs:30-32) */
                overlaps = seg_cur.y != seg_cur.w && (int32_t)floor(fmax(seg_cur.y, seg_cur.w)) >= band0 &&
                           (int32_t)floor(fmin(seg_cur.y, seg_cur.w)) <= band1;
            const unsigned long long rest = __ballot(overlaps);
            if (!rest) continue; /* 64 calls of glyphs in other bands: no division spent on them */
            double* const ev_val = sh_ev_val[k & 1u];
            uint16_t* const ev_key = sh_ev_key[k & 1u];
            if (overlaps) {
                const osmt_label_seg sg = label_seg_prep(seg_cur);
                const int32_t ya = max(sg.yf, band0), yb = min(sg.yl, band1);
                /* two stripes (different stripe parity) of at most two A cells (different column parity) and one S
                 * cell never meet in a channel; anything taller or wider is replayed */
                slow = yb - ya >= 2;
                auto emit = [&](uint32_t kind, uint32_t row, uint32_t col, double val) {
                    const uint32_t ch = (col & 1u) | (kind << 1) | ((row & 1u) << 2);
                    chmask |= 1u << ch;
                    ev_key[ch * LC_KSTRIDE + lane] = (uint16_t)((kind * LC_CELLS + row * cols + col) * 8u);
                    ev_val[ch * LC_VSTRIDE + lane] = val;
                };
                for (int32_t yy = ya; yy <= yb && !slow; ++yy) {
                    /* font/rasterizer.rs:46-80 for stripe yy */
                    const double y_bottom = fmax((double)yy, sg.y_min);
                    const double y_top = fmin((double)(yy + 1), sg.y_max);
                    const double y_delta = y_top - y_bottom;
                    const double x_at_bottom = sg.x0 + (y_bottom - sg.y0) * sg.slope;
                    const double x_at_top = sg.x0 + (y_top - sg.y0) * sg.slope;
                    const bool flip_edge = !(x_at_bottom <= x_at_top);
                    const double x_smallest = flip_edge ? x_at_top : x_at_bottom;
                    const double x_largest = flip_edge ? x_at_bottom : x_at_top;
                    const int32_t x_to = (int32_t)floor(x_largest);
                    const int32_t x_from = (int32_t)floor(x_smallest);
                    if (x_from < cx0 || x_to + 1 >= cx0 + (int32_t)cols) { /* cannot happen: the window is conservative */
                        oob = true;
                        continue;
                    }
                    if (x_to - x_from >= 2) { /* three cells in one stripe share a channel: replay */
                        slow = true;
                        break;
                    }
                    const uint32_t row = (uint32_t)(yy - band0);
                    /* the stripe's keys span [x_from, x_to + 1] (save_to_figure's x_min / x_max, :115-120); a call
                     * that turns out to be replayed has the same span, so adding it twice is harmless */
                    atomicMin(&sh_cmin[row], (uint32_t)(x_from - cx0));
                    atomicMax(&sh_cmax[row], (uint32_t)(x_to + 1 - cx0));
                    for (int32_t x = x_from; x <= x_to; ++x) {
                        const double x_left = fmax((double)x, x_smallest);
                        const double x_next = (double)(x + 1);
                        const double x_right = fmin(x_next, x_largest);
                        double pixel_area = (x_next - x_right) * y_delta;
                        const double trapezoid_width = x_right - x_left;
                        if (trapezoid_width > 0.0) {
                            const double y_at_left = sg.y0 + (x_left - sg.x0) * sg.slope_recip;
                            const double y_at_right = sg.y0 + (x_right - sg.x0) * sg.slope_recip;
                            const double trapezoid_height = flip_edge ? (y_top - y_at_left) + (y_top - y_at_right)
                                                                      : (y_at_left - y_bottom) + (y_at_right - y_bottom);
                            pixel_area += trapezoid_width * trapezoid_height / 2.0;
                        }
                        emit(0u, row, (uint32_t)(x - cx0), sg.sign * pixel_area);
                    }
                    emit(1u, row, (uint32_t)(x_to + 1 - cx0), sg.sign * y_delta);
                }
            }
            LcBatch* const hb = &sh_batch[k & 1u];
            const unsigned long long slowm = __ballot(overlaps && slow);
            /* every channel's parked sums move to the front of its list (in place: the wave reads before it writes,
             * and a sum never moves up), so that the consumer walks them without looking for the next call */
            if (!overlaps || slow) chmask = 0u;
            uint32_t mine_lo = 0u, mine_hi = 0u; /* lane c: channel c's calls */
/* lane c of (mine_lo, mine_hi) takes the ballot.  gfx950 wants two wait states between the v_cmp that writes the mask
 * and a VALU read of it as a scalar operand; the compiler does not look into inline assembly, hence the s_nop */
#define LC_MINE(c)                                                              \
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" \
        : "+v"(mine_lo), "+v"(mine_hi)                                          \
        : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "n"(c));
#define LC_COMPACT(c)                                                                                                  \
    {                                                                                                                  \
        const bool has = (chmask & (1u << (c))) != 0u;                                                                 \
        const unsigned long long m = __ballot(has);                                                                    \
        LC_MINE(c)                                                                                                     \
        if (has) { /* one wave: the reads of all lanes are done before the first write lands */                        \
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); \
            const uint16_t key = ev_key[(c) * LC_KSTRIDE + lane];                                                      \
            const double val = ev_val[(c) * LC_VSTRIDE + lane];                                                        \
            wave_lds_order();                                                                                          \
            ev_key[(c) * LC_KSTRIDE + rank] = key;                                                                     \
            ev_val[(c) * LC_VSTRIDE + rank] = val;                                                                     \
        }                                                                                                              \
        wave_lds_order();                                                                                              \
    }
            LC_COMPACT(0) LC_COMPACT(1) LC_COMPACT(2) LC_COMPACT(3) LC_COMPACT(4) LC_COMPACT(5) LC_COMPACT(6) LC_COMPACT(7)
#undef LC_COMPACT
#undef LC_MINE
            if (lane < LC_CH) hb->mine[lane] = ((unsigned long long)mine_hi << 32) | mine_lo;
            if (lane == 0u) {
                hb->rest = rest;
                hb->slowm = slowm;
                hb->base = base;
                hb->done = 0u;
            }
            __syncthreads(); /* hand-over k: the consumer takes this buffer, the producer moves on to the other one */
            ++k;
        }
        if (lane == 0u) sh_batch[k & 1u].done = 1u;
        __syncthreads();
    } else {
        /* ---- lane = channel (the first LC_CH lanes): the parked sums are applied strictly in call order.  Every
         * cell belongs to exactly one channel, so the channels' sums are independent chains that advance side by
         * side; a lane keeps the running sum of the cell it is on in a register and puts it back before it reads the
         * cell of the next sum (the same cell again, most of the time: that costs less than testing for it). ---- */
        const bool active = lane < nrow;
        const int32_t y = ry0 + (int32_t)(rbase + lane);
        double* a_row = sh_a + (active ? lane * cols : 0u);
        double* s_row = sh_s + (active ? lane * cols : 0u);
        uint32_t ccell = LC_TRASH; /* the cell this channel is adding to (byte offset), its running sum in cval */
        auto cell = [&](uint32_t off) -> double* { return reinterpret_cast<double*>(reinterpret_cast<char*>(sh_acc) + off); };
        double cval = 0.0;
        uint32_t c_min = 0xFFFFFFFFu, c_max = 0u; /* columns of the stripe's keys added by replayed calls (x - cx0) */
        for (uint32_t k = 0;; ++k) {
            __syncthreads(); /* hand-over k */
            const LcBatch* const hb = &sh_batch[k & 1u];
            if (__builtin_amdgcn_readfirstlane((int)hb->done)) break;
            const double* const my_val = sh_ev_val[k & 1u] + (lane < LC_CH ? lane * LC_VSTRIDE : 0u);
            const uint16_t* const my_key = sh_ev_key[k & 1u] + (lane < LC_CH ? lane * LC_KSTRIDE : 0u);
            unsigned long long rest = first_lane_u64(hb->rest);
            const unsigned long long slowm = first_lane_u64(hb->slowm);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb->base);
            const unsigned long long mine = lane < LC_CH ? hb->mine[lane] : 0ull;
            uint32_t pos = 0; /* sums of my channel applied so far */
            while (rest) {
                /* calls before the next replayed one: their sums go through the channels */
                const unsigned long long sl_rest = slowm & rest;
                const uint32_t sl = sl_rest ? (uint32_t)__builtin_ctzll(sl_rest) : 64u;
                const unsigned long long seg = sl < 64u ? (rest & ((1ull << sl) - 1ull)) : rest;
                const uint32_t end = pos + (uint32_t)__popcll(mine & seg);
                while (pos < end) {
                    /* no test for "same cell as before": the sum goes back to its cell and the next cell is read,
                     * in this order — when the channel stays on a cell that reads back what was just written */
                    const uint32_t K = my_key[pos];
                    const double v = my_val[pos];
                    ++pos;
                    *cell(ccell) = cval;
                    wave_lds_order();
                    cval = *cell(K) + v;
                    ccell = K;
                }
                if (sl >= 64u) break;
                { /* the replayed call works on LDS directly: write the cached cells back first */
                    const osmt_label_seg q = label_seg_prep(segs[base + sl]);
                    *cell(ccell) = cval;
                    ccell = LC_TRASH;
                    wave_lds_order();
                    if (active && y >= q.yf && y <= q.yl) {
                        int32_t x_min = INT32_MAX, x_max = INT32_MIN;
                        oob |= !label_stripe(q, y, cx0, cols, a_row, s_row, x_min, x_max);
                        if (x_min <= x_max) {
                            c_min = min(c_min, (uint32_t)(x_min - cx0));
                            c_max = max(c_max, (uint32_t)(x_max - cx0));
                        }
                    }
                    wave_lds_order();
                }
                rest &= ~((2ull << sl) - 1ull);
            }
        }
        *cell(ccell) = cval;
        wave_lds_order();
        c_min = min(c_min, sh_cmin[lane]);
        c_max = max(c_max, sh_cmax[lane]);
        /* save_to_figure (:115-147) for this stripe: keys span [c_min, c_max]; the rest of the row stays 0 */
        if (active && c_min <= c_max) {
            double s_acc = 0.0;
            for (uint32_t c = c_min; c <= c_max; ++c) {
                s_acc += s_row[c];
                a_row[c] = fmin(a_row[c] + s_acc, 1.0);
            }
        }
    }
    __syncthreads();
    /* the band goes out coalesced: the totals for k_raster, and one bit per cell (total > 0: the pixels the label
     * would set, rasterizer.rs:137) for k_label_resolve */
    double* __restrict__ dst = A + (size_t)rbase * cols;
    unsigned long long* __restrict__ bits = g_bits + li->wide_off + (size_t)(rbase / band_rows) * osmt_label_band_words(cols);
    for (uint32_t i0 = (threadIdx.x & ~63u); i0 < cnt; i0 += 128u) {
        const uint32_t i = i0 + lane;
        const double v = i < cnt ? sh_a[i] : 0.0;
        if (i < cnt) dst[i] = v;
        const unsigned long long m = __ballot(v > 0.0);
        if (lane == 0u) bits[i0 >> 6] = m;
    }
    if (oob) atomicOr(g_err, 1u);
}

/* Windows wider than LC_CELLS columns (a glyph far to the side of labels_bb in a stripe that crosses it):
 * the same walk with the accumulator rows in global memory. */
__global__ __launch_bounds__(64) void k_label_cover_wide(const osmt_labelinfo* __restrict__ g_lab, const uint32_t* __restrict__ g_wide,
                                                         uint32_t n_wide, const double4* __restrict__ g_seg, double* g_a, double* g_s,
                                                         uint32_t* g_err) {
    if (blockIdx.x >= n_wide) return;
    const osmt_labelinfo* __restrict__ li = g_lab + g_wide[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    const int32_t ry0 = li->ry0, cx0 = li->cx0;
    const uint32_t R = (uint32_t)(li->ry1 - ry0 + 1), cols = li->cols;
    const uint32_t n_segs = li->n_segs;
    const double4* __restrict__ segs = g_seg + li->seg_off;
    double* A = g_a + li->plane_off;
    double* S = g_s + li->wide_off;
    bool oob = false;
    for (uint32_t rbase = 0; rbase < R; rbase += 64u) {
        const uint32_t nrow = min(64u, R - rbase);
        {
            const size_t cnt = (size_t)nrow * cols;
            for (size_t i = lane; i < cnt; i += 64u) {
                A[(size_t)rbase * cols + i] = 0.0;
                S[i] = 0.0;
            }
        }
        __syncthreads(); /* one wave per block: orders the zeroing before the row owners' read-modify-writes */
        const bool active = lane < nrow;
        const int32_t y = ry0 + (int32_t)(rbase + lane);
        double* a_row = A + (size_t)(rbase + (active ? lane : 0u)) * cols;
        double* s_row = S + (size_t)(active ? lane : 0u) * cols;
        int32_t x_min = INT32_MAX, x_max = INT32_MIN;
        for (uint32_t si = 0; si < n_segs; ++si) {
            const osmt_label_seg sg = label_seg_prep(segs[si]);
            if (!active || y < sg.yf || y > sg.yl) continue;
            oob |= !label_stripe(sg, y, cx0, cols, a_row, s_row, x_min, x_max);
        }
        if (active && x_min <= x_max) {
            double s_acc = 0.0;
            for (int32_t x = x_min; x <= x_max; ++x) {
                s_acc += s_row[x - cx0];
                a_row[x - cx0] = fmin(a_row[x - cx0] + s_acc, 1.0);
            }
        }
        __syncthreads();
    }
    if (oob) atomicOr(g_err, 1u);
}


/* i / d and i % d from d's float reciprocal; `small`: i < 2^22, so that (float)i is exact and the product is within
 * one of the quotient (relative error 2^-23): one step either way settles it.  Otherwise the plain division. */
__device__ __forceinline__ uint32_t resolve_divmod(bool small, uint32_t i, uint32_t d, float rcp, uint32_t& rem) {
    if (!small) {
        rem = i % d;
        return i / d;
    }
    uint32_t q = (uint32_t)((float)i * rcp);
    int32_t r = (int32_t)(i - q * d);
    if (r < 0) {
        --q;
        r += (int32_t)d;
    } else if (r >= (int32_t)d) {
        ++q;
        r -= (int32_t)d;
    }
    rem = (uint32_t)r;
    return q;
}
/* LDS_BM: the (3W)^2-bit ownership map lives in LDS (scale 1: 72 KB); otherwise in global memory. */
/* ONE WAVE per tile: a label's pixels are handled 32 at a time — a row of an icon is a run of ones, a row of a text
 * window is a piece of the band's coverage bit stream (k_label_cover) shifted onto the ownership map — so a label is a
 * few dozen word operations and no workgroup barrier stands between two labels.  The labels are a serial chain (a
 * verdict decides what the next label collides with) and only two such waves fit a CU beside their 72 KB maps, so
 * nothing else would hide a trip to memory: the tile's label records are staged in LDS 64 at a time, and the stream
 * words of label l + 1 are requested before label l is decided.  Windows wider than the LDS band
 * (k_label_cover_wide, no bit stream) are walked cell by cell. */
#define LR_STAGE 64u
template <bool LDS_BM>
__global__ __launch_bounds__(64) void k_label_resolve(
    const osmt_labelinfo* __restrict__ g_lab, const uint32_t* __restrict__ g_job_label_off, uint32_t n_jobs, uint32_t scale,
    const double* __restrict__ g_a, const unsigned long long* __restrict__ g_bits, uint32_t* g_bitmap, uint8_t* g_ok,
    osmt_tile_label* __restrict__ g_tl, uint32_t* __restrict__ g_tl_cnt) {
    extern __shared__ uint32_t sh_bm[];
    __shared__ osmt_labelinfo sh_li[LR_STAGE];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_jobs) return;
    const uint32_t lane = threadIdx.x;
    const int32_t W = (int32_t)(OSMT_TILE_SIZE * scale);
    const int32_t EW = 3 * W; /* labels_bb is the 3x3-tile square [-W, 2W) (tile_pixels.rs:67-72); a multiple of 32 */
    const uint32_t row_words = (uint32_t)EW / 32u;
    const size_t words = (size_t)EW * row_words;
    uint32_t* bm = LDS_BM ? sh_bm : g_bitmap + (size_t)tile * words;
    const uint32_t l0 = g_job_label_off[tile], l1 = g_job_label_off[tile + 1];
    auto stage = [&](uint32_t first) { /* records first .. first + 63 -> LDS, one per lane */
        if (first + lane < l1) {
            const uint4* src = reinterpret_cast<const uint4*>(g_lab + first + lane);
            uint4* dst = reinterpret_cast<uint4*>(&sh_li[lane]);
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[k] = src[k];
        }
    };
    stage(l0);
    if (LDS_BM) {
        uint4* z = reinterpret_cast<uint4*>(sh_bm); /* EW * EW / 32 words: a multiple of 4 */
        for (size_t i = lane; i < words / 4u; i += 64u) z[i] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (size_t i = lane; i < words; i += 64u) bm[i] = 0u;
        __threadfence();
    }
    wave_lds_order();
    /* pass 0: does the word collide with an earlier succeeded label?  pass 1: take ownership.  Words of one row run
     * never repeat; an icon and a text of the same label may meet in a word, hence the atomic OR. */
    auto word_op = [&](int pass, uint32_t wi, uint32_t m, bool& hit) {
        if (!m) return;
        if (pass == 0) {
            const uint32_t have = LDS_BM ? bm[wi] : __hip_atomic_load(bm + wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hit |= (have & m) != 0u;
        } else {
            atomicOr(bm + wi, m);
        }
    };
    /* what a label is, in map words */
    struct Plan {
        int32_t ix0, iy0, ry0, ry1, cx0;
        uint32_t iw, ih, cols;
        bool has_cells, streamed;
        int32_t ia, ib, ta, tb, iya; /* clipped icon / text column runs, first icon row inside the map */
        uint32_t iw0, inw, irows, tw0, tnw, R, band_rows, band_words;
        const uint32_t* stream;
        const double* A;
    };
    auto plan_of = [&](uint32_t l) -> Plan {
        const osmt_labelinfo* li = &sh_li[(l - l0) % LR_STAGE];
        Plan p;
        p.ix0 = li->icon_x, p.iy0 = li->icon_y, p.ry0 = li->ry0, p.ry1 = li->ry1, p.cx0 = li->cx0;
        p.iw = li->icon_w, p.ih = li->icon_h, p.cols = li->cols;
        p.has_cells = li->has_text && p.ry0 <= p.ry1 && p.cols > 0;
        /* a row run [dx0, dx1) of map columns, clipped to the map (set_label_pixel: outside labels_bb -> true,
         * nothing to own): words w0 .. w0 + nw - 1 of the row */
        auto clip_run = [&](int32_t dx0, int32_t dx1, int32_t& a, int32_t& b, uint32_t& w0, uint32_t& nw) {
            a = max(dx0, 0), b = min(dx1, EW);
            w0 = (uint32_t)a >> 5;
            nw = a < b ? (((uint32_t)(b - 1) >> 5) - w0 + 1u) : 0u;
        };
        clip_run(p.ix0 + W, p.ix0 + W + (int32_t)p.iw, p.ia, p.ib, p.iw0, p.inw);
        p.iya = max(p.iy0, -W);
        const int32_t iyb = min(p.iy0 + (int32_t)p.ih, 2 * W);
        p.irows = (p.iw && p.iya < iyb) ? (uint32_t)(iyb - p.iya) : 0u;
        clip_run(p.cx0 + W, p.cx0 + W + (int32_t)p.cols, p.ta, p.tb, p.tw0, p.tnw);
        p.streamed = p.has_cells && p.cols <= LC_CELLS;
        p.R = p.has_cells ? (uint32_t)(p.ry1 - p.ry0 + 1) : 0u; /* rows are clipped to the map already */
        p.band_rows = p.streamed ? osmt_label_band_rows(p.cols) : 1u;
        p.band_words = p.streamed ? osmt_label_band_words(p.cols) : 0u;
        p.stream = reinterpret_cast<const uint32_t*>(g_bits + li->wide_off);
        p.A = g_a + li->plane_off;
        return p;
    };
    /* text item `it` of band `band` (rows rb ..): where its word lands, and which stream bits it is */
    struct TextItem {
        uint32_t wi, sbit, nb, sh;
    };
    auto text_item = [&](const Plan& p, uint32_t rb, uint32_t it, float rcp) -> TextItem {
        uint32_t k;
        const uint32_t r = resolve_divmod(true, it, p.tnw, rcp, k); /* < 64 rows x 21 words */
        const uint32_t w = p.tw0 + k;
        const int32_t lo = max(p.ta, (int32_t)(w << 5)), hi = min(p.tb, (int32_t)(w << 5) + 32);
        TextItem t;
        t.nb = (uint32_t)(hi - lo);
        t.sh = (uint32_t)lo & 31u;
        t.sbit = r * p.cols + (uint32_t)(lo - (p.cx0 + W)); /* first cell of the piece in the band */
        t.wi = (uint32_t)(p.ry0 + (int32_t)(rb + r) + W) * row_words + w;
        return t;
    };
    auto text_mask = [&](const TextItem& t, uint32_t w0, uint32_t w1) -> uint32_t {
        const uint64_t pair = (uint64_t)w0 | ((uint64_t)w1 << 32);
        const uint32_t m = (uint32_t)(pair >> (t.sbit & 31u));
        return (t.nb >= 32u ? m : (m & ((1u << t.nb) - 1u))) << t.sh;
    };
    /* the first 64 text items of a label (band 0): requested one label ahead */
    auto request = [&](const Plan& p, TextItem& t, uint32_t& w0, uint32_t& w1) {
        t = TextItem{0u, 0u, 0u, 0u};
        w0 = w1 = 0u;
        if (!p.streamed) return;
        const uint32_t n_items = min(p.band_rows, p.R) * p.tnw;
        if (lane < n_items) {
            t = text_item(p, 0u, lane, 1.0f / (float)max(p.tnw, 1u));
            w0 = p.stream[t.sbit >> 5];
            w1 = p.stream[(t.sbit >> 5) + 1u];
        }
    };
    uint32_t n_out = 0; /* succeeded labels that reach into the tile itself */
    Plan cur = {};
    TextItem ct = {};
    uint32_t cw0 = 0, cw1 = 0;
    if (l0 < l1) {
        cur = plan_of(l0);
        request(cur, ct, cw0, cw1);
    }
    for (uint32_t l = l0; l < l1; ++l) {
        Plan nxt = {};
        TextItem nt = {};
        uint32_t nw0 = 0, nw1 = 0;
        if (l + 1 < l1) {
            if ((l + 1 - l0) % LR_STAGE == 0u) { /* every staged record has been planned: next 64 */
                wave_lds_order();
                stage(l + 1);
                wave_lds_order();
            }
            nxt = plan_of(l + 1);
            request(nxt, nt, nw0, nw1);
        }
        const Plan& p = cur;
        const uint32_t first_items = p.streamed ? min(p.band_rows, p.R) * p.tnw : 0u;
        const uint32_t first_mask = lane < first_items ? text_mask(ct, cw0, cw1) : 0u;
        bool failed = false;
        for (int pass = 0; pass < 2 && !failed; ++pass) {
            bool hit = false;
            { /* icon rows: runs of ones */
                const uint32_t n_items = p.irows * p.inw;
                const float rcp = 1.0f / (float)max(p.inw, 1u);
                for (uint32_t it = lane; it < n_items; it += 64u) {
                    uint32_t k;
                    const uint32_t r = resolve_divmod(n_items < (1u << 22), it, p.inw, rcp, k);
                    const uint32_t w = p.iw0 + k;
                    const int32_t lo = max(p.ia, (int32_t)(w << 5)), hi = min(p.ib, (int32_t)(w << 5) + 32);
                    const uint32_t nb = (uint32_t)(hi - lo);
                    const uint32_t m = (nb >= 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u)) << ((uint32_t)lo & 31u);
                    word_op(pass, (uint32_t)(p.iya + W + (int32_t)r) * row_words + w, m, hit);
                }
            }
            if (p.streamed) { /* text rows: pieces of the bands' coverage bit streams */
                word_op(pass, ct.wi, first_mask, hit);
                const float rcp = 1.0f / (float)max(p.tnw, 1u);
                uint32_t band = 0;
                for (uint32_t rb = 0; rb < p.R; rb += p.band_rows, ++band) {
                    const uint32_t n_items = min(p.band_rows, p.R - rb) * p.tnw;
                    const uint32_t* __restrict__ sw = p.stream + (size_t)band * p.band_words * 2u;
                    for (uint32_t it = lane + (rb == 0u ? 64u : 0u); it < n_items; it += 64u) {
                        const TextItem t = text_item(p, rb, it, rcp);
                        word_op(pass, t.wi, text_mask(t, sw[t.sbit >> 5], sw[(t.sbit >> 5) + 1u]), hit);
                    }
                }
            } else if (p.has_cells) { /* no bit stream: cell by cell from the totals */
                const uint32_t n_cells = p.R * p.cols;
                const float rcp = 1.0f / (float)p.cols;
                for (uint32_t i = lane; i < n_cells; i += 64u) {
                    if (!(p.A[i] > 0.0)) continue;
                    uint32_t c;
                    const uint32_t r = resolve_divmod(n_cells < (1u << 22), i, p.cols, rcp, c);
                    const int32_t dx = p.cx0 + W + (int32_t)c;
                    if (dx < 0 || dx >= EW) continue; /* rows are clipped already */
                    word_op(pass, (uint32_t)(p.ry0 + (int32_t)r + W) * row_words + ((uint32_t)dx >> 5), 1u << ((uint32_t)dx & 31u), hit);
                }
            }
            if (pass == 0) {
                failed = __ballot(hit) != 0ull;
                if (lane == 0) g_ok[l] = failed ? 0 : 1; /* bump_label_generation(succeeded) */
            }
            if (!LDS_BM) __threadfence();
            wave_lds_order();
        }
        if (!failed && lane == 0) {
            /* what k_raster has to look at: the label's pixels clipped to the tile [0, W)^2 */
            int32_t bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = INT32_MIN, by1 = INT32_MIN;
            if (p.has_cells) {
                bx0 = p.cx0, bx1 = p.cx0 + (int32_t)p.cols - 1, by0 = p.ry0, by1 = p.ry1;
            }
            if (p.iw) {
                bx0 = min(bx0, p.ix0), bx1 = max(bx1, p.ix0 + (int32_t)p.iw - 1);
                by0 = min(by0, p.iy0), by1 = max(by1, p.iy0 + (int32_t)p.ih - 1);
            }
            bx0 = max(bx0, 0), by0 = max(by0, 0), bx1 = min(bx1, W - 1), by1 = min(by1, W - 1);
            if (bx0 <= bx1 && by0 <= by1) {
                osmt_tile_label e;
                e.x0 = (int16_t)bx0, e.y0 = (int16_t)by0, e.x1 = (int16_t)bx1, e.y1 = (int16_t)by1;
                e.label = l;
                e._pad = 0;
                g_tl[l0 + n_out++] = e;
            }
        }
        cur = nxt;
        ct = nt;
        cw0 = nw0;
        cw1 = nw1;
    }
    if (lane == 0) g_tl_cnt[tile] = n_out;
}

hipError_t osmt_launch_labels(const osmt_label_launch& a, hipStream_t st) {
    if (a.n_labels == 0 || a.n_jobs == 0) return hipSuccess;
    const double4* segs = reinterpret_cast<const double4*>(a.segs);
    /* verdicts and the error word start from zero on THIS stream, ordered before the kernels that set them (a clear
     * issued at upload time on another stream could land after them) */
    hipError_t ce = hipMemsetAsync(a.ok, 0, a.n_labels, st);
    if (ce == hipSuccess) ce = hipMemsetAsync(a.err, 0, 4, st);
    if (ce != hipSuccess) return ce;
    if (a.n_bands) hipLaunchKernelGGL(k_label_cover, dim3(a.n_bands), dim3(128), 0, st, a.info, a.bands, a.n_bands, segs, a.plane_a, a.cell_bits, a.err);
    if (a.n_wide)
        hipLaunchKernelGGL(k_label_cover_wide, dim3(a.n_wide), dim3(64), 0, st, a.info, a.wide, a.n_wide, segs, a.plane_a,
                           a.plane_s_wide, a.err);
    const size_t EW = 3u * (size_t)OSMT_TILE_SIZE * a.scale;
    const size_t bm_bytes = ((EW * EW + 31u) / 32u) * 4u;
    if (bm_bytes <= 96u * 1024u) {
        /* per device and cheap: set on every launch rather than caching a process-wide flag */
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_label_resolve<true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (ae != hipSuccess) return ae;
        hipLaunchKernelGGL(k_label_resolve<true>, dim3(a.n_jobs), dim3(64), bm_bytes, st, a.info, a.job_label_off, a.n_jobs,
                           a.scale, a.plane_a, a.cell_bits, a.bitmap, a.ok, a.tile_labels, a.tile_label_cnt);
    } else {
        hipLaunchKernelGGL(k_label_resolve<false>, dim3(a.n_jobs), dim3(64), 0, st, a.info, a.job_label_off, a.n_jobs,
                           a.scale, a.plane_a, a.cell_bits, a.bitmap, a.ok, a.tile_labels, a.tile_label_cnt);
    }
    return hipGetLastError();
}
