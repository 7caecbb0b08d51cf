/*
 * osmt_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the tile hot path.
 *
 *   k_project    Point::from_node                 (src/tile.rs:88-106, src/draw/point.rs:11-19)
 *   k_opinfo     per-op pixel extents, traveled distances and dash tables
 *                (src/draw/line.rs:21-33, src/draw/opacity_calculator.rs:16-30,98-143)
 *   k_raster     fill_contour + draw_lines + set_pixel/blend + to_rgb_triples, fused per
 *                32x16-pixel sub-tile (src/draw/fill.rs, line.rs, opacity_calculator.rs,
 *                tile_pixels.rs, drawer.rs:133-219)
 *   k_composite  blend_pixel over L resident layers + to_rgb_triples (tile_pixels.rs:205-223,164-181)
 *
 * The label pass (k_label_cover / k_label_resolve) lives in osmt_labels.hip — k_raster<LABELS> here blends its
 * survivors — and the PNG encoder (k_png_encode*) in osmt_pngenc.hip.
 *
 * Everything is f64 / integer and compiled with -ffp-contract=off: the reference never
 * forms an FMA and its u8 output is a truncation, so contraction would flip pixels
 * (SURVEY.md §7).  MFMA is unused on purpose: nothing here is a dense contraction.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/osmtile.h"
#include "osmt_geom.h"
#include "osmt_internal.h"

namespace {

constexpr double PI = 3.14159265358979323846264338327950288;

/* Rust `f64 as i32` (saturating, NaN -> 0) */
__device__ __forceinline__ int32_t f64_as_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return (int32_t)v;
}
/* Rust `f64 as u8`: truncation, saturating, NaN -> 0.  fmax(NaN, 0) == 0 (maxNum), so the clamp
 * covers every case in two instructions before the conversion. */
__device__ __forceinline__ uint32_t f64_as_u8(double v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OSMT_V_PLAIN_CVT)
    /* v_cvt_u32_f64 truncates, turns NaN and negative values into 0 and saturates above: one clamp from above is left */
    uint32_t r;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
    return min(r, 255u);
#else
    return (uint32_t)(int32_t)fmin(fmax(v, 0.0), 255.0);
#endif
}

/* Inclusive prefix sum over the 64 lanes of a wave on the DPP network: shifts inside the rows of 16 lanes (a lane
 * whose source falls outside the row adds 0), then the row totals broadcast to the following rows (row_bcast:15 into
 * rows 1 and 3, row_bcast:31 into rows 2 and 3).  Six VALU adds — no LDS permutes to wait for, no lane masks held in
 * scalar registers. */
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false); /* row_shr:1 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false); /* row_shr:2 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false); /* row_shr:4 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false); /* row_shr:8 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); /* row_bcast:15 -> rows 1, 3 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); /* row_bcast:31 -> rows 2, 3 */
    return x;
}

/* ------------------------------------------------------------------------- */
/* tile.rs:88-106 + point.rs:11-19 */
__device__ __forceinline__ void project_point(double lat, double lon, uint32_t zoom, uint32_t tx, uint32_t ty,
                                              double scale, int32_t* ox, int32_t* oy) {
    const double lat_rad = lat * (PI / 180.0);
    const double lon_rad = lon * (PI / 180.0);
    const double x = lon_rad + PI;
    const double y = PI - log(tan((PI / 4.0) + (lat_rad / 2.0)));
    const double dim = (double)(OSMT_TILE_SIZE * (1u << zoom));
    const double px = (x / (2.0 * PI)) * dim;
    const double py = (y / (2.0 * PI)) * dim;
    const double rx = px - (double)(uint32_t)(tx * OSMT_TILE_SIZE);
    const double ry = py - (double)(uint32_t)(ty * OSMT_TILE_SIZE);
    *ox = f64_as_i32(round(rx * scale));
    *oy = f64_as_i32(round(ry * scale));
}

__global__ __launch_bounds__(256) void k_project(const osmt_tile_job* __restrict__ jobs,
                                                 const uint32_t* __restrict__ pt_job,
                                                 const double2* __restrict__ latlon, const uint32_t* __restrict__ refs,
                                                 uint32_t n_pts, double scale, int2* __restrict__ pts, uint32_t* __restrict__ zero,
                                                 uint32_t n_zero) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    /* the pre-pass cursors and list counts of the step that follows (a memset node of its own was 4 us of a 90 us request) */
    if (i < n_zero) zero[i] = 0u;
    if (i >= n_pts) return;
    const uint32_t j = pt_job[i];
    if (j == 0xFFFFFFFFu) {
        pts[i] = make_int2(0, 0);
        return;
    }
    const osmt_tile_job job = jobs[j];
    const double2 ll = latlon[refs ? refs[i] : i]; /* OSMT_COORD_NODE_REF: gather from the shared node table */
    int32_t x, y;
    project_point(ll.x, ll.y, job.zoom, job.x, job.y, scale, &x, &y);
    pts[i] = make_int2(x, y);
}

/* point -> job (big uploads build the table here instead of on the host): one block per job */
__global__ __launch_bounds__(256) void k_ptjob(const osmt_tile_job* __restrict__ jobs, uint32_t* __restrict__ pt_job) {
    const osmt_tile_job job = jobs[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < job.n_pts; i += 256u) pt_job[job.pt_off + i] = blockIdx.x;
}

/* One point, explicit tile (osmt_project). */
__global__ __launch_bounds__(256) void k_project_single(const double2* __restrict__ latlon, uint32_t n, uint32_t zoom,
                                                        uint32_t tx, uint32_t ty, double scale,
                                                        int2* __restrict__ pts) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const double2 ll = latlon[i];
    int32_t x, y;
    project_point(ll.x, ll.y, zoom, tx, ty, scale, &x, &y);
    pts[i] = make_int2(x, y);
}

/* ------------------------------------------------------------------------- */
/* point.rs:21-25 */
__device__ __forceinline__ double point_dist(int32_t ax, int32_t ay, int32_t bx, int32_t by) {
    const double dx = (double)(ax - bx);
    const double dy = (double)(ay - by);
    return sqrt(dx * dx + dy * dy);
}

/* point.rs:27-35 push_away_from */
__device__ __forceinline__ int2 push_away_from(int2 self, int2 other, double by) {
    const double dist = point_dist(self.x, self.y, other.x, other.y);
    const double k = by / dist;
    int2 r;
    r.x = self.x + f64_as_i32(round((double)(self.x - other.x) * k));
    r.y = self.y + f64_as_i32(round((double)(self.y - other.y) * k));
    return r;
}

/* opacity_calculator.rs:98-143 compute_segments, for one calculator: the first `max_out` segments go to `segs`, the number
 * of segments the reference pushes is returned. */
__device__ __forceinline__ int compute_segments(double hlw, const double* __restrict__ dashes, int n_dashes, int cap, osmt_dash_seg* segs,
                                                int max_out, double* total_len) {
    double len_before = 0.0;
    int n = 0;
    for (int it = 0; it <= n_dashes; ++it) {
        const int idx = (it < n_dashes) ? it : 0; /* (0..len).chain(0..1) */
        const double dash = dashes[idx];
        double start = len_before;
        if (idx != 0 || n == 0) len_before += dash;
        if (idx % 2 != 0) continue;
        double end = start + dash;
        osmt_dash_seg s;
        s.orig_a = start;
        s.orig_b = end;
        if (cap == OSMT_CAP_SQUARE || cap == OSMT_CAP_ROUND) {
            start -= hlw;
            end += hlw;
        }
        const double midpoint = (start + end) / 2.0;
        s.start_from = fmin(start - 0.5, midpoint - 1.0);
        s.start_to = fmin(start + 0.5, midpoint);
        s.end_from = fmax(end - 0.5, midpoint);
        s.end_to = fmax(end + 0.5, midpoint + 1.0);
        s.opacity_mul = fmin(end - start, 1.0);
        s.r_start = 1.0 / (s.start_to - s.start_from); /* both ramps are 1 px long up to a rounding: finite */
        s.r_end = 1.0 / (s.end_to - s.end_from);
        if (n < max_out) segs[n] = s;
        ++n;
    }
    *total_len = len_before;
    return n;
}

/* Sub-tiles a virtual segment (an edge or a cap stub) can draw into: every pixel it sets lies within t_extra pixels
 * of the segment's box along its minor axis and within c_extra along its major axis (osmt_reach_of) — the first
 * cull of seg_ranges.  k_opinfo counts these candidates to reserve the op's slots, k_stroke_bin walks exactly the
 * same window with the same inputs (the edge length comes from the same array), so the reservation is exact. */
struct SubWindow {
    int32_t sx0, sx1, sy0, sy1; /* inclusive; empty when sx0 > sx1 or sy0 > sy1 */
};
__device__ __forceinline__ double stroke_ft(double half_width) { return fmax(fabs(half_width) + 0.5, 1.0); }
__device__ __forceinline__ SubWindow vseg_window(int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, double len, double ft,
                                                 int32_t n_sub_x, int32_t n_sub_y) {
    const int32_t dx = abs(p2x - p1x), dy = abs(p2y - p1y);
    const bool swap = dx > dy; /* x is the major axis */
    const osmt_run_reach rr = osmt_reach_of(swap ? dy : dx, swap ? dx : dy, len, ft);
    const int32_t rx = swap ? rr.c_extra : rr.t_extra, ry = swap ? rr.t_extra : rr.c_extra;
    SubWindow w;
    w.sx0 = max((min(p1x, p2x) - rx) >> 5, 0);
    w.sx1 = min((max(p1x, p2x) + rx) >> 5, n_sub_x - 1);
    w.sy0 = max((min(p1y, p2y) - ry) >> OSMT_SUB_H_LOG2, 0);
    w.sy1 = min((max(p1y, p2y) + ry) >> OSMT_SUB_H_LOG2, n_sub_y - 1);
    return w;
}
__device__ __forceinline__ uint32_t window_count(const SubWindow& w) {
    if (w.sx0 > w.sx1 || w.sy0 > w.sy1) return 0u;
    return (uint32_t)(w.sx1 - w.sx0 + 1) * (uint32_t)(w.sy1 - w.sy0 + 1);
}

/* Per-op pre-pass, one thread per op: everything k_raster needs in one 64-byte osmt_opinfo; the traveled distance
 * before every edge of a stroke (line.rs:31: add_traveled_distance, summed in edge order) and every edge's length
 * (= center_dist_denom, line.rs:104) with its reciprocal; the two dash tables and the cap stubs of draw_lines
 * (line.rs:21-22,33-57); 64-edge block boxes of long ops; and the RESERVATIONS in the two arenas: a fill op gets
 * 16 row words per sub-tile of its clipped window, a stroke op one record per (virtual segment, candidate sub-tile).
 * Offsets come from two global cursors (atomicAdd: the order of the ops in the arenas does not matter, k_raster
 * finds everything through osmt_opinfo).  With fmask_cap == srec_cap == 0 the kernel only sizes (upload-time pass). */
#ifndef OSMT_V_OPINFO_THREADS
#define OSMT_V_OPINFO_THREADS 64
#endif
constexpr uint32_t OPINFO_THREADS = OSMT_V_OPINFO_THREADS; /* ops per wave.  Half-empty waves, more of them (measured: config-2 pre-pass 257 us with
                                                           * 64, 274 with 32, 309 with 16) do not help: a wave runs as long as its longest op anyway */
static_assert(OPINFO_THREADS == 64 || OPINFO_THREADS == 32 || OPINFO_THREADS == 16, "the wave scans read their totals from the last active lane");
/* Waves per workgroup.  A wave's two arena reservations are one atomicAdd per cursor — and a same-address atomic costs ~12 ns:
 * 1440 waves (config 2) queue up for 17 us behind each cursor at the end of a 45 us kernel, the 36 000 waves of 256 config-5 tiles
 * for 0.4 ms of its 0.50.  The waves of a workgroup share ONE atomicAdd per cursor (totals and bases handed over through LDS):
 * pre-pass of config 2 0.232 -> 0.209 ms with four waves, 0.205 with eight; 2.80 -> 2.69 ms on 256 config-5 tiles
 * (profiles/r05_o_opinfo_waves_stage_times.txt). */
#ifndef OSMT_V_OPINFO_WAVES
#define OSMT_V_OPINFO_WAVES 8
#endif
constexpr uint32_t OPINFO_WAVES = OSMT_V_OPINFO_WAVES;
static_assert(OPINFO_WAVES == 1 || OPINFO_THREADS == 64, "several waves per workgroup: whole waves");
__global__ __launch_bounds__(OPINFO_THREADS * OPINFO_WAVES) void k_opinfo(const osmt_prepass_args a) {
    const uint32_t op_lane = threadIdx.x % OPINFO_THREADS, op_wave = threadIdx.x / OPINFO_THREADS;
    const uint32_t wblk = blockIdx.x * OPINFO_WAVES + op_wave; /* the wave's 64 ops (waves behind the pool shadow its last op) */
    const uint32_t o_raw = wblk * OPINFO_THREADS + op_lane;
    const bool live = o_raw < a.n_ops;
    const uint32_t o = live ? o_raw : a.n_ops - 1u; /* lanes past the pool shadow the last op and store nothing: the wave
                                                     * stays whole for the reservation at the end */
    osmt_op op = a.ops[o];
    if (!live) op.kind = OSMT_OP_NONE;
    const osmt_ring* __restrict__ rings = a.rings;
    const int2* __restrict__ pts = a.pts;
    const uint32_t sub_rows = a.sub_rows;
    {
        /* the sub-tile words of the wave's 64 ops are one contiguous piece: cleared with whole-line stores (a lane
         * clearing its own op's 16 words touched 64 lines per store) */
        const size_t w_first = min((size_t)a.n_ops, (size_t)wblk * OPINFO_THREADS) * sub_rows;
        const size_t w_end = min((size_t)a.n_ops, (size_t)(wblk + 1u) * OPINFO_THREADS) * sub_rows;
        for (size_t i = w_first + op_lane; i < w_end; i += OPINFO_THREADS) a.submask[i] = 0u;
    }
    osmt_opinfo oi;
    oi.x0 = oi.y0 = INT32_MAX;
    oi.x1 = oi.y1 = INT32_MIN;
    oi.aux = a.op_aux[o];
    oi.n_edges = 0;
    oi.first_pt = op.kind != OSMT_OP_NONE && op.n_rings ? rings[op.ring_off].first_pt : 0u;
    oi.n_rings = op.kind != OSMT_OP_NONE ? op.n_rings : 0u;
    oi.kind = op.kind;
    oi.cap = op.cap;
    oi.color[0] = op.color[0];
    oi.color[1] = op.color[1];
    oi.color[2] = op.color[2];
    oi._pad[0] = oi._pad[1] = oi._pad[2] = 0;
    oi.arena_off = 0;
    oi.rec_cap = 0;
    oi.fill_geom = 0;
    oi.image_id = op.image_id;
    oi.opacity = op.opacity;
    const bool none = op.kind == OSMT_OP_NONE;
    if (none) op.n_rings = 0u;
    const int32_t W = (int32_t)(OSMT_TILE_SIZE * a.scale);
    const int32_t n_sub_y = (int32_t)sub_rows, n_sub_x = W / OSMT_SUB_W;
    const bool is_stroke = op.kind == OSMT_OP_STROKE;
    /* total edges first (rings only): the cap stub of the LAST iterated edge is recognised on the fly */
    uint32_t n_edges = 0;
    for (uint32_t r = 0; r < op.n_rings; ++r) {
        const uint32_t np = rings[op.ring_off + r].n_pts;
        if (np >= 2u) n_edges += np - 1u;
    }
    oi.n_edges = n_edges;
    const double hw = op.width / 2.0;
    const double ft = stroke_ft(hw);
    const bool caps = is_stroke && (op.cap == OSMT_CAP_ROUND || op.cap == OSMT_CAP_SQUARE);
    osmt_cap_seg c0 = {0, 0, 0, 0, 0, 0u, 1.0}, c1 = {0, 0, 0, 0, 0, 0u, 1.0};
    unsigned long long cand = 0ull; /* slots reserved so far: one per (virtual segment, sub-tile of its window) */
    double traveled = 0.0;
    uint32_t seen = 0; /* running edge index + 1 over all rings (point_pairs.rs:36-40) */
    /* bounding boxes of the 64-edge blocks (ops with more than 64 edges only) */
    const uint32_t blk_off = a.op_blk[o];
    const uint32_t vbase = a.op_vseg[o];
    uint32_t cur_blk = 0xFFFFFFFFu;
    osmt_blk_bbox bb = {INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN};
    /* ONE pass over the points.  A thread walks its op alone, so what it waits for is the latency of its own loads:
     * the points are fetched four iterations ahead through a rotating register window. */
    for (uint32_t r = 0; r < op.n_rings; ++r) {
        const osmt_ring ring = rings[op.ring_off + r];
        const int2* __restrict__ rp = pts + ring.first_pt;
        const uint32_t np = ring.n_pts;
        const int2 zero2 = make_int2(0, 0);
        int2 q0 = np > 0u ? rp[0] : zero2, q1 = np > 1u ? rp[1] : zero2, q2 = np > 2u ? rp[2] : zero2, q3 = np > 3u ? rp[3] : zero2;
        int2 prev = zero2;
        for (uint32_t i = 0; i < np; ++i) {
            const int2 p = q0;
            q0 = q1;
            q1 = q2;
            q2 = q3;
            q3 = (i + 4u < np) ? rp[i + 4u] : zero2;
            oi.x0 = min(oi.x0, p.x);
            oi.x1 = max(oi.x1, p.x);
            oi.y0 = min(oi.y0, p.y);
            oi.y1 = max(oi.y1, p.y);
            if (i > 0u) {
                ++seen;
                if (blk_off != 0xFFFFFFFFu) {
                    const uint32_t b_ = (seen - 1u) >> 6; /* running edge index / 64 */
                    if (b_ != cur_blk) {
                        if (cur_blk != 0xFFFFFFFFu) a.blk[blk_off + cur_blk] = bb;
                        cur_blk = b_;
                        bb = {INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN};
                    }
                    bb.x0 = min(bb.x0, min(prev.x, p.x));
                    bb.y0 = min(bb.y0, min(prev.y, p.y));
                    bb.x1 = max(bb.x1, max(prev.x, p.x));
                    bb.y1 = max(bb.y1, max(prev.y, p.y));
                }
                if (is_stroke) {
                    /* |p2 - p1| is both the traveled increment (line.rs:31) and center_dist_denom
                     * (line.rs:104: sqrt(dy*dy + dx*dx) of the absolute deltas — the same f64) */
                    const double len = point_dist(prev.x, prev.y, p.x, p.y);
                    const uint32_t e = vbase + seen - 1u; /* the edge's virtual segment: private to this op even when rings are shared */
                    {
                        osmt_vseg vs;
                        vs.p1x = prev.x; vs.p1y = prev.y; vs.p2x = p.x; vs.p2y = p.y;
                        vs.trav = traveled; /* traveled BEFORE this edge */
                        vs.den = len;
                        vs.rden = 1.0 / len; /* correctly rounded; inf for a degenerate edge (never walked) */
                        vs.cand_off = (uint32_t)min(cand, 0xFFFFFFFFull);
                        vs.vop = o;
                        a.vseg[e] = vs;
                    }
                    traveled += len;
                    if (!(prev.x == p.x && prev.y == p.y)) { /* a degenerate edge draws nothing (line.rs:73-75) */
                        cand += window_count(vseg_window(prev.x, prev.y, p.x, p.y, len, ft, n_sub_x, n_sub_y));
                        /* cap stubs (line.rs:33-57): only for the first / last iterated edge, only if it is not
                         * degenerate (`first` is consumed by a degenerate first edge) */
                        if (caps && seen == 1u) {
                            const int2 ce = push_away_from(prev, p, hw);
                            c0 = {prev.x, prev.y, ce.x, ce.y, 1, 0u, point_dist(ce.x, ce.y, prev.x, prev.y)};
                        }
                        if (caps && seen == n_edges) {
                            const int2 ce = push_away_from(p, prev, hw);
                            c1 = {p.x, p.y, ce.x, ce.y, 1, 0u, point_dist(ce.x, ce.y, p.x, p.y)};
                        }
                    }
                }
            }
            prev = p;
        }
    }
    if (cur_blk != 0xFFFFFFFFu) a.blk[blk_off + cur_blk] = bb;
    /* ---- the two arena reservations of the wave's 64 ops go out as ONE atomicAdd per cursor (wave sums, per-lane
     * offsets from a prefix scan): 2.3 M single-lane atomics on two addresses were what bounded this kernel on the
     * dense config (~1 atomic per clock at the L2) ---- */
    unsigned long long want_f = 0ull, want_s = 0ull; /* fill groups / stroke slots this op reserves */
    uint32_t fill_geom_ok = 0u;
    if (is_stroke) {
        /* a stub that push_away_from rounds back onto its own start draws nothing (line.rs:73-75) */
        if (c0.valid && (c0.p1x != c0.p2x || c0.p1y != c0.p2y)) {
            c0.cand_off = (uint32_t)min(cand, 0xFFFFFFFFull);
            cand += window_count(vseg_window(c0.p1x, c0.p1y, c0.p2x, c0.p2y, c0.denom, ft, n_sub_x, n_sub_y));
        }
        if (c1.valid && (c1.p1x != c1.p2x || c1.p1y != c1.p2y)) {
            c1.cand_off = (uint32_t)min(cand, 0xFFFFFFFFull);
            cand += window_count(vseg_window(c1.p1x, c1.p1y, c1.p2x, c1.p2y, c1.denom, ft, n_sub_x, n_sub_y));
        }
        if (caps) { /* the two stubs are the op's last two virtual segments; an invalid one is stored degenerate (p1 == p2) */
            const osmt_cap_seg* cs[2] = {&c0, &c1};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t e = vbase + n_edges + (uint32_t)i;
                const bool ok = cs[i]->valid != 0;
                osmt_vseg vs;
                vs.p1x = ok ? cs[i]->p1x : 0; vs.p1y = ok ? cs[i]->p1y : 0; vs.p2x = ok ? cs[i]->p2x : 0; vs.p2y = ok ? cs[i]->p2y : 0;
                vs.trav = 0.0;
                vs.den = cs[i]->denom;
                vs.rden = 1.0 / cs[i]->denom;
                vs.cand_off = cs[i]->cand_off;
                vs.vop = o | 0x80000000u;
                a.vseg[e] = vs;
            }
        }
        /* the op's constants: built in registers, stored as three whole lines */
        osmt_stroke_aux sv;
        sv.half_width = hw;
        sv.hlw0 = sqrt(hw * hw - 0.0 * 0.0);
        sv.ff0 = fmax(sv.hlw0 - 0.5, 0.0);
        sv.ft0 = fmax(sv.hlw0 + 0.5, 1.0);
        sv.fd0 = sv.ft0 - sv.ff0;
        sv.rfd0 = 1.0 / sv.fd0;
        sv.mul0 = fmin(2.0 * sv.hlw0, 1.0);
        const int cap_for_dashes = op.use_caps_for_dashes ? op.cap : OSMT_CAP_NONE;
        sv.main_n_segs = 0;
        sv.main_has_orig = 0;
        sv.main_total_len = 0.0;
        sv.main_r_total = 0.0;
        if (op.has_dashes) {
            sv.main_n_segs = compute_segments(hw, a.dashes + op.dashes_off, (int)op.n_dashes, cap_for_dashes,
                                              a.dseg + (size_t)oi.aux * OSMT_MAX_DASH_SEGS, OSMT_MAX_DASH_SEGS, &sv.main_total_len);
            sv.main_has_orig = (cap_for_dashes == OSMT_CAP_ROUND) ? 1 : 0;
            sv.main_r_total = 1.0 / sv.main_total_len; /* used only when total_len > 0 */
        }
        /* the chained (0..1) pass of compute_segments pushes the same segment twice for dashes = [0.0]; max / min over two
         * equal entries equals one entry: one is kept */
        const double zero = 0.0;
        double caps_total;
        (void)compute_segments(hw, &zero, 1, op.cap, &sv.caps_seg, 1, &caps_total);
        sv.caps_has_orig = (op.cap == OSMT_CAP_ROUND) ? 1 : 0;
        sv._pad = 0;
        a.aux[oi.aux] = sv;
        if (cand > 0xFFFFFFFFull) cand = 0xFFFFFFFFull; /* cannot happen below 2^32 records per scene (checked by the host) */
        oi.rec_cap = (uint32_t)cand;
        oi.stroke_ft = ft;
        want_s = cand;
    } else if (!none) {
        /* fills: rows ytop+1 .. ybot carry records (fill.rs:66-72), spans lie inside the points' x range */
        const int32_t ylo = max(oi.y0 + 1, 0), yhi = min(oi.y1, W - 1);
        const int32_t xlo = max(oi.x0, 0), xhi = min(oi.x1, W - 1);
        if (oi.x0 <= oi.x1 && ylo <= yhi && xlo <= xhi) {
            const uint32_t sr0 = (uint32_t)(ylo >> OSMT_SUB_H_LOG2), nsr = (uint32_t)(yhi >> OSMT_SUB_H_LOG2) - sr0 + 1u;
            const uint32_t c0 = (uint32_t)(xlo >> 5), ncols = (uint32_t)(xhi >> 5) - c0 + 1u;
            want_f = (unsigned long long)nsr * ncols;
            fill_geom_ok = sr0 | (c0 << 8) | (ncols << 16) | (nsr << 24);
        }
    }
    {
        /* per-op counts are below 2^32 (clamped) and a fill window below 2^16: 16-bit halves scan without overflow */
        const uint32_t lane = op_lane;
        const uint32_t f = (uint32_t)want_f;
        const uint32_t s_lo = (uint32_t)want_s & 0xFFFFu, s_hi = (uint32_t)(want_s >> 16);
        const uint32_t f_incl = wave_incl_scan(f);
        const unsigned long long s_incl = ((unsigned long long)wave_incl_scan(s_hi) << 16) + wave_incl_scan(s_lo);
        const uint32_t f_tot = (uint32_t)__builtin_amdgcn_readlane((int)f_incl, OPINFO_THREADS - 1);
        const unsigned long long s_tot = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(s_incl >> 32), OPINFO_THREADS - 1) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)s_incl, OPINFO_THREADS - 1);
        unsigned long long f_base = 0ull, s_base = 0ull;
        if (OPINFO_WAVES == 1u) {
            if (lane == 0u) {
                if (f_tot) f_base = atomicAdd(&a.cursors[0], (unsigned long long)f_tot);
                if (s_tot) s_base = atomicAdd(&a.cursors[1], s_tot);
            }
        } else {
            /* the workgroup's waves hand their totals to thread 0, which reserves for all of them and hands the bases back */
            __shared__ unsigned long long s_tot_f[OPINFO_WAVES], s_tot_s[OPINFO_WAVES];
            if (lane == 0u) {
                s_tot_f[op_wave] = f_tot;
                s_tot_s[op_wave] = s_tot;
            }
            __syncthreads();
            if (threadIdx.x == 0u) {
                unsigned long long bf = 0ull, bs = 0ull;
                for (uint32_t w = 0; w < OPINFO_WAVES; ++w) {
                    bf += s_tot_f[w];
                    bs += s_tot_s[w];
                }
                bf = bf ? atomicAdd(&a.cursors[0], bf) : 0ull;
                bs = bs ? atomicAdd(&a.cursors[1], bs) : 0ull;
                for (uint32_t w = 0; w < OPINFO_WAVES; ++w) { /* totals -> first group / record of every wave */
                    const unsigned long long tf = s_tot_f[w], ts = s_tot_s[w];
                    s_tot_f[w] = bf;
                    s_tot_s[w] = bs;
                    bf += tf;
                    bs += ts;
                }
            }
            __syncthreads();
            if (lane == 0u) {
                f_base = s_tot_f[op_wave];
                s_base = s_tot_s[op_wave];
            }
        }
        f_base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(f_base >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)f_base);
        s_base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(s_base >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s_base);
        if (want_s) {
            const unsigned long long off = s_base + (s_incl - want_s);
            oi.arena_off = (uint32_t)off;
            if (a.srec_cap && off + want_s > a.srec_cap) { /* never: the arena was sized by this same code */
                oi.rec_cap = 0;
                if (a.err) *(volatile uint32_t*)a.err = OSMT_PREPASS_ERR_STROKE_ARENA;
            }
        }
        if (want_f) {
            const unsigned long long off = f_base + (f_incl - f);
            oi.arena_off = (uint32_t)off;
            if (!(a.fmask_cap && off + want_f > a.fmask_cap))
                oi.fill_geom = fill_geom_ok;
            else if (a.err)
                *(volatile uint32_t*)a.err = OSMT_PREPASS_ERR_FILL_ARENA;
        }
    }
    if (live) a.info[o] = oi;
}

/* ------------------------------------------------------------------------- */
/* opacity_calculator.rs:171-185 */
__device__ __forceinline__ double opacity_by_center_distance(double cd, double hlw) {
    const double feather_from = fmax(hlw - 0.5, 0.0);
    const double feather_to = fmax(hlw + 0.5, 1.0);
    const double feather_dist = feather_to - feather_from;
    const double opacity_mul = fmin(2.0 * hlw, 1.0);
    double v;
    if (cd < feather_from)
        v = 1.0;
    else if (cd < feather_to)
        v = (feather_to - cd) / feather_dist;
    else
        v = 0.0;
    return opacity_mul * v;
}

/* f64::from(c) / 255.0 for every u8 (tile_pixels.rs:226-228): the compiler folds each entry with a correctly rounded
 * IEEE division, so a lookup returns exactly what the reference computes — without three f64 divisions per op. */
#define OSMT_C4(i) (double)(i) / 255.0, (double)((i) + 1) / 255.0, (double)((i) + 2) / 255.0, (double)((i) + 3) / 255.0
#define OSMT_C16(i) OSMT_C4(i), OSMT_C4((i) + 4), OSMT_C4((i) + 8), OSMT_C4((i) + 12)
#define OSMT_C64(i) OSMT_C16(i), OSMT_C16((i) + 16), OSMT_C16((i) + 32), OSMT_C16((i) + 48)
__constant__ double k_u8_over_255[256] = {OSMT_C64(0), OSMT_C64(64), OSMT_C64(128), OSMT_C64(192)};

/* ---- the fused raster kernel and its two binning kernels ------------------------------------- */
constexpr int SUB = OSMT_SUB_W;    /* sub-tile width in pixels (one 32-bit coverage word per row) */
constexpr int SUBH = OSMT_SUB_H;   /* sub-tile height */
constexpr int NTHREADS = 64;       /* one wave per sub-tile: no cross-wave barrier anywhere */
constexpr int PXT = SUB * SUBH / NTHREADS; /* pixels per thread */
constexpr int ROWSTEP = NTHREADS / SUB;    /* rows between a thread's consecutive pixels */
#ifndef OSMT_V_OPCHUNK
#define OSMT_V_OPCHUNK 16
#endif
/* list entries staged per pass (config 2: ~5 per sub-tile, config 5: ~155).  16 with STAGECAP 16: every fill and every
 * stroke of a chunk has its words / constants staged (with 32 and 8, two in five of config 5's fills fetched theirs with
 * a scalar load of their own in the middle of the per-op loop: 1.38 -> 1.32 ms on 64 config-5 tiles, config 2 unchanged) */
constexpr int OPCHUNK = OSMT_V_OPCHUNK;
static_assert(OPCHUNK == 16 || OPCHUNK == 32 || OPCHUNK == 64, "a chunk is staged by one wave");
#ifndef OSMT_V_SEGCAP
#define OSMT_V_SEGCAP 32
#endif
constexpr int SEGCAP = OSMT_V_SEGCAP; /* slots of one filter pass = stroke records of one group held in LDS (<= 64 lanes) */
static_assert(SEGCAP >= 8 && SEGCAP <= 64, "a filter pass is at most one wave wide");
#ifndef OSMT_V_PLANE_STRIDE
#define OSMT_V_PLANE_STRIDE 32
#endif
constexpr int PLANE_STRIDE = OSMT_V_PLANE_STRIDE;

#if defined(OSMT_ABL) && OSMT_ABL == 5
#define OSMT_DBG(...) __VA_ARGS__
#else
#define OSMT_DBG(...)
#endif
/* One list entry of a chunk as the sequential per-op loop wants it, staged in LDS by the lane that loaded it: the loop
 * reads it with three uniform 16-byte LDS loads and never waits for global (or constant) memory.  The colour terms are
 * the values the reference computes per op — from_color's o * (c / 255) (tile_pixels.rs:12-19) for a fill, c / 255 for
 * a stroke (whose o is per pixel) — looked up and multiplied ONCE by the staging lane, 64 entries side by side,
 * instead of three scalar table loads and their wait in front of every op. */
struct alignas(16) StagedEnt {
    double c0, c1, c2; /* FILL_COLOR: opacity * c/255 of r, g, b; STROKE: c/255 */
    double op;         /* FILL_COLOR: 1.0 - opacity (blend_pixel's factor of the old colour); STROKE: opacity */
    uint32_t arena;    /* as osmt_ent */
    uint32_t kind_stage; /* kind | stage << 8; stage = FILL: index of the staged coverage words, STROKE: of the staged constants; 255: not staged */
    uint32_t aux;
    uint32_t nv;
};
static_assert(sizeof(StagedEnt) == 48, "three 16-byte LDS loads");
/* per-op constants of the un-dashed / cap_dist == 0 across test (osmt_stroke_aux), staged with the entry */
struct StrokeConst {
    double ff0, ft0, fd0, rfd0, mul0;
    uint32_t flags; /* STROKE_* */
    uint32_t _pad;
};
constexpr uint32_t STROKE_PLAIN_MAIN = 1u; /* the main calculator has no dash segments */
constexpr uint32_t STROKE_UNIT_FD = 2u;    /* feather_dist == 1.0 exactly */
constexpr uint32_t STROKE_TINY_MUL = 4u;   /* opacity_mul below 1e-100 (or NaN): no shortcut may assume mul0 * v > 0 */
constexpr uint32_t STROKE_DASH_FAST = 8u;  /* dashed, no original_endpoints, pattern length > 0, mul0 comfortably positive: walk_items_dashed */
__device__ __forceinline__ uint32_t stroke_flags(const osmt_stroke_aux* __restrict__ sa) {
    const bool tiny = !(sa->mul0 >= 1e-100);
    const int n = sa->main_n_segs;
#ifdef OSMT_V_NO_DASH_FAST
    const bool fast = false;
#else
    const bool fast = n > 0 && sa->main_has_orig == 0 && sa->main_total_len > 0.0 && !tiny;
#endif
    return (n == 0 ? STROKE_PLAIN_MAIN : 0u) | (sa->fd0 == 1.0 ? STROKE_UNIT_FD : 0u) | (tiny ? STROKE_TINY_MUL : 0u) | (fast ? STROKE_DASH_FAST : 0u);
}
#ifndef OSMT_V_STAGECAP
#define OSMT_V_STAGECAP 16
#endif
constexpr int STAGECAP = OSMT_V_STAGECAP; /* fills / strokes of one chunk whose data is staged in LDS; the rest reads global memory */

/* What every perpendicular run of a record needs and only the record determines (line.rs:75-104), computed ONCE by the
 * record's lane of the filter pass instead of by every item lane of every walk pass. */
struct alignas(8) SegDer {
    int32_t a, b;     /* mn_delta, mx_delta (0 <= a <= b) */
    float r2b, r2a;   /* v_rcp_f32 of 2b, 2a: quotient estimates of the closed forms (osmt_udiv24r_small) */
    uint32_t w;       /* SEGW_* */
    uint32_t n_main;  /* k_n0 + k_n1 */
    /* center_dist_raw (line.rs:116-117) at the run start (k, c) is k * ru + c * rv, one pixel along the minor axis adds
     * rv, one along the major axis ru (integers below 2^30: exact as f64) */
    double ru, rv;
};
static_assert(sizeof(SegDer) == 40, "five 8-byte LDS words");
constexpr uint32_t SEGW_INCX_NEG = 1u, SEGW_INCY_NEG = 2u, SEGW_SWAP = 4u, SEGW_CAP = 8u, SEGW_SLOW = 16u;

#ifndef OSMT_V_FILTCAP
#define OSMT_V_FILTCAP 256
#endif
constexpr uint32_t FILTCAP = OSMT_V_FILTCAP; /* slots of a group's stroke entries one filter pass looks at (four rounds of 64 lanes) */
#if OSMT_V_OPCHUNK <= OSMT_V_STAGECAP
#define OSMT_STAGE_UNION 1
#else
#define OSMT_STAGE_UNION 0
#endif
constexpr int DTAB_F = 7; /* start_from, start_to, end_from, end_to, opacity_mul, r_start, r_end */
struct RasterShared {
    OSMT_DBG(uint32_t dbg[8];) /* diagnostic build: [0] stroke visits [1] passes [2] items [3] filter passes of groups [4] groups ended by the
                                * 33rd kept record [5] fill visits [6] ops with more than SEGCAP records in the sub-tile [7] ops with more than FILTCAP slots */
    osmt_srec seg[SEGCAP];          /* records of the current group that belong to this sub-tile, compacted */
    SegDer der[SEGCAP];
    uint32_t pre[SEGCAP];           /* inclusive item prefix of the compacted records */
    unsigned long long plane[PLANE_STRIDE * SUBH]; /* generation alpha plane (f64 bit patterns) */
    StagedEnt ent[OPCHUNK];         /* ops of the chunk that draw into this sub-tile, in order */
#if OSMT_STAGE_UNION
    /* an entry is a fill OR a stroke: with a staging slot per ENTRY of the chunk (stage = the entry's place in it) the
     * coverage words and the stroke constants share their 64 bytes — 768 bytes that the dash table and the queue of the
     * dashed walk (below) take instead */
    union {
        uint32_t fmask[SUBH];
        StrokeConst sconst;
    } stg[STAGECAP];
#else
    uint32_t fmask_[STAGECAP][SUBH]; /* coverage words of the first STAGECAP fills of the chunk */
    StrokeConst sconst_[STAGECAP];   /* constants of the first STAGECAP strokes of the chunk */
#endif
    /* walk_items_dashed: the DashSegments of the op being walked (seven of a record's nine doubles: no original_endpoints
     * on that path), staged once per visit, and the queue of the pixels whose dash phase lies within a ramp */
    double dtab[OSMT_MAX_DASH_SEGS][DTAB_F];
    uint32_t queue[64];
#ifdef OSMT_V_LDSPAD
    uint8_t occupancy_experiment_pad[OSMT_V_LDSPAD];
#endif
};
static_assert(sizeof(RasterShared) <= 10240, "16 waves per CU (four per SIMD) share 160 KB of LDS");
#if OSMT_STAGE_UNION
#define SH_FMASK(sh, st) ((sh).stg[st].fmask)
#define SH_SCONST(sh, st) ((sh).stg[st].sconst)
#else
#define SH_FMASK(sh, st) ((sh).fmask_[st])
#define SH_SCONST(sh, st) ((sh).sconst_[st])
#endif
static_assert(FILTCAP <= sizeof(osmt_srec) * SEGCAP && 8u * SEGCAP <= sizeof(SegDer) * SEGCAP && FILTCAP % 256u == 0u,
              "the filter pass borrows seg[] for its entry marks (one 4-byte store per lane) and der[] for the kept slots");

struct SubRect {
    int32_t x0, y0, x1, y1; /* inclusive */
};

__device__ __forceinline__ void blend_px(double* acc, double sr, double sg, double sb, double sa) {
    /* tile_pixels.rs:209-219: new + (1.0 - a) * old, mul then add, no FMA */
    const double k = 1.0 - sa;
    acc[0] = sr + k * acc[0];
    acc[1] = sg + k * acc[1];
    acc[2] = sb + k * acc[2];
    acc[3] = sa + k * acc[3];
}
/* the same for the raster kernel's r,g,b-only accumulators (alpha is the constant 1.0) */
__device__ __forceinline__ void blend_rgb(double* acc, double sr, double sg, double sb, double sa) {
    const double k = 1.0 - sa;
    acc[0] = sr + k * acc[0];
    acc[1] = sg + k * acc[1];
    acc[2] = sb + k * acc[2];
}

/* The lane id as a value the compiler cannot connect to its earlier uses: addresses derived from it (output pixel,
 * staging slots, plane cells) are then computed WHERE they are used instead of once in the prologue and carried — or,
 * at 128 registers, spilled: nine such values cost 300 MB of scratch stores per launch. */
__device__ __forceinline__ uint32_t fresh_lane() {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

/* The kernel's own argument block, re-read from the kernel-argument segment at the point of use: the empty asm makes
 * the pointer opaque, so the compiler can neither hoist the (invariant) loads to the top of the kernel nor keep their
 * results alive across the loops in between. */
/* a (uniform) index the compiler cannot connect to the loads it already made with it */
__device__ __forceinline__ uint32_t late_index(uint32_t i) {
    asm volatile("" : "+s"(i));
    return i;
}

__device__ __forceinline__ const osmt_raster_args* late_args() {
    const osmt_raster_args* p = (const osmt_raster_args*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

/* ---- perpendicular runs (line.rs:108-137) --------------------------------------------------------------------------
 * State of one run, set up from a record and an item index (walk_setup): the pixel in sub-tile coordinates, the error
 * term, and center_dist_raw (line.rs:116-117) as an f64 that is updated by ADDITIONS only.  That is exact: raw is an
 * integer, a run starts within ~2 px of the ideal line (|raw| < 2^33) and continues only while the pixel is in the
 * line (|raw| < feather_to * len < 2^47), so every value the additions produce is an integer below 2^53 — the same
 * number the reference converts from its i64 for every pixel. */
struct RunState {
    int32_t rx, ry;         /* pixel - sub-tile origin */
    int32_t err;            /* perpendicular error term (line.rs:111,128-131) */
    int32_t sx, sy, cx, cy; /* pixel step of every iteration / extra step of a correction */
    int32_t two_a, two_b, b;
    double raw, raw_step, raw_corr;
    double denom, rdenom;
    int32_t d1x, d1y;       /* pixel - p1 (dashed runs: dist(pixel, p1), line.rs:119) */
};

/* The record's share of a run's set-up (filter pass, lane = record). */
__device__ __forceinline__ SegDer seg_derive(const osmt_srec& r, bool is_cap) {
    const int32_t dxs = r.p2x - r.p1x, dys = r.p2y - r.p1y; /* sdx, sdy (line.rs:102-103) */
    const int32_t adx = abs(dxs), ady = abs(dys);
    const bool swap = adx > ady; /* x is the major axis */
    const int32_t incx = r.p1x <= r.p2x ? 1 : -1, incy = r.p1y <= r.p2y ? 1 : -1;
    SegDer d;
    d.a = swap ? ady : adx;
    d.b = swap ? adx : ady;
    d.r2b = osmt_rcp24(2 * d.b);
    d.r2a = osmt_rcp24(2 * d.a); /* a == 0: infinite, and never used (such a segment has no extra perpendiculars) */
    d.w = (incx < 0 ? SEGW_INCX_NEG : 0u) | (incy < 0 ? SEGW_INCY_NEG : 0u) | (swap ? SEGW_SWAP : 0u) | (is_cap ? SEGW_CAP : 0u) |
          (d.b >= OSMT_STEP24_MAX_B ? SEGW_SLOW : 0u);
    d.n_main = (uint32_t)r.k_n0 + (uint32_t)r.k_n1;
    /* raw = sdy * (px - p1x) - sdx * (py - p1y) (line.rs:116-117, the constant cancels at p1); the major axis unit is
     * (incx, 0) when swapped, (0, incy) otherwise, the minor one the other */
    const double fx = (double)(dys * incx), fy = (double)(-dxs * incy); /* |.| <= 2^29: 32-bit products of a delta and +-1 */
    d.ru = swap ? fx : fy;
    d.rv = swap ? fy : fx;
    return d;
}

/* One item of a segment record = one perpendicular run: items [0, k_n0 + k_n1) are the main perpendiculars of steps on
 * side +1 then -1, the rest the extra perpendiculars of line.rs:152-154.  Both kinds go through ONE instruction stream:
 *   main  step k:  c = max(0, ceil((2a k - b) / 2b)),  d = max(0, ceil((2a c - b) / 2b)),  pe = 2a c - 2b d   (osmt_stroke_main)
 *   extra event m: c = floor((2b m - b) / 2a) + 1,      k = floor((2b c - b) / 2a),          pe = 2a c - 2b m   (osmt_extra_event)
 * i.e. twice q = n <= 0 ? 0 : floor((n + add) / Q) with (P, Q) = (2a, 2b) / (2b, 2a) and add = Q - 1 (a ceiling),
 * Q (floor + 1; the numerator of an event is positive) or 0.  Records with b >= 2048 (SEGW_SLOW) take the 64-bit forms.
 * *skip_first: the run is the -1 side of a MAIN step — its first pixel is the Bresenham centre the +1 side of the
 * same step starts on as well (line.rs:139-141 calls both from the same point); see walk_plain. */
__device__ __forceinline__ void walk_setup(RunState& st, const osmt_srec& r, const SegDer& d, uint32_t local, const SubRect& rc, bool* skip_first) {
    const bool is_main = local < d.n_main;
    const uint32_t x = is_main ? local : local - d.n_main;
    const uint32_t n_first = is_main ? (uint32_t)r.k_n0 : (uint32_t)r.n_x0;
    const bool side1 = x >= n_first;
    const int32_t lo = is_main ? (side1 ? r.k_lo1 : r.k_lo0) : (side1 ? r.m_lo1 : r.m_lo0);
    const int32_t idx = lo + (int32_t)(side1 ? x - n_first : x); /* the step k, or the event m */
    *skip_first = is_main && side1;
    const int32_t a = d.a, b = d.b;
    int32_t c, k, pe;
    if (!(d.w & SEGW_SLOW)) {
        const int32_t P = is_main ? 2 * a : 2 * b, Q = is_main ? 2 * b : 2 * a;
        const float rq = is_main ? d.r2b : d.r2a;
        const int32_t n1 = OSMT_MUL24(P, idx) - b;
        c = n1 <= 0 ? 0 : osmt_udiv24r_small(n1 + (is_main ? Q - 1 : Q), Q, rq);
        const int32_t n2 = OSMT_MUL24(P, c) - b;
        const int32_t v2 = n2 <= 0 ? 0 : osmt_udiv24r_small(n2 + (is_main ? Q - 1 : 0), Q, rq);
        k = is_main ? idx : v2;
        pe = OSMT_MUL24(2 * a, c) - OSMT_MUL24(2 * b, is_main ? v2 : idx);
    } else if (is_main) {
        k = idx;
        osmt_stroke_main(a, b, idx, &c, &pe);
    } else {
        osmt_extra_event(a, b, idx, &c, &k, &pe);
    }
    /* start (p_mn, p_mx) = (mx, mn) of the main loop (line.rs:109-110), un-swapped (line.rs:113): k steps along the
     * major axis, c along the minor one */
    const bool swap = (d.w & SEGW_SWAP) != 0u;
    const bool xneg = (d.w & SEGW_INCX_NEG) != 0u, yneg = (d.w & SEGW_INCY_NEG) != 0u;
    const int32_t tx = swap ? k : c, ty = swap ? c : k;
    st.d1x = xneg ? -tx : tx; /* selects and negations, no integer multiplies (quarter rate) */
    st.d1y = yneg ? -ty : ty;
    st.rx = r.p1x + st.d1x - rc.x0;
    st.ry = r.p1y + st.d1y - rc.y0;
    st.err = side1 ? -pe : pe; /* mul * p_error */
    /* every iteration moves the pixel mul along the MINOR axis (p_mx += mul * mn_inc), a correction -mul along the major one */
    const int32_t mx = (side1 != xneg) ? -1 : 1, my = (side1 != yneg) ? -1 : 1; /* mul * incx, mul * incy */
    st.sx = swap ? 0 : mx;
    st.sy = swap ? my : 0;
    st.cx = swap ? -mx : 0;
    st.cy = swap ? 0 : -my;
    st.two_a = 2 * a;
    st.two_b = 2 * b;
    st.b = b;
    if (!(d.w & SEGW_SLOW)) {
        st.raw = fma((double)k, d.ru, (double)c * d.rv); /* integers below 2^42 (k, c <= 2048): exact whatever the rounding */
    } else {
        const int32_t dxs = r.p2x - r.p1x, dys = r.p2y - r.p1y;
        st.raw = (double)((int64_t)dys * (int64_t)st.d1x - (int64_t)dxs * (int64_t)st.d1y);
    }
    const double sg = side1 ? -1.0 : 1.0;
    st.raw_step = sg * d.rv;
    st.raw_corr = -sg * d.ru;
    st.denom = r.denom;
    st.rdenom = r.rdenom;
}

/* update_error (line.rs:91-100) + the pixel move of one iteration (line.rs:128-134) */
__device__ __forceinline__ void walk_advance(RunState& st) {
    const bool corr = st.err + st.two_a > st.b;
    st.err += st.two_a - (corr ? st.two_b : 0);
    st.rx += st.sx + (corr ? st.cx : 0);
    st.ry += st.sy + (corr ? st.cy : 0);
    st.d1x += st.sx + (corr ? st.cx : 0);
    st.d1y += st.sy + (corr ? st.cy : 0);
    st.raw += corr ? st.raw_step + st.raw_corr : st.raw_step; /* integers below 2^53: exact in either order */
}

__device__ __forceinline__ void plane_max(unsigned long long* __restrict__ plane, const RunState& st, double alpha) {
    /* set_pixel inside one generation keeps the larger alpha (tile_pixels.rs:114-118); alpha >= +0, so the u64 order of
     * the bit pattern is the f64 order */
    if ((((uint32_t)st.rx & ~(uint32_t)(SUB - 1)) | ((uint32_t)st.ry & ~(uint32_t)(SUBH - 1))) == 0u)
        atomicMax(&plane[st.ry * PLANE_STRIDE + st.rx], (unsigned long long)__double_as_longlong(alpha));
}

/* A run of a calculator WITHOUT dash segments (get_opacity_by_start_distance returns (1.0, None) without looking at
 * the distance, opacity_calculator.rs:50-55): half_line_width = sqrt(h*h - 0*0) and the feather terms are the per-op
 * constants of osmt_stroke_aux.  With mul0 > 0 (the caller skips ops whose mul0 is not comfortably positive):
 *   is_in_line = cdop > 0  <=>  cd < feather_to   (cd < ft0 makes (ft0 - cd) / fd0 > 0, times mul0 > 0);
 *   opacity = min(1.0, cdop) = cdop               (v <= 1: fl(ft0 - cd) <= fl(ft0 - ff0) = fd0 for cd >= ff0; mul0 <= 1).
 * UNIT_FD: feather_dist == 1.0 exactly (every width whose +-0.5 is exact): x / 1.0 == x.
 * skip_first: the -1 side of a main step starts on the centre pixel the +1 side has just set to the same alpha
 * (max-alpha: a no-op), and that pixel is ALWAYS in the line — it lies within half a pixel of the ideal line along
 * the minor axis, i.e. at distance <= 0.5 < 1.0 <= feather_to — so the run goes on from it without evaluating it.
 * osmt_seg_ranges lists the +1 run of a step for every sub-tile its start can touch, so the centre pixel is never lost. */
template <bool UNIT_FD>
__device__ __forceinline__ void walk_plain(RunState& st, const StrokeConst& kc, double initial_opacity, bool skip_first,
                                           unsigned long long* __restrict__ plane) {
    const double ff0 = kc.ff0, ft0 = kc.ft0, fd0 = kc.fd0, rfd0 = kc.rfd0, mul0 = kc.mul0;
    if (!UNIT_FD && (kc.flags & STROKE_TINY_MUL)) {
        /* opacity_mul so small (|width| < 1e-100) that mul0 * v may underflow to 0: the literal rule, no shortcuts (cold) */
        for (;;) {
            const double cd = osmt_div_exact(fabs(st.raw), st.denom, st.rdenom);
            const double cdop = mul0 * (cd < ff0 ? 1.0 : (cd < ft0 ? (ft0 - cd) / fd0 : 0.0));
            if (!(cdop > 0.0)) break;
            plane_max(plane, st, initial_opacity * fmin(1.0, cdop));
            walk_advance(st);
        }
        return;
    }
    if (skip_first) walk_advance(st);
    for (;;) {
        const double cd = osmt_div_exact(fabs(st.raw), st.denom, st.rdenom); /* == fabs(raw) / denom (line.rs:116-118) */
        if (!(cd < ft0)) break;
        const double num = ft0 - cd;
        const double q = UNIT_FD ? num : osmt_div_exact(num, fd0, rfd0);
        const double cdop = mul0 * (cd < ff0 ? 1.0 : q);
        plane_max(plane, st, initial_opacity * cdop);
        walk_advance(st);
    }
}

/* A run of a calculator WITH dash segments — a dashed edge, or a cap stub (opacity_calculator_for_outer_caps,
 * line.rs:22) — opacity_calculator.rs:32-80 in full.  `t` is wave-uniform (edges and stubs are walked in separate
 * passes), so the table is read with scalar loads. */
__device__ __forceinline__ void walk_dashed(RunState& st, const StrokeConst& kc, const int n, const bool has_orig, const double total,
                                            const double r_total, const osmt_dash_seg* __restrict__ segs, double half_width, double traveled,
                                            double initial_opacity, unsigned long long* __restrict__ plane) {
    const double ff0 = kc.ff0, ft0 = kc.ft0, fd0 = kc.fd0, rfd0 = kc.rfd0, mul0 = kc.mul0;
    for (;;) {
        const double cd = osmt_div_exact(fabs(st.raw), st.denom, st.rdenom);
        /* without original_endpoints cap_dist is 0 for every pixel: the across test is the per-op one, and a pixel
         * beyond feather_to ends the run before any of the along-the-line arithmetic */
        if (!has_orig && !(cd < ft0)) break;
        const double ddx = (double)st.d1x, ddy = (double)st.d1y;
        const double ld = sqrt(ddx * ddx + ddy * ddy);          /* dist(pixel, p1), line.rs:119 */
        const double sd = sqrt(fmax(ld * ld - cd * cd, 0.0));    /* line.rs:120 */
        double dist_rem = traveled + sd;
        if (total > 0.0) { /* dist_rem >= 0: exact `%` (opacity_calculator.rs:57-60), quotient estimated with RN(1 / total) */
            double nq = trunc(dist_rem * r_total);
            double rr = fma(-nq, total, dist_rem);
            if (rr < 0.0) {
                nq -= 1.0;
                rr = fma(-nq, total, dist_rem);
            } else if (rr >= total) {
                nq += 1.0;
                rr = fma(-nq, total, dist_rem);
            }
            dist_rem = rr;
        }
        double sd_op = 0.0, dic = 0.0;
        bool has = false;
        for (int i = 0; i < n; ++i) {
            const osmt_dash_seg* __restrict__ s = &segs[i];
            if (dist_rem < s->start_from || dist_rem > s->end_to) continue; /* :145-157 */
            double base;
            if (dist_rem <= s->start_to) {
                const double d = s->start_to - s->start_from, x = dist_rem - s->start_from;
                base = d == 1.0 ? x : osmt_div_exact(x, d, s->r_start);
            } else if (dist_rem < s->end_from) {
                base = 1.0;
            } else {
                const double d = s->end_to - s->end_from, x = s->end_to - dist_rem;
                base = d == 1.0 ? x : osmt_div_exact(x, d, s->r_end);
            }
            sd_op = fmax(sd_op, s->opacity_mul * base);
            if (has_orig) { /* :159-169 */
                const double dd = dist_rem < s->orig_a ? s->orig_a - dist_rem : (dist_rem <= s->orig_b ? 0.0 : dist_rem - s->orig_b);
                if (!has || dd < dic) {
                    has = true;
                    dic = dd;
                }
            }
        }
        const double cap_dist = has ? dic : 0.0;
        double cdop;
        if (cap_dist == 0.0) {
            const double num = ft0 - cd;
            const double q = fd0 == 1.0 ? num : osmt_div_exact(num, fd0, rfd0);
            const double v = cd < ff0 ? 1.0 : (cd < ft0 ? q : 0.0);
            cdop = mul0 * v;
        } else {
            cdop = opacity_by_center_distance(cd, sqrt(half_width * half_width - cap_dist * cap_dist));
        }
        if (!(cdop > 0.0)) break;
        plane_max(plane, st, initial_opacity * fmin(sd_op, cdop));
        walk_advance(st);
    }
}

/* ---- dashed edges without original_endpoints: the calculator only where the dash phase needs it (round 6) -----------
 * A pixel of a dashed edge costs the full calculator — dist(pixel, p1), two square roots, the exact `%`, the loop over
 * the DashSegments (opacity_calculator.rs:32-80): ~280 instructions against the 57 of an un-dashed one; a quarter of
 * config 2's stroke ops cost as much as the other three quarters (profiles/r06_a_heavy_split.txt).  But WITHOUT
 * original_endpoints (cap_dist == 0 for every pixel: the line cap is not Round, or use_caps_for_dashes is off)
 *   (1) how far a run goes does not depend on the dashes: is_in_line <=> cdop > 0 <=> cd < feather_to, as for an
 *       un-dashed edge (mul0 > 0; STROKE_TINY_MUL ops do not come here);
 *   (2) a pixel whose phase dist_rem lies strictly inside (start_to, end_from) of a segment with opacity_mul == 1.0 has
 *       sd_op == 1.0 (every segment's value is <= 1), so its opacity is min(1.0, cdop) = cdop — the un-dashed pixel;
 *   (3) a pixel whose phase lies outside every [start_from, end_to] has sd_op == 0: opacity 0, set_pixel is a no-op
 *       (max-alpha against a plane of +0.0; initial_opacity >= 0 is validated).
 * Which of the three holds is decided from an APPROXIMATE phase — |dot(pixel - p1, p2 - p1)| / len instead of
 * sqrt(dist(pixel, p1)^2 - cd^2), one multiply on a value that is updated by additions like center_dist_raw — with a
 * margin `mu` that covers the distance between the two.  The reference evaluates, in f64 (S, CD, SD: the exact
 * squared distance to p1, distance to the line, distance along it; S = SD^2 + CD^2),
 *     ld^2 = S (1 + d3), |d3| <= 2^-50;   cd^2 = CD^2 (1 + d5), |d5| <= 2^-49;   diff = SD^2 + e,
 *     |e| <= S 2^-50 + CD^2 2^-49 + 2^-53 |diff| <= SD^2 2^-49 + CD^2 2^-48,
 *     |sqrt(max(diff, 0)) - SD| <= min(|e| / SD, sqrt(2 |e|)) <= SD 2^-49 + CD 2^-23  (CD < feather_to <= 2^15 + 1: <= 2^-7.9),
 * then adds `traveled` (one rounding, <= 2^-53 of the sum) and takes an exact `%`; the approximation has relative error
 * 2^-50.  mu = 2^-7 + 2^-40 (traveled + distance along the line + 64) is above all of it for every input the
 * validation admits (|coordinates| <= 2^28, widths <= 65536).  Pixels within mu of a ramp, of 0 or of the pattern
 * length — a few per cent for real dash patterns — are queued (sub-tile cell + record: four bytes) and evaluated by the
 * EXACT calculator with lanes packed, once the queue holds a wave's worth or the op's runs are walked; everything
 * a queued pixel needs is recomputed from its cell and its record, bit for bit what the run held
 * (center_dist_raw is an exact integer either way).  tests/test_gpu_parity_ops.py::test_dash_phase_*, the fuzzer and the
 * golden crops "dashed" / "subway" hold the two paths against the oracle. */
__device__ __forceinline__ double dashed_exact_opacity(const double (*dtab)[DTAB_F], int n, double total, double r_total, double ddx,
                                                       double ddy, double cd, double traveled) {
    const double ld = sqrt(ddx * ddx + ddy * ddy);       /* dist(pixel, p1), line.rs:119 */
    const double sd = sqrt(fmax(ld * ld - cd * cd, 0.0)); /* line.rs:120 */
    double dist_rem = traveled + sd;
    { /* total > 0, dist_rem >= 0: exact `%` (opacity_calculator.rs:57-60), quotient estimated with RN(1 / total) */
        double nq = trunc(dist_rem * r_total);
        double rr = fma(-nq, total, dist_rem);
        if (rr < 0.0) {
            nq -= 1.0;
            rr = fma(-nq, total, dist_rem);
        } else if (rr >= total) {
            nq += 1.0;
            rr = fma(-nq, total, dist_rem);
        }
        dist_rem = rr;
    }
    double sd_op = 0.0;
    for (int i = 0; i < n; ++i) { /* opacity_calculator.rs:145-157 */
        const double s_from = dtab[i][0], s_to = dtab[i][1], e_from = dtab[i][2], e_to = dtab[i][3];
        if (dist_rem < s_from || dist_rem > e_to) continue;
        double base;
        if (dist_rem <= s_to) {
            const double d = s_to - s_from, x = dist_rem - s_from;
            base = d == 1.0 ? x : osmt_div_exact(x, d, dtab[i][5]);
        } else if (dist_rem < e_from) {
            base = 1.0;
        } else {
            const double d = e_to - e_from, x = e_to - dist_rem;
            base = d == 1.0 ? x : osmt_div_exact(x, d, dtab[i][6]);
        }
        sd_op = fmax(sd_op, dtab[i][4] * base);
    }
    return sd_op;
}

template <class Shared>
__device__ __forceinline__ void dashed_drain(Shared& sh, uint32_t qn, const StrokeConst& kc, int n, double total, double r_total, double initial_opacity,
                                             const SubRect& rc) {
    const uint32_t q = fresh_lane();
    if (q < qn) {
        const uint32_t e = sh.queue[q];
        const uint32_t rx = e & (uint32_t)(SUB - 1), ry = (e >> 5) & (uint32_t)(SUBH - 1);
        const osmt_srec& r = sh.seg[e >> 16];
        const int32_t ddx = rc.x0 + (int32_t)rx - r.p1x, ddy = rc.y0 + (int32_t)ry - r.p1y;
        const int32_t dxs = r.p2x - r.p1x, dys = r.p2y - r.p1y;
        /* center_dist_raw (line.rs:116-117; the constant cancels at p1): the integer the run's additions held */
        const double raw = (double)((int64_t)dys * (int64_t)ddx - (int64_t)dxs * (int64_t)ddy);
        const double cd = osmt_div_exact(fabs(raw), r.denom, r.rdenom);
        const double sd_op = dashed_exact_opacity(sh.dtab, n, total, r_total, (double)ddx, (double)ddy, cd, r.traveled);
        const double num = kc.ft0 - cd; /* cd < feather_to: the run checked it */
        const double qv = kc.fd0 == 1.0 ? num : osmt_div_exact(num, kc.fd0, kc.rfd0);
        const double cdop = kc.mul0 * (cd < kc.ff0 ? 1.0 : qv);
        atomicMax(&sh.plane[ry * PLANE_STRIDE + rx], (unsigned long long)__double_as_longlong(initial_opacity * fmin(sd_op, cdop)));
    }
}

/* the EDGE items among [it_lo, it_hi) of a dashed op without original_endpoints (cap stubs are left to walk_items) */
template <class Shared>
__device__ __forceinline__ void walk_items_dashed(Shared& sh, uint32_t slot0, uint32_t nslot, uint32_t item_base, uint32_t it_lo, uint32_t it_hi,
                                                  const StrokeConst& kc, int n, double total, double r_total, double initial_opacity, const SubRect& rc) {
    const double ff0 = kc.ff0, ft0 = kc.ft0, fd0 = kc.fd0, rfd0 = kc.rfd0, mul0 = kc.mul0;
    uint32_t qn = 0u; /* queued pixels (wave-uniform) */
    for (uint32_t b0 = it_lo; b0 < it_hi; b0 += 64u) {
        const uint32_t it = b0 + fresh_lane();
        bool active = it < it_hi;
        RunState st;
        /* the run's WINDOW: the interval of distances along the polyline (traveled + distance along the edge, un-reduced)
         * around the run's first pixel in which the verdict of that pixel holds — the inside of a dash (2), kind 1, or a gap
         * between dashes (3), kind 2, shrunk by mu on both sides; kind 0: the run starts within a ramp, no window.  The
         * phase changes by less than a pixel per step of a run, so most pixels of a run stay in its window: the per-pixel
         * test is one multiply-add and two compares, the DashSegments are looked at once per RUN. */
        double dotv = 0.0, dot_step = 0.0, dot_corr = 0.0, trav = 0.0, w_lo = 0.0, w_hi = -1.0;
        uint32_t slot_tag = 0u, kind = 0u;
        if (active) {
            uint32_t lo_s = slot0, nn = nslot;
            while (nn > 1u) {
                const uint32_t half = nn >> 1;
                const bool right = sh.pre[lo_s + half - 1u] <= it;
                lo_s = right ? lo_s + half : lo_s;
                nn = right ? nn - half : half;
            }
            const uint32_t base_items = (lo_s == slot0) ? item_base : sh.pre[lo_s - 1u];
            const osmt_srec& r = sh.seg[lo_s];
            const SegDer& d = sh.der[lo_s];
            if (d.w & SEGW_CAP) {
                active = false;
            } else {
                bool skip_first;
                walk_setup(st, r, d, it - base_items, rc, &skip_first);
                const double dxs = (double)(r.p2x - r.p1x), dys = (double)(r.p2y - r.p1y);
                dotv = dxs * (double)st.d1x + dys * (double)st.d1y;
                dot_step = dxs * (double)st.sx + dys * (double)st.sy;
                dot_corr = dxs * (double)st.cx + dys * (double)st.cy;
                trav = r.traveled;
                slot_tag = lo_s << 16;
                const double dt0 = trav + fabs(dotv) * st.rdenom;
                const double mu = 0x1p-7 + 0x1p-40 * (dt0 + 64.0);
                const double base = trunc(dt0 * r_total) * total;
                const double p0 = dt0 - base;
                bool in_any = false, interior = false;
                double g_lo = 0.0, g_hi = total, i_lo = 0.0, i_hi = 0.0;
                for (int i = 0; i < n; ++i) {
                    const double s_from = sh.dtab[i][0], s_to = sh.dtab[i][1], e_from = sh.dtab[i][2], e_to = sh.dtab[i][3];
                    in_any = in_any || (p0 >= s_from && p0 <= e_to);
                    g_lo = e_to < p0 ? fmax(g_lo, e_to) : g_lo;
                    g_hi = s_from > p0 ? fmin(g_hi, s_from) : g_hi;
                    if (sh.dtab[i][4] == 1.0 && p0 > s_to && p0 < e_from) {
                        interior = true;
                        i_lo = fmax(s_to, 0.0);
                        i_hi = fmin(e_from, total);
                    }
                }
                if (interior) {
                    kind = 1u;
                    w_lo = base + i_lo + mu;
                    w_hi = base + i_hi - mu;
                } else if (!in_any) {
                    kind = 2u;
                    w_lo = base + g_lo + mu;
                    w_hi = base + g_hi - mu;
                }
            }
        }
        while (__ballot(active)) {
            bool slow = false;
            if (active) {
                const double cd = osmt_div_exact(fabs(st.raw), st.denom, st.rdenom); /* == fabs(raw) / denom (line.rs:116-118) */
                if (!(cd < ft0)) {
                    active = false; /* (1): the run ends where an un-dashed one would */
                } else if ((((uint32_t)st.rx & ~(uint32_t)(SUB - 1)) | ((uint32_t)st.ry & ~(uint32_t)(SUBH - 1))) == 0u) {
                    const double dt = trav + fabs(dotv) * st.rdenom;
                    if (dt > w_lo && dt < w_hi) {
                        if (kind == 1u) { /* (2) */
                            const double num = ft0 - cd;
                            const double qv = fd0 == 1.0 ? num : osmt_div_exact(num, fd0, rfd0);
                            const double cdop = mul0 * (cd < ff0 ? 1.0 : qv);
                            atomicMax(&sh.plane[st.ry * PLANE_STRIDE + st.rx], (unsigned long long)__double_as_longlong(initial_opacity * cdop));
                        } /* else (3): opacity 0 */
                    } else {
                        slow = true;
                    }
                }
            }
            const unsigned long long sb = __ballot(slow);
            if (sb) {
                const uint32_t ns = (uint32_t)__popcll(sb);
                if (qn + ns > 64u) {
                    dashed_drain(sh, qn, kc, n, total, r_total, initial_opacity, rc);
                    qn = 0u;
                }
                if (slow) sh.queue[qn + (uint32_t)__popcll(sb & ((1ull << fresh_lane()) - 1ull))] = slot_tag | ((uint32_t)st.ry << 5) | (uint32_t)st.rx;
                qn += ns;
            }
            if (active) {
                const bool corr = st.err + st.two_a > st.b;
                dotv += corr ? dot_step + dot_corr : dot_step;
                walk_advance(st);
            }
        }
    }
    if (qn) dashed_drain(sh, qn, kc, n, total, r_total, initial_opacity, rc);
}

/* Items [it_lo, it_hi) of the compacted records [slot0, slot0 + nslot) of ONE op — its edges' records and, behind them,
 * its cap stubs' — lanes packed: the record of item `it` is the first slot whose inclusive item prefix exceeds it
 * (bisection over the LDS prefix).  Edges and stubs share a pass (round 3 walked them in separate ones: a quarter of all
 * passes carried the 2-6 items of two stubs and paid a whole set-up for them); what differs is the calculator, so the
 * lanes of un-dashed edges run the plain loop and then the others run the full one once per table that has lanes —
 * the table of a round is wave-uniform (scalar loads). */
template <class Shared>
__device__ __forceinline__ void walk_items(Shared& sh, uint32_t lane, uint32_t slot0, uint32_t nslot, uint32_t item_base, uint32_t it_lo, uint32_t it_hi,
                                           const StrokeConst& kc, uint32_t sflags, const osmt_stroke_aux* __restrict__ sa, double initial_opacity,
                                           const SubRect& rc) {
    OSMT_DBG(if (lane == 0u) { sh.dbg[1] += (it_hi - it_lo + 63u) / 64u; sh.dbg[2] += it_hi - it_lo; })
    const bool main_plain = (sflags & STROKE_PLAIN_MAIN) != 0u;
    const bool unit_fd = (sflags & (STROKE_UNIT_FD | STROKE_TINY_MUL)) == STROKE_UNIT_FD;
    const bool dash_fast = (sflags & STROKE_DASH_FAST) != 0u;
    if (dash_fast) {
        walk_items_dashed(sh, slot0, nslot, item_base, it_lo, it_hi, kc, sa->main_n_segs, sa->main_total_len, sa->main_r_total, initial_opacity, rc);
        /* the op's cap stubs (another calculator, line.rs:22), if it has any in this sub-tile, go through the loop below */
        const uint32_t l_ = fresh_lane();
        if (!__ballot(l_ < nslot && (sh.der[slot0 + l_].w & SEGW_CAP) != 0u)) return;
    }
    for (uint32_t it = it_lo + lane; it < it_hi; it += 64u) {
        uint32_t lo_s = slot0, n = nslot;
        while (n > 1u) {
            const uint32_t half = n >> 1;
            const bool right = sh.pre[lo_s + half - 1u] <= it;
            lo_s = right ? lo_s + half : lo_s;
            n = right ? n - half : half;
        }
        const uint32_t base_items = (lo_s == slot0) ? item_base : sh.pre[lo_s - 1u];
        const osmt_srec& r = sh.seg[lo_s];
        const SegDer& d = sh.der[lo_s];
        const bool is_cap = (d.w & SEGW_CAP) != 0u;
        RunState st;
        bool skip_first;
        walk_setup(st, r, d, it - base_items, rc, &skip_first);
        if (main_plain && !is_cap) {
            if (unit_fd)
                walk_plain<true>(st, kc, initial_opacity, skip_first, sh.plane);
            else
                walk_plain<false>(st, kc, initial_opacity, skip_first, sh.plane);
        } else {
#ifndef OSMT_V_NODASH /* compile-time probe: the kernel without the full calculator */
#pragma unroll 1
            for (uint32_t t = main_plain ? 1u : 0u; t < 2u; ++t) { /* t = 0: dashed edges (calculator `main`), 1: cap stubs (line.rs:22) */
                if (is_cap != (t == 1u)) continue;
                if (dash_fast && t == 0u) continue; /* walked above */
#ifdef OSMT_V_SKIPHEAVY /* timing probe (wrong pixels): 1 = dashed edges not walked, 2 = cap stubs not walked */
                if (t == (OSMT_V_SKIPHEAVY - 1)) continue;
#endif
                const bool cp = t == 1u; /* wave-uniform: the table of a round is read with scalar loads */
                const osmt_raster_args* la = late_args();
                const osmt_dash_seg* sgs = cp ? &sa->caps_seg : la->dseg + (size_t)(sa - la->aux) * OSMT_MAX_DASH_SEGS;
                walk_dashed(st, kc, cp ? 1 : sa->main_n_segs, (cp ? sa->caps_has_orig : sa->main_has_orig) != 0, cp ? 0.0 : sa->main_total_len,
                            cp ? 0.0 : sa->main_r_total, sgs, sa->half_width, r.traveled, initial_opacity, sh.plane);
            }
#endif
        }
    }
}

/* fill.rs:23-45 for ONE row without storing its records: stream them in (x_min, edge) order by
 * repeated minimum search and OR the paired spans that fall into [x0, x1].  Used only for rows
 * whose crossings overflow the record segment of the fast path (cold; deliberately not inlined). */
__device__ __noinline__ uint32_t fill_row_streaming(const osmt_ring* __restrict__ rings, const int2* __restrict__ pts,
                                                    uint32_t ring_off, uint32_t n_rings, int32_t y, int32_t x0,
                                                    int32_t x1) {
    uint32_t m = 0u;
    int32_t last_x = INT32_MIN;
    int64_t last_e = -1;
    bool have_last = false;
    uint32_t k = 0;
    int32_t from_x = 0;
    for (;;) {
        bool found = false;
        int32_t bx = 0, bxm = 0;
        int64_t be = 0;
        uint32_t eb = 0;
        for (uint32_t r = 0; r < n_rings; ++r) {
            const osmt_ring ring = rings[ring_off + r];
            if (ring.n_pts < 2) continue;
            for (uint32_t e = 0; e + 1 < ring.n_pts; ++e) {
                const int2 p1 = pts[ring.first_pt + e];
                const int2 p2 = pts[ring.first_pt + e + 1];
                int32_t xmn, xmx;
                if (!osmt_fill_row_extent(p1.x, p1.y, p2.x, p2.y, y, &xmn, &xmx)) continue;
                const int64_t ge = (int64_t)eb + e;
                const bool after = !have_last || xmn > last_x || (xmn == last_x && ge > last_e);
                if (!after) continue;
                if (!found || xmn < bx || (xmn == bx && ge < be)) {
                    found = true;
                    bx = xmn;
                    bxm = xmx;
                    be = ge;
                }
            }
            eb += ring.n_pts - 1;
        }
        if (!found) break;
        if ((k & 1u) == 0u) {
            from_x = bx;
        } else {
            const int32_t from = max(from_x, x0);
            const int32_t to = min(bxm, x1);
            if (from <= to) {
                const uint32_t len = (uint32_t)(to - from + 1);
                const uint32_t bits = (len >= 32u) ? 0xFFFFFFFFu : ((1u << len) - 1u);
                m |= bits << (uint32_t)(from - x0);
            }
        }
        have_last = true;
        last_x = bx;
        last_e = be;
        ++k;
    }
    return m;
}

#ifndef OSMT_V_WAVES
#define OSMT_V_WAVES 4
#endif
#if OSMT_V_WAVES > 0
/* waves per SIMD the register allocator must leave room for */
#define OSMT_RASTER_BOUNDS __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(OSMT_V_WAVES, OSMT_V_WAVES)))
#else
#define OSMT_RASTER_BOUNDS __launch_bounds__(NTHREADS)
#endif
#ifdef OSMT_V_NORESTRICT
#define OSMT_R
#else
#define OSMT_R __restrict__
#endif

/* ---- k_fillrows: fill_contour's coverage, once per op and row (fill.rs:16-104) ------------------------------------
 * The unit of work is a CROSSING — an (edge, row) pair on which the edge leaves an un-poisoned record (fill.rs:66-87)
 * — not an (edge, row) combination: only a third of the combinations of a polygon's bounding rows cross, and a
 * polygon a few rows tall would leave most of a wave idle.
 *   pass     64 rows = four slots of 16; a slot is one sub-tile row of one FILL op.  A wave owns FILL_GROUP consecutive
 *            ops and deals their sub-tile rows to the slots in order: four rows of one tall polygon, or the rows of
 *            several small ones.  (FILL_GROUP, measured: pre-pass of config 2 / @2x / config 5 with 1: 263 / 140 / 1177 us,
 *            2: 252 / 145 / 1032, 3: 261 / 161 / 1028, 4: 261 / 169 / 1021, 8: 285 / 204 / 1015 — tall polygons want
 *            more waves, small ones fuller passes.)
 *   step 1   lane = edge (64 per round; ops with more: further rounds, 64-edge blocks whose box misses the rows are
 *            skipped): rows of the pass the edge crosses -> a compact edge list with the running crossing count, and a
 *            difference array over the rows (+1 on the first row, -1 behind the last).
 *   step 1b  lane = row: scans of the difference array give every row's crossing count and its segment of the record
 *            buffer.
 *   step 2   lane = crossing (bisection of the edge list): closed form of the Zingl-Bresenham walk for that row
 *            (osmt_fill_row_extent: Edge{x_min, x_max}), appended to the row's segment.
 *   step 3   lane = row: insertion sort by (x_min, running edge index) = sort_by_key of records inserted in edge order
 *            (fill.rs:24-25), pairs (0,1),(2,3).. OR-ed into one 32-bit word per sub-tile column; the exact "draws
 *            here" bits of the op's sub-tile mask and the list counts of k_sublist fall out of the same words.
 * A window of rows with more crossing edges / crossings than the buffers hold is halved until it fits; a single row
 * beyond them streams its records storage-free (fill_row_streaming, cold). */
#ifndef OSMT_V_FILL_GROUP
#define OSMT_V_FILL_GROUP 2
#endif
constexpr uint32_t FILL_GROUP = OSMT_V_FILL_GROUP;
constexpr uint32_t FILL_EMAX = 128;
#ifndef OSMT_V_FILL_RMAX
#define OSMT_V_FILL_RMAX 384
#endif
/* crossing records of one pass (LDS: 12 bytes each).  384: the workgroup's LDS is 9.5 KB and 16 single-wave workgroups fit a CU
 * (512: 11 KB, 14 — measured in round 5: pre-pass 0.252 -> 0.241 ms on config 2, 3.28 -> 3.15 on 256 config-5 tiles) */
constexpr uint32_t FILL_RMAX = OSMT_V_FILL_RMAX;
struct FillShared {
    int32_t r_xmin[FILL_RMAX];
    int32_t r_xmax[FILL_RMAX];
    uint32_t r_key[FILL_RMAX];
    int2 e_p1[FILL_EMAX], e_p2[FILL_EMAX];
    int32_t e_y0[FILL_EMAX];     /* first crossed row */
    uint32_t e_lane0[FILL_EMAX]; /* its lane in the pass */
    uint32_t e_pre[FILL_EMAX];   /* inclusive crossing count up to this edge */
    uint32_t e_key[FILL_EMAX];   /* running edge index inside the op (fill.rs:19) */
    int32_t diff[65];
    uint32_t rowfill[64];
    uint32_t rowstart[64];
};

__device__ __forceinline__ void fill_rows_body(FillShared& sh, const uint32_t group, const uint32_t lane, const osmt_op* __restrict__ g_ops,
                                               uint32_t n_ops, const osmt_opinfo* __restrict__ g_info, const osmt_ring* __restrict__ g_rings,
                                               const int2* __restrict__ g_pts, const uint32_t* __restrict__ g_op_blk,
                                               const osmt_blk_bbox* __restrict__ g_blk, uint32_t scale, uint32_t sub_rows,
                                               uint32_t* __restrict__ g_submask, uint32_t* __restrict__ g_fmask,
                                               const uint32_t* __restrict__ g_op_job, uint32_t* __restrict__ g_cnt) {
    const uint32_t o_first = group * FILL_GROUP;
    const uint32_t n_sub_x = OSMT_TILE_SIZE * scale / SUB;
    /* the group's ops and their sub-tile rows: slot t of the group = (op k, sub-tile row sr0_k + t - base_k) */
    uint32_t g_geom[FILL_GROUP], g_arena[FILL_GROUP], g_base[FILL_GROUP + 1];
    uint32_t g_ne[FILL_GROUP], g_pt0[FILL_GROUP]; /* SIMPLE ops (one ring, <= 16 edges): edge count and first point; else g_ne = 0xFFFFFFFF */
    g_base[0] = 0u;
#pragma unroll
    for (uint32_t k = 0; k < FILL_GROUP; ++k) {
        uint32_t geom = 0u, arena = 0u, ne = 0u, pt0 = 0u;
        if (o_first + k < n_ops) {
            const osmt_opinfo* __restrict__ oi = &g_info[o_first + k];
            const uint32_t kind = oi->kind;
            if (kind == OSMT_OP_FILL_COLOR || kind == OSMT_OP_FILL_IMAGE) {
                geom = oi->fill_geom; /* nsr == 0: no covered row inside the tile */
                arena = oi->arena_off;
                ne = 0xFFFFFFFFu;
                if (oi->n_rings == 1u && oi->n_edges <= 16u) { /* everything about a small one-ring polygon is in its opinfo */
                    ne = oi->n_edges;
                    pt0 = oi->first_pt;
                }
            }
        }
        g_geom[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)geom);
        g_arena[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)arena);
        g_ne[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ne);
        g_pt0[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)pt0);
        g_base[k + 1] = g_base[k] + (g_geom[k] >> 24);
    }
    const uint32_t n_slots = g_base[FILL_GROUP];
    if (n_slots == 0u) return;
    sh.rowfill[lane] = 0u;
    sh.diff[lane] = 0;
    if (lane == 0u) sh.diff[64] = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n_slots; t0 += 4u) {
        /* ---- this lane's row: slot t0 + lane / 16 ---- */
        const uint32_t t = t0 + (lane >> OSMT_SUB_H_LOG2);
        const bool row_valid = t < n_slots;
        uint32_t my_k = 0u;
#pragma unroll
        for (uint32_t k = 1; k < FILL_GROUP; ++k)
            if (t >= g_base[k] && g_base[k] < n_slots) my_k = k;
        uint32_t geom = g_geom[0], arena = g_arena[0], base = g_base[0];
#pragma unroll
        for (uint32_t k = 1; k < FILL_GROUP; ++k)
            if (my_k == k) {
                geom = g_geom[k];
                arena = g_arena[k];
                base = g_base[k];
            }
        const uint32_t my_o = o_first + my_k;
        const uint32_t sr0 = geom & 255u, c0 = (geom >> 8) & 255u, ncols = row_valid ? (geom >> 16) & 255u : 0u;
        const uint32_t sr = sr0 + (t - base);
        const int32_t y = (int32_t)(sr * SUBH + (lane & (SUBH - 1u)));
        uint32_t row_hit = 0u; /* absolute columns in which this row has coverage */

        /* ---- windows of rows [w0, w0 + ww): halved while the buffers overflow ---- */
        uint32_t w0 = 0u, ww = 64u;
        while (w0 < 64u) {
            const bool in_win = lane >= w0 && lane < w0 + ww;
            /* ---- step 1: crossing edges of the ops that own rows of the window ---- */
            uint32_t n_list = 0u, n_cross = 0u;
            bool overflow = false;
            /* SIMPLE ops only in this window (the common case: a polygon is one ring of a few edges): ONE round for all of
             * them, lane = (op lane / 16, edge lane % 16), instead of a round per op with a quarter of the lanes */
            bool pooled = true;
#pragma unroll
            for (uint32_t k = 0; k < FILL_GROUP; ++k) {
                const uint32_t s_lo = max(g_base[k], t0), s_hi = min(g_base[k + 1], t0 + 4u);
                if (s_lo < s_hi && max((s_lo - t0) * SUBH, w0) < min((s_hi - t0) * SUBH, w0 + ww) && g_ne[k] == 0xFFFFFFFFu) pooled = false;
            }
            if (pooled) {
                /* the (at most four) ops with slots in this pass, in order: lane / 16 picks one of them */
                uint32_t kq[4] = {0xFFu, 0xFFu, 0xFFu, 0xFFu}, nq = 0u;
#pragma unroll
                for (uint32_t q = 0; q < FILL_GROUP; ++q)
                    if (max(g_base[q], t0) < min(g_base[q + 1], t0 + 4u)) {
                        if (nq == 0u) kq[0] = q;
                        if (nq == 1u) kq[1] = q;
                        if (nq == 2u) kq[2] = q;
                        if (nq == 3u) kq[3] = q;
                        ++nq;
                    }
                const uint32_t qi = lane >> 4, e = lane & 15u;
                const uint32_t k = qi == 0u ? kq[0] : (qi == 1u ? kq[1] : (qi == 2u ? kq[2] : kq[3]));
                uint32_t kbase = g_base[0], kend = k == 0u ? g_base[1] : 0u, kgeom = g_geom[0], kne = k == 0u ? g_ne[0] : 0u, kpt0 = g_pt0[0];
#pragma unroll
                for (uint32_t q = 1; q < FILL_GROUP; ++q)
                    if (k == q) {
                        kbase = g_base[q];
                        kend = g_base[q + 1];
                        kgeom = g_geom[q];
                        kne = g_ne[q];
                        kpt0 = g_pt0[q];
                    }
                const uint32_t s_lo = max(kbase, t0), s_hi = min(kend, t0 + 4u);
                uint32_t cnt = 0u, l0 = 0u;
                int32_t first = 0;
                int2 p1 = make_int2(0, 0), p2 = p1;
                if (s_lo < s_hi && e < kne) {
                    const uint32_t la = max((s_lo - t0) * SUBH, w0), lb = min((s_hi - t0) * SUBH, w0 + ww);
                    if (la < lb) {
                        const int32_t ybase = ((int32_t)(kgeom & 255u) + (int32_t)t0 - (int32_t)kbase) * SUBH;
                        const int32_t ya = ybase + (int32_t)la, yb = ybase + (int32_t)lb - 1;
                        p1 = g_pts[kpt0 + e];
                        p2 = g_pts[kpt0 + e + 1u];
                        const int32_t ytop = min(p1.y, p2.y), ybot = max(p1.y, p2.y);
                        first = max(ytop + 1, ya);
                        const int32_t last = min(ybot, yb);
                        cnt = last >= first ? (uint32_t)(last - first + 1) : 0u;
                        l0 = la + (uint32_t)(first - ya);
                    }
                }
                const unsigned long long has = __ballot(cnt != 0u);
                if (has != 0ull) {
                    const uint32_t incl = wave_incl_scan(cnt);
                    const uint32_t pos = (uint32_t)__popcll(has & ((1ull << lane) - 1ull));
                    n_list = (uint32_t)__popcll(has); /* <= 64 <= FILL_EMAX */
                    n_cross = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (n_cross > FILL_RMAX) {
                        overflow = true;
                    } else if (cnt != 0u) {
                        sh.e_p1[pos] = p1;
                        sh.e_p2[pos] = p2;
                        sh.e_y0[pos] = first;
                        sh.e_lane0[pos] = l0;
                        sh.e_pre[pos] = incl;
                        sh.e_key[pos] = e;
                        atomicAdd(&sh.diff[l0], 1);
                        atomicAdd(&sh.diff[l0 + cnt], -1);
                    }
                }
            } else
#pragma unroll 1
            for (uint32_t k = 0; k < FILL_GROUP; ++k) {
                /* slots of op k inside this pass and window -> lanes [la, lb) */
                const uint32_t s_lo = max(g_base[k], t0), s_hi = min(g_base[k + 1], t0 + 4u);
                if (s_lo >= s_hi) continue;
                const uint32_t la = max((s_lo - t0) * SUBH, w0), lb = min((s_hi - t0) * SUBH, w0 + ww);
                if (la >= lb) continue;
                const uint32_t o = o_first + k;
                /* row of lane l of this op: (sr0_k + t0 + l / 16 - base_k) * 16 + l % 16 = ybase + l */
                const int32_t ybase = ((int32_t)(g_geom[k] & 255u) + (int32_t)t0 - (int32_t)g_base[k]) * SUBH;
                const int32_t ya = ybase + (int32_t)la, yb = ybase + (int32_t)lb - 1;
                const osmt_op* __restrict__ op = &g_ops[o];
                const uint32_t ring_off = op->ring_off, n_rings = op->n_rings;
                const uint32_t blk_off = g_op_blk[o];
                uint32_t ring_cur = 0u, e_base = 0u; /* rings before ring_cur lie entirely behind the current round */
                for (uint32_t R0 = 0; ring_cur < n_rings; R0 += 64u) {
                    /* the point of running edge R0 + lane: walk the rings that overlap [R0, R0 + 64) */
                    const uint32_t E = R0 + lane;
                    uint32_t pt = 0xFFFFFFFFu;
                    while (ring_cur < n_rings) {
                        const osmt_ring ring = g_rings[ring_off + ring_cur];
                        const uint32_t ne = ring.n_pts >= 2u ? ring.n_pts - 1u : 0u;
                        if (E >= e_base && E < e_base + ne) pt = ring.first_pt + (E - e_base);
                        if (e_base + ne > R0 + 64u) break; /* the ring goes on in the next round */
                        e_base += ne;
                        ++ring_cur;
                    }
                    if (blk_off != 0xFFFFFFFFu) {
                        const osmt_blk_bbox bb = g_blk[blk_off + (R0 >> 6)];
                        if (bb.y1 < ya || bb.y0 >= yb) continue; /* rows with records of an edge: ytop < y <= ybot */
                    }
                    uint32_t cnt = 0u;
                    int32_t first = 0;
                    int2 p1 = make_int2(0, 0), p2 = p1;
                    if (pt != 0xFFFFFFFFu) {
                        p1 = g_pts[pt];
                        p2 = g_pts[pt + 1u];
                        const int32_t ytop = min(p1.y, p2.y), ybot = max(p1.y, p2.y);
                        first = max(ytop + 1, ya);
                        const int32_t last = min(ybot, yb);
                        cnt = last >= first ? (uint32_t)(last - first + 1) : 0u;
                    }
                    const unsigned long long has = __ballot(cnt != 0u);
                    if (has == 0ull) continue;
                    const uint32_t incl = wave_incl_scan(cnt);
                    const uint32_t pos = n_list + (uint32_t)__popcll(has & ((1ull << lane) - 1ull));
                    n_list += (uint32_t)__popcll(has);
                    const uint32_t n_before = n_cross;
                    n_cross += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (n_list > FILL_EMAX || n_cross > FILL_RMAX) {
                        overflow = true;
                        continue; /* keep walking the rings (cheap); nothing more is stored */
                    }
                    if (cnt != 0u && !overflow) {
                        sh.e_p1[pos] = p1;
                        sh.e_p2[pos] = p2;
                        sh.e_y0[pos] = first;
                        const uint32_t l0 = la + (uint32_t)(first - ya);
                        sh.e_lane0[pos] = l0;
                        sh.e_pre[pos] = n_before + incl;
                        sh.e_key[pos] = E;
                        atomicAdd(&sh.diff[l0], 1);
                        atomicAdd(&sh.diff[l0 + cnt], -1);
                    }
                }
            }
            __syncthreads();
            /* ---- step 1b: crossings per row and the rows' segments of the record buffer ---- */
            const int32_t d = sh.diff[lane];
            sh.diff[lane] = 0;
            if (lane == 0u) sh.diff[64] = 0;
            if (overflow) {
                __syncthreads(); /* the cleared array before the next step 1 */
                if (ww > 1u) {
                    ww >>= 1; /* too many for the buffers: the window's first half, then the rest */
                    continue;
                }
                /* a single row beyond the buffers: storage-free streaming (cold) */
                if (in_win && row_valid) {
                    const osmt_op* __restrict__ op = &g_ops[my_o];
                    for (uint32_t c = 0; c < ncols; ++c) {
                        const int32_t x0 = (int32_t)((c0 + c) * SUB);
                        const uint32_t m = fill_row_streaming(g_rings, g_pts, op->ring_off, op->n_rings, y, x0, x0 + SUB - 1);
                        g_fmask[((size_t)arena + (size_t)(sr - sr0) * ncols + c) * SUBH + (lane & (SUBH - 1u))] = m;
                        if (m) row_hit |= 1u << (c0 + c);
                    }
                }
            }
            if (!overflow) {
                const uint32_t rc_incl = wave_incl_scan((uint32_t)d); /* running sum of the difference array = crossings of the row */
                const uint32_t rowcnt = in_win ? rc_incl : 0u;
                const uint32_t rowstart = wave_incl_scan(rowcnt) - rowcnt;
                sh.rowstart[lane] = rowstart;
                __syncthreads();
                /* ---- step 2: one closed form per crossing ---- */
#if defined(OSMT_ABL) && OSMT_ABL == 11
                if (false)
#endif
                for (uint32_t i = lane; i < n_cross; i += 64u) {
                    uint32_t lo = 0u, n = n_list;
                    while (n > 1u) { /* first edge whose inclusive count exceeds i */
                        const uint32_t half = n >> 1;
                        const bool right = sh.e_pre[lo + half - 1u] <= i;
                        lo = right ? lo + half : lo;
                        n = right ? n - half : half;
                    }
                    const uint32_t kk = i - (lo ? sh.e_pre[lo - 1u] : 0u);
                    const int2 p1 = sh.e_p1[lo], p2 = sh.e_p2[lo];
                    const uint32_t L = sh.e_lane0[lo] + kk;
                    int32_t xmn = 0, xmx = 0;
                    osmt_fill_row_extent(p1.x, p1.y, p2.x, p2.y, sh.e_y0[lo] + (int32_t)kk, &xmn, &xmx); /* crosses by construction */
                    const uint32_t at = sh.rowstart[L] + atomicAdd(&sh.rowfill[L], 1u);
                    sh.r_xmin[at] = xmn;
                    sh.r_xmax[at] = xmx;
                    sh.r_key[at] = sh.e_key[lo];
                }
                __syncthreads();
                sh.rowfill[lane] = 0u;
#if defined(OSMT_ABL) && (OSMT_ABL == 11 || OSMT_ABL == 12)
                if (false) {
#else
                if (in_win && row_valid) {
#endif
                    /* ---- step 3: stable order of the row's records (fill.rs:24-25): by x_min, ties by edge index ---- */
                    for (uint32_t i = 1; i < rowcnt; ++i) {
                        const int32_t kx = sh.r_xmin[rowstart + i], km = sh.r_xmax[rowstart + i];
                        const uint32_t ke = sh.r_key[rowstart + i];
                        int32_t j = (int32_t)i - 1;
                        while (j >= 0) {
                            const int32_t jx = sh.r_xmin[rowstart + (uint32_t)j];
                            const uint32_t je = sh.r_key[rowstart + (uint32_t)j];
                            if (!(jx > kx || (jx == kx && je > ke))) break;
                            sh.r_xmin[rowstart + (uint32_t)j + 1u] = jx;
                            sh.r_xmax[rowstart + (uint32_t)j + 1u] = sh.r_xmax[rowstart + (uint32_t)j];
                            sh.r_key[rowstart + (uint32_t)j + 1u] = je;
                            --j;
                        }
                        sh.r_xmin[rowstart + (uint32_t)(j + 1)] = kx;
                        sh.r_xmax[rowstart + (uint32_t)(j + 1)] = km;
                        sh.r_key[rowstart + (uint32_t)(j + 1)] = ke;
                    }
                    for (uint32_t c = 0; c < ncols; ++c) {
                        const int32_t x0 = (int32_t)((c0 + c) * SUB), x1 = x0 + SUB - 1;
                        uint32_t m = 0u;
                        for (uint32_t k2 = 0; k2 + 1u < rowcnt; k2 += 2u) { /* fill.rs:27-45: pairs; an odd trailing record is ignored */
                            const int32_t from = max(sh.r_xmin[rowstart + k2], x0);
                            const int32_t to = min(sh.r_xmax[rowstart + k2 + 1u], x1);
                            if (from <= to) {
                                const uint32_t len = (uint32_t)(to - from + 1);
                                const uint32_t bits = (len >= 32u) ? 0xFFFFFFFFu : ((1u << len) - 1u);
                                m |= bits << (uint32_t)(from - x0);
                            }
                        }
                        g_fmask[((size_t)arena + (size_t)(sr - sr0) * ncols + c) * SUBH + (lane & (SUBH - 1u))] = m;
                        if (m) row_hit |= 1u << (c0 + c);
                    }
                }
                __syncthreads(); /* the record buffer and the edge list are rewritten by the next window */
            }
            w0 += ww;
            while (ww < 64u && (w0 & (2u * ww - 1u)) == 0u) ww <<= 1; /* both halves done: the wider window again */
        }
        /* ---- the slot's sub-tile bits: OR over its 16 rows (DPP shifts inside the row of 16 lanes) ---- */
        uint32_t hm = row_hit;
        hm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm, 0x111, 0xF, 0xF, false);
        hm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm, 0x112, 0xF, 0xF, false);
        hm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm, 0x114, 0xF, 0xF, false);
        hm |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm, 0x118, 0xF, 0xF, false);
        if (row_valid && (lane & (SUBH - 1u)) == SUBH - 1u) {
            g_submask[(size_t)my_o * sub_rows + sr] = hm;
            /* one more op for the list of every sub-tile hit (k_sublist sizes its lists from these counts) */
            uint32_t* ct = g_cnt + ((size_t)g_op_job[my_o] * sub_rows + sr) * n_sub_x;
            while (hm) {
                atomicAdd(ct + (uint32_t)__builtin_ctz(hm), 1u);
                hm &= hm - 1u;
            }
        }
    }
}

/* ---- k_stroke_bin: which sub-tiles a stroke segment draws into, and with which runs ----------------------------
 * One thread per VIRTUAL SEGMENT of the scene (an edge of a stroke op, or one of its two cap stubs, line.rs:33-57),
 * found by bisection of the host-built prefix table.  For every sub-tile of the segment's window the thread derives
 * the ranges of perpendicular runs that can reach it (seg_ranges) and, when there are any, appends a record to the
 * op's slice of the stroke arena and sets the op's bit for that sub-tile.  k_raster's waves then only FILTER the
 * records of the ops they meet (one coalesced key load per 64 records) instead of each re-deriving the ranges of
 * every segment of every op that comes near: the work is done once per (segment, sub-tile), with full lanes. */
/* what a segment's lane leaves in LDS for the lanes that work on its (segment, sub-tile) pairs */
struct StrokeBinSeg {
    osmt_srec rec;      /* end points, traveled, length, reciprocal (the ranges are filled per pair) */
    double ft;          /* feather_to of the op: max(|half_width| + 0.5, 1.0) */
    int32_t sx0, sy0;   /* first sub-tile of the window */
    uint32_t ncols;     /* window width in sub-tiles */
    uint32_t slot0;     /* absolute arena slot of the window's first sub-tile */
    uint32_t op, job;
    uint32_t is_cap;
    uint32_t _pad;
};
struct StrokeBinShared {
    uint32_t incl[64]; /* inclusive pair count over the block's segments */
    StrokeBinSeg seg[64];
};

__device__ __forceinline__ void stroke_bin_body(StrokeBinShared& sh, const uint32_t blk, const uint32_t lane,
                                                const osmt_opinfo* __restrict__ g_info, uint32_t n_vsegs,
                                                uint32_t scale, uint32_t sub_rows, uint32_t* __restrict__ g_submask,
                                                osmt_srec* __restrict__ g_srec, uint2* __restrict__ g_skey, const uint32_t* __restrict__ g_op_job,
                                                uint32_t* __restrict__ g_cnt, const osmt_vseg* __restrict__ g_vseg) {
    /* ---- step A, lane = virtual segment: its op, end points, sub-tile window — everything k_opinfo left per segment
     * comes in with ONE level of loads, the op's record with a second ---- */
    const uint32_t g = blk * 64u + lane;
    const int32_t W = (int32_t)(OSMT_TILE_SIZE * scale);
    const int32_t n_sub_x = W / SUB, n_sub_y = (int32_t)sub_rows;
    uint32_t n_pairs = 0u;
    if (g < n_vsegs) {
        const osmt_vseg vsr = g_vseg[g];
        const uint32_t vo = vsr.vop;
        const int4 pp = make_int4(vsr.p1x, vsr.p1y, vsr.p2x, vsr.p2y);
        const uint32_t o = vo & 0x7FFFFFFFu;
        StrokeBinSeg sg;
        sg.is_cap = vo >> 31;
        sg.rec.p1x = pp.x; sg.rec.p1y = pp.y; sg.rec.p2x = pp.z; sg.rec.p2y = pp.w;
        sg.rec.traveled = vsr.trav; /* 0 for a cap stub */
        sg.rec.denom = vsr.den;
        sg.rec.rdenom = vsr.rden;
        const uint32_t cand_off = vsr.cand_off;
        if (!(sg.rec.p1x == sg.rec.p2x && sg.rec.p1y == sg.rec.p2y)) { /* line.rs:73-75; also how an invalid stub is stored */
            const osmt_opinfo* __restrict__ oi = &g_info[o];
            sg.ft = oi->stroke_ft;
            const uint32_t rec_cap = oi->rec_cap, arena_off = oi->arena_off;
            const SubWindow w = vseg_window(sg.rec.p1x, sg.rec.p1y, sg.rec.p2x, sg.rec.p2y, sg.rec.denom, sg.ft, n_sub_x, n_sub_y);
            const uint32_t wc = window_count(w);
            if (wc && (unsigned long long)cand_off + wc <= rec_cap) { /* always: k_opinfo reserved this very window */
                sg.sx0 = w.sx0;
                sg.sy0 = w.sy0;
                sg.ncols = (uint32_t)(w.sx1 - w.sx0 + 1);
                sg.slot0 = arena_off + cand_off;
                sg.op = o;
                sg.job = g_op_job[o];
                sg._pad = 0u;
                sg.rec.k_lo0 = sg.rec.k_lo1 = sg.rec.m_lo0 = sg.rec.m_lo1 = 0;
                sg.rec.k_n0 = sg.rec.k_n1 = sg.rec.n_x0 = sg.rec.n_x1 = 0;
                sh.seg[lane] = sg;
                n_pairs = wc;
            }
        }
    }
    const uint32_t incl = wave_incl_scan(n_pairs);
    sh.incl[lane] = incl;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    __syncthreads();
    /* ---- step B, lane = (segment, sub-tile of its window): the ranges of perpendicular runs that can reach the
     * sub-tile (osmt_seg_ranges) -> the slot's record and key, or a hole.  One slot per sub-tile of the window, row-major.
     * Segments have windows of 1 .. 30 sub-tiles: dealt pair by pair the lanes stay full, dealt segment by segment
     * (round 2) the wave ran as long as its largest window with a third of its lanes. ---- */
    for (uint32_t i = lane; i < total; i += 64u) {
        uint32_t v = 0u; /* first segment whose inclusive count exceeds i */
#pragma unroll
        for (uint32_t step = 32u; step; step >>= 1)
            if (sh.incl[v + step - 1u] <= i) v += step;
        const StrokeBinSeg& sg = sh.seg[v];
        const uint32_t j = i - (v ? sh.incl[v - 1u] : 0u);
        const uint32_t ncols = sg.ncols;
        uint32_t qy = (uint32_t)((float)j * __builtin_amdgcn_rcpf((float)ncols)); /* j / ncols for j < 2^11: estimate + fix-up */
        if ((qy + 1u) * ncols <= j) ++qy;
        if (qy * ncols > j) --qy;
        const int32_t sx = sg.sx0 + (int32_t)(j - qy * ncols), sy = sg.sy0 + (int32_t)qy;
        const int32_t x0 = sx * SUB, y0 = sy * SUBH;
        osmt_srec rec = sg.rec;
        osmt_item_ranges ir;
        const uint32_t cnt = osmt_seg_ranges(rec.p1x, rec.p1y, rec.p2x, rec.p2y, rec.denom, sg.ft, x0, y0, x0 + SUB - 1, y0 + SUBH - 1, &ir);
        const size_t slot = (size_t)sg.slot0 + j;
        if (cnt == 0u) {
            g_skey[slot] = make_uint2(0xFFFFFFFFu, 0u);
            continue;
        }
        rec.k_lo0 = ir.k_lo0; rec.k_n0 = (uint16_t)ir.k_n0; rec.k_lo1 = ir.k_lo1; rec.k_n1 = (uint16_t)ir.k_n1;
        rec.m_lo0 = ir.m_lo0; rec.n_x0 = (uint16_t)ir.n_x0; rec.m_lo1 = ir.m_lo1; rec.n_x1 = (uint16_t)ir.n_x1;
        g_srec[slot] = rec;
        g_skey[slot] = make_uint2((uint32_t)(sy * n_sub_x + sx), cnt | (sg.is_cap << 31));
        /* the thread that sets an op's bit first also counts the op into that sub-tile's list (k_sublist) */
        const uint32_t bit = 1u << sx;
        if (!(atomicOr(&g_submask[(size_t)sg.op * sub_rows + (uint32_t)sy], bit) & bit))
            atomicAdd(g_cnt + ((size_t)sg.job * sub_rows + (uint32_t)sy) * (uint32_t)n_sub_x + (uint32_t)sx, 1u);
    }
}

/* Both binning jobs in ONE launch: blocks [0, n_vblk) bin 64 stroke segments each (latency-bound: a bisection, a chain
 * of dependent loads, scattered 72-byte stores), the rest build the coverage rows of FILL_GROUP ops each (issue-bound) —
 * the two kinds overlap on the machine instead of running back to back. */
__global__ __launch_bounds__(64) void k_prebin(const osmt_op* __restrict__ g_ops, uint32_t n_ops, const osmt_opinfo* __restrict__ g_info,
                                               const osmt_ring* __restrict__ g_rings, const int2* __restrict__ g_pts,
                                               const uint32_t* __restrict__ g_op_blk,
                                               const osmt_blk_bbox* __restrict__ g_blk, uint32_t n_vsegs, uint32_t n_vblk,
                                               uint32_t scale, uint32_t sub_rows, uint32_t* __restrict__ g_submask,
                                               uint32_t* __restrict__ g_fmask,
                                               osmt_srec* __restrict__ g_srec, uint2* __restrict__ g_skey, const uint32_t* __restrict__ g_op_job,
                                               uint32_t* __restrict__ g_cnt, const osmt_vseg* __restrict__ g_vseg) {
    __shared__ union {
        FillShared fill;
        StrokeBinShared bin;
    } shu;
    const uint32_t b = blockIdx.x;
#if defined(OSMT_ABL) && OSMT_ABL == 9
    if (b < n_vblk) return; /* ablation: no stroke binning */
#endif
#if defined(OSMT_ABL) && OSMT_ABL == 10
    if (b >= n_vblk) return; /* ablation: no fill rows */
#endif
    if (b < n_vblk)
        stroke_bin_body(shu.bin, b, threadIdx.x, g_info, n_vsegs, scale, sub_rows, g_submask,
                        g_srec, g_skey, g_op_job, g_cnt, g_vseg);
    else
        fill_rows_body(shu.fill, b - n_vblk, threadIdx.x, g_ops, n_ops, g_info, g_rings, g_pts, g_op_blk, g_blk, scale, sub_rows, g_submask, g_fmask,
                       g_op_job, g_cnt);
}

/* ---- k_sublist: the op bits turned round — one ordered list per (tile, sub-tile) ------------------------------------
 * One workgroup per TILE.  The binning kernels left the length of every sub-tile's list (cnt); the workgroup turns
 * the tile's counts into offsets (exclusive scan in LDS) and reserves the tile's total with ONE atomicAdd (131 072
 * waves bumping one cursor cost 1.5 ms; 1024 workgroups do not show).  Then one wave per sub-tile ROW scans the
 * tile's op words of that row 64 ops at a time — one load serves all columns of the row, the op's record is fetched
 * once — and, column by column, the lanes whose op draws there are compacted in order (ballot + popcount: order is
 * semantics, `over` is not commutative) and write the op's entry, arena position resolved for that sub-tile, at
 * their rank. */
#ifndef OSMT_V_SUBLIST_THREADS
#define OSMT_V_SUBLIST_THREADS 1024
#endif
constexpr uint32_t SUBLIST_THREADS = OSMT_V_SUBLIST_THREADS;
#ifndef OSMT_V_SUBLIST_AHEAD
#define OSMT_V_SUBLIST_AHEAD 4
#endif
constexpr uint32_t SUBLIST_AHEAD = OSMT_V_SUBLIST_AHEAD; /* chunks of 64 ops whose records are in flight */
constexpr uint32_t SUBLIST_MAX_SUB = (OSMT_TILE_SIZE * OSMT_MAX_SCALE / OSMT_SUB_W) * (OSMT_TILE_SIZE * OSMT_MAX_SCALE / OSMT_SUB_H);
__global__ __launch_bounds__(SUBLIST_THREADS) void k_sublist(const osmt_tile_job* __restrict__ g_jobs, uint32_t g_scale,
                                                             const osmt_opinfo* __restrict__ g_info, const uint32_t* __restrict__ g_submask,
                                                             uint32_t g_sub_rows, const uint32_t* __restrict__ g_cnt,
                                                             unsigned long long* __restrict__ g_cursor, uint2* __restrict__ g_hdr,
                                                             osmt_ent* __restrict__ g_ent, unsigned long long ent_cap, uint32_t* g_err,
                                                             uint32_t g_fold_max_ops) {
    __shared__ uint32_t s_off[SUBLIST_MAX_SUB]; /* counts, then exclusive offsets inside the tile */
    __shared__ uint32_t s_base[2];              /* first entry of the tile; 1 if the reservation fits */
    __shared__ uint2 s_q[SUBLIST_THREADS / 64u][128]; /* per wave: the ops that draw into its row, sifted out of the tile's ops in order */
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tile = blockIdx.x;
    /* fold mode only (small batches): k_raster's waves build the lists of such a tile themselves.  With g_fold_max_ops == 0 EVERY
     * tile gets its headers written here — an empty tile too (0 <= 0 used to skip it and k_raster read recycled memory) */
    if (g_fold_max_ops != 0u && g_jobs[tile].n_ops <= g_fold_max_ops) return;
    const uint32_t W = OSMT_TILE_SIZE * g_scale;
    const uint32_t nsx = W / SUB;
    const uint32_t nsub = nsx * g_sub_rows;
    const uint32_t* __restrict__ cnt = g_cnt + (size_t)tile * nsub;
    /* The kernel is a chain of dependent round trips — counts -> (scan) -> the tile's reservation -> job -> op bits -> op
     * records -> entries — with one workgroup per tile and nothing else to do meanwhile.  Round 5: what does not depend on the
     * reservation is asked for BEFORE it: the job record with the counts, every wave's first 64 op words while wave 0 scans,
     * their op records while its atomicAdd is on its way. */
    const osmt_tile_job job = g_jobs[tile];
    const uint32_t n_ops = job.n_ops;
    for (uint32_t i = tid; i < nsub; i += SUBLIST_THREADS) s_off[i] = cnt[i];
    auto load_bits = [&](uint32_t sy, uint32_t b0) -> uint32_t {
        const uint32_t i = b0 + lane;
        return (i < n_ops && sy < g_sub_rows) ? g_submask[(size_t)(job.op_off + i) * g_sub_rows + sy] : 0u;
    };
    /* what an entry takes from its op's record, as loaded (bytes 16..19 and 32..63 of the 64-byte osmt_opinfo): the request only
     * names registers, everything that READS them — and so waits for them — happens in `unpack`, one chunk later */
    struct OpRaw {
        uint32_t aux;
        uint4 a, b; /* a: kind | cap | color[0..1], color[2], arena_off, rec_cap;  b: fill_geom, image_id, opacity (two words) */
    };
    static_assert(offsetof(osmt_opinfo, aux) == 16 && offsetof(osmt_opinfo, kind) == 32 && offsetof(osmt_opinfo, color) == 34 &&
                      offsetof(osmt_opinfo, arena_off) == 40 && offsetof(osmt_opinfo, rec_cap) == 44 &&
                      offsetof(osmt_opinfo, fill_geom) == 48 && offsetof(osmt_opinfo, image_id) == 52 && offsetof(osmt_opinfo, opacity) == 56,
                  "k_sublist reads osmt_opinfo as three vector loads");
    struct OpRec {
        osmt_ent e;
        uint32_t geom, arena0;
        bool is_stroke;
    };
    auto fetch = [&](uint32_t w, uint32_t op) -> OpRaw {
        OpRaw r;
        r.aux = 0u;
        r.a = make_uint4(0u, 0u, 0u, 0u);
        r.b = make_uint4(0u, 0u, 0u, 0u);
        if (w != 0u) {
            const char* __restrict__ hi = reinterpret_cast<const char*>(&g_info[job.op_off + op]);
            r.aux = *reinterpret_cast<const uint32_t*>(hi + 16);
            r.a = *reinterpret_cast<const uint4*>(hi + 32);
            r.b = *reinterpret_cast<const uint4*>(hi + 48);
        }
        return r;
    };
    auto unpack = [&](const OpRaw& q) -> OpRec {
        OpRec r;
        r.e = {};
        const uint32_t kind = q.a.x & 255u;
        r.is_stroke = kind == OSMT_OP_STROKE;
        r.arena0 = q.a.z;
        r.geom = q.b.x;
        /* kind | color[0] << 8 | color[1] << 16 | color[2] << 24 (color sits at bytes 2, 3 of the first word and 0 of the second) */
        r.e.kind_color = kind | ((q.a.x >> 16) << 8) | (q.a.y << 24);
        r.e.opacity = __hiloint2double((int)q.b.w, (int)q.b.z);
        r.e.aux = r.is_stroke ? q.aux : q.b.y;
        r.e.nv = r.is_stroke ? q.a.w : 0u;
        return r;
    };
    const uint32_t pre_w = load_bits(wave, 0u); /* the wave's first row is row `wave` */
    __syncthreads();
    if (wave == 0u) {
        /* lane l owns the contiguous piece [l*per, (l+1)*per): serial inside, wave scan across */
        const uint32_t per = (nsub + 63u) / 64u;
        const uint32_t i0 = min(lane * per, nsub), i1 = min(i0 + per, nsub);
        uint32_t sum = 0;
        for (uint32_t i = i0; i < i1; ++i) sum += s_off[i];
        const uint32_t incl = wave_incl_scan(sum);
        uint32_t run = incl - sum;
        for (uint32_t i = i0; i < i1; ++i) {
            const uint32_t c = s_off[i];
            s_off[i] = run;
            run += c;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (lane == 0u) {
            const unsigned long long first = total ? atomicAdd(g_cursor, (unsigned long long)total) : 0ull;
            s_base[0] = (uint32_t)first;
            s_base[1] = (first + total <= ent_cap) ? 1u : 0u; /* always: the arena holds every (op, sub-tile) pair the binning can produce */
            if (!s_base[1] && g_err) *(volatile uint32_t*)g_err = OSMT_PREPASS_ERR_LIST_ARENA; /* the tile would be blank: tell the host */
        }
    }
    __syncthreads();
    const uint32_t base = s_base[0];
    const bool fits = s_base[1] != 0u;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    for (uint32_t sy = wave; sy < g_sub_rows; sy += SUBLIST_THREADS / 64u) {
        /* lane sx keeps the write cursor of column sx */
        uint32_t cur = 0u, row_n = 0u;
        if (lane < nsx) {
            const uint32_t c = cnt[sy * nsx + lane];
            cur = base + s_off[sy * nsx + lane];
            row_n = fits ? c : 0u;
            g_hdr[(size_t)tile * nsub + sy * nsx + lane] = make_uint2(cur, row_n);
        }
        if (__ballot(row_n != 0u) == 0ull) continue; /* nothing draws into this row (or the reservation failed) */
        const bool first_row = sy == wave;
        /* Round 6.  A wave used to take the tile's ops 64 at a time and run the eight-column loop below on every chunk in which
         * ANY op draws into its row: on a config-5 tile 141 chunks of ~240 instructions with six busy lanes each (34 k
         * instructions per wave, 0.32-0.35 ms per 256 tiles, issue-bound at four waves per SIMD).  Now the chunks are only SIFTED
         * — the ops that draw into this row go, in op order, into a ring in LDS (ballot, rank, one store) — and the column
         * loop runs on 64 ops that all do: a tenth of the passes.  The op words are requested SUBLIST_AHEAD chunks before they
         * are sifted; a batch's op records while the next batch is being sifted (the request only names registers, `unpack`
         * is what waits). */
        auto work = [&](const uint32_t w, const OpRaw& raw) {
            const OpRec rec = unpack(raw);
            osmt_ent e = rec.e;
            const uint32_t geom = rec.geom, arena0 = rec.arena0;
            const bool is_stroke = rec.is_stroke;
            const uint32_t sr0 = geom & 255u, c0 = (geom >> 8) & 255u, ncols = (geom >> 16) & 255u;
            for (uint32_t sx = 0; sx < nsx; ++sx) {
                const bool hit = (w >> sx) & 1u;
                const unsigned long long bal = __ballot(hit);
                if (bal == 0ull) continue;
                const uint32_t col_cur = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)sx);
                const uint32_t col_left = (uint32_t)__builtin_amdgcn_readlane((int)row_n, (int)sx);
                const uint32_t pos = (uint32_t)__popcll(bal & lanes_below);
                if (hit && pos < col_left) {
                    /* FILL: word index of this sub-tile's 16 rows; STROKE: the op's first slot */
                    e.arena = is_stroke ? arena0 : (arena0 + (sy - sr0) * ncols + (sx - c0)) * SUBH;
                    g_ent[(size_t)col_cur + pos] = e;
                }
                const uint32_t took = min((uint32_t)__popcll(bal), col_left);
                if (lane == sx) {
                    cur += took;
                    row_n -= took;
                }
            }
        };
        constexpr uint32_t PD = SUBLIST_AHEAD;
        uint2* const q = s_q[wave]; /* (op, its word for this row), 128 places: a sift adds at most 64 to fewer than 64 */
        uint32_t head = 0u, tail = 0u;
        bool have = false;
        uint32_t w_p = 0u;
        OpRaw raw_p = {};
        auto request = [&](uint32_t n) { /* the next n <= 64 sifted ops: their records are asked for */
            __builtin_amdgcn_wave_barrier(); /* one wave's LDS traffic stays in program order; this keeps the compiler from moving it */
            asm volatile("" ::: "memory");
            const uint2 it = lane < n ? q[(head + lane) & 127u] : make_uint2(0u, 0u);
            w_p = it.y;
            raw_p = fetch(it.y, it.x);
            head += n;
            have = true;
        };
        uint32_t wq[PD];
#pragma unroll
        for (uint32_t j = 0; j < PD; ++j) wq[j] = (j == 0u && first_row) ? pre_w : (64u * j < n_ops ? load_bits(sy, 64u * j) : 0u);
        for (uint32_t b0 = 0; b0 < n_ops; b0 += 64u * PD) {
#pragma unroll
            for (uint32_t j = 0; j < PD; ++j) {
                const uint32_t c = b0 + 64u * j; /* this chunk; slot j of the ring holds its words */
                if (c >= n_ops) break;
                const uint32_t w = wq[j];
                wq[j] = c + 64u * PD < n_ops ? load_bits(sy, c + 64u * PD) : 0u;
                const unsigned long long m = __ballot(w != 0u);
                if (m == 0ull) continue;
                if (w != 0u) q[(tail + (uint32_t)__popcll(m & lanes_below)) & 127u] = make_uint2(c + lane, w);
                tail += (uint32_t)__popcll(m);
                if (tail - head >= 64u) {
                    if (have) work(w_p, raw_p);
                    request(64u);
                }
            }
        }
        if (have) work(w_p, raw_p);
        have = false;
        if (tail != head) {
            request(tail - head);
            work(w_p, raw_p);
        }
    }
}

/* Scenes are rendered by waves that each own one 32x16 sub-tile for the whole display list: the premultiplied f64
 * accumulators of their pixels live in registers from reset() to to_rgb_triples().  The pre-pass has already decided
 * what every op does where: fills arrive as 16 coverage words per (op, sub-tile) (k_fill_rows), strokes as records
 * of perpendicular-run ranges per (segment, sub-tile) (k_stroke_bin).  What is left here is the part that needs the
 * pixels: walking the runs of a generation into the LDS alpha plane (set_pixel keeps the larger alpha,
 * tile_pixels.rs:114-118) and blending generation after generation in order (tile_pixels.rs:205-223). */
/* blend_pixel (tile_pixels.rs:209-219) of one wave-uniform source colour into the pixels whose bit is set in `m`
 * (bit = lane): new = s + k * old, k = 1 - alpha, mul then add (no FMA).  The coverage of a fill arrives as whole
 * words, so the set of lanes IS the execution mask: six instructions for the covered lanes instead of six plus six
 * selects (and the bit extraction) for all of them. */
__device__ __forceinline__ void blend_masked(double& r, double& g, double& b, double sr, double sg, double sb, double k, unsigned long long m) {
    unsigned long long save;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_and_b64 exec, %[sv], %[m]\n\t"
        "v_mul_f64 %[r], %[k], %[r]\n\t"
        "v_mul_f64 %[g], %[k], %[g]\n\t"
        "v_mul_f64 %[b], %[k], %[b]\n\t"
        "v_add_f64 %[r], %[sr], %[r]\n\t"
        "v_add_f64 %[g], %[sg], %[g]\n\t"
        "v_add_f64 %[b], %[sb], %[b]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [r] "+v"(r), [g] "+v"(g), [b] "+v"(b), [sv] "=&s"(save)
        : [k] "v"(k), [sr] "v"(sr), [sg] "v"(sg), [sb] "v"(sb), [m] "s"(m)
        : "scc");
}

/* blend_pixel for the pending pixels of ONE stroke generation (tile_pixels.rs:205-223), one of a lane's eight pixels at a time:
 * source colour from_color(c, al) = (al * c/255, al) (tile_pixels.rs:12-19), dst = src + (1 - al) * dst, mul then add.  Only the
 * lanes whose cell was drawn into (al > 0: an untouched cell holds +0.0, for which the blend is the identity, bit for bit) — and
 * when NO lane of the wave has one, i.e. the stroke did not touch these two rows of the sub-tile, the ten instructions are branched
 * over: a stroke that crosses a sub-tile touches three or four of its eight row pairs.  (Written as compiler-visible control flow
 * the same skip cost the registers it saved — round 4; inside one asm statement the allocator sees a straight line.) */
__device__ __forceinline__ void blend_stroke_masked(double& r, double& g, double& b, double al, double c0, double c1, double c2) {
#ifdef OSMT_V_PLAIN_STROKE_BLEND
    const double k = 1.0 - al;
    r = al * c0 + k * r;
    g = al * c1 + k * g;
    b = al * c2 + k * b;
#else
    unsigned long long save;
    double t0, t1, t2, k;
    asm volatile(
        "v_cmp_lt_f64 vcc, 0, %[al]\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "s_cbranch_execz 1f\n\t"
        "v_mul_f64 %[t0], %[al], %[c0]\n\t"
        "v_mul_f64 %[t1], %[al], %[c1]\n\t"
        "v_mul_f64 %[t2], %[al], %[c2]\n\t"
        "v_add_f64 %[k], 1.0, -%[al]\n\t"
        "v_mul_f64 %[r], %[k], %[r]\n\t"
        "v_mul_f64 %[g], %[k], %[g]\n\t"
        "v_mul_f64 %[b], %[k], %[b]\n\t"
        "v_add_f64 %[r], %[t0], %[r]\n\t"
        "v_add_f64 %[g], %[t1], %[g]\n\t"
        "v_add_f64 %[b], %[t2], %[b]\n"
        "1:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [r] "+v"(r), [g] "+v"(g), [b] "+v"(b), [sv] "=&s"(save), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [k] "=&v"(k)
        : [al] "v"(al), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2)
        : "vcc", "scc");
#endif
}

template <bool OUT_F64, bool LABELS, bool FOLD>
__global__ OSMT_RASTER_BOUNDS void k_raster(
    /* ONE by-value argument block.  The tables of the hot loops (lists, coverage words, stroke records, calculator
     * constants) are taken from it once and live in SGPRs; everything that is needed only at one point — the job record
     * and list header at the start, the icon pool of a rare image fill, the label tables and the output pointer of the
     * epilogue — is re-read from the kernel-argument segment WHERE it is used (late()): 20 pointers held across the
     * whole kernel cost 61-87 SGPR spills (v_writelane / v_readlane traffic in every loop level). */
    const osmt_raster_args a) {
    const osmt_ent* OSMT_R g_ent = a.ent;
    const osmt_stroke_aux* OSMT_R g_aux = a.aux;
    const uint32_t* OSMT_R g_fmask = a.fmask;
    const osmt_srec* OSMT_R g_srec = a.srec;
    const uint2* OSMT_R g_skey = a.skey;
    const uint32_t g_scale = a.scale;
    __shared__ RasterShared sh;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid;
    const uint32_t W = OSMT_TILE_SIZE * g_scale;
    const uint32_t subs_per_row = W / SUB;
    const uint32_t nsub = subs_per_row * (W / SUBH);

    /* XCD-aware block -> (tile, sub-tile): blocks b, b+8, b+16.. land on one XCD, so give
     * them the sub-tiles of the same tiles (they share that tile's tables in L2). */
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u;
    const uint32_t rest = b >> 3;
    const uint32_t tile = (rest / nsub) * 8u + xcd;
    const uint32_t sub = rest % nsub;
    if (tile >= a.n_jobs) return;

    const osmt_tile_job job = a.jobs[tile];
    SubRect rc;
    const uint32_t sub_x = sub % subs_per_row, sub_y = sub / subs_per_row;
    rc.x0 = (int32_t)(sub_x * SUB);
    rc.y0 = (int32_t)(sub_y * SUBH);
    rc.x1 = rc.x0 + SUB - 1;
    rc.y1 = rc.y0 + SUBH - 1;

    /* thread -> pixels: column tid % 32, rows tid / 32 + 2*j; a wave covers two full 128-byte rows.  (Both are
     * re-derived from a fresh lane id where they are used: carried from here they occupy registers across every loop.) */

    /* tile_pixels.rs:89-93 reset.  Only r,g,b are carried: the canvas alpha starts at 1.0 and
     * blend_pixel keeps it at exactly 1.0 — fl(a + fl(1-a)*1.0) == 1.0 for every alpha in
     * [0, 2^52] (1-a is exact for a >= 0.5; below, the rounding error of 1-a is <= 2^-54 and
     * 1 + e rounds to 1.0) — so it is a constant, checked bit-for-bit by the f64 parity tests. */
#ifdef OSMT_V_ACCPROBE /* occupancy probe (wrong pixels): pixels share accumulators, the instruction stream is the same */
#define AJ(j) ((j) % OSMT_V_ACCPROBE)
#else
#define AJ(j) (j)
#endif
    double acc[PXT][3];
    {
        double r = 0.0, g = 0.0, bl = 0.0;
        if (job.has_canvas) {
            r = 1.0 * k_u8_over_255[job.canvas_rgb[0]];
            g = 1.0 * k_u8_over_255[job.canvas_rgb[1]];
            bl = 1.0 * k_u8_over_255[job.canvas_rgb[2]];
        }
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            acc[AJ(j)][0] = r;
            acc[AJ(j)][1] = g;
            acc[AJ(j)][2] = bl;
        }
    }
    bool plane_clean = false; /* the alpha plane is cleared when the first stroke op shows up */
    OSMT_DBG(if (lane < 8) sh.dbg[lane] = 0u; __syncthreads();)

    /* this sub-tile's own list (k_sublist): the ops that draw here, in order, OPCHUNK at a time.  (Fetching the next
     * chunk's entries a chunk ahead cost eight registers for the whole chunk: 21 more spilled registers at 128, whose
     * scratch stores more than doubled the kernel's HBM writes.) */
    /* Small batches: a tile of at most fold_max_ops (<= 128) ops has no lists (k_sublist skips it): lane l looks at the bits of ops l and
     * l + 64 for this sub-tile's row, two ballots give the ops that draw here IN OP ORDER, and a chunk's entries are put
     * together from the op records by the lanes of their ops — the same two dependent loads as header -> entries, one
     * kernel (32 us of a config-2 step, 8 us of a one-tile request) fewer. */
    /* FOLD: the instantiation small batches are rendered with (osmt_launch_raster); the other one is the kernel big batches
     * have always had — not an instruction of it differs */
    const bool fold = FOLD && job.n_ops <= OSMT_FOLD_MAX_OPS;
    /* the op bits of this sub-tile, asked for again by every chunk (one chunk per sub-tile on config 2): a word that stayed
     * in a register across the chunk loop is a register the kernel does not have */
    auto op_bits = [&](unsigned long long* b0, unsigned long long* b1) {
        const osmt_raster_args* la = late_args();
        const osmt_tile_job* OSMT_R jb = &la->jobs[tile];
        const uint32_t n_ops = jb->n_ops, sub_rows = W / SUBH;
        const uint32_t* OSMT_R sm = la->submask + (size_t)jb->op_off * sub_rows + sub / subs_per_row;
        const uint32_t t_ = fresh_lane(), sx_ = sub % subs_per_row;
        const uint32_t w0 = t_ < n_ops ? sm[(size_t)t_ * sub_rows] : 0u;
        const uint32_t w1 = t_ + 64u < n_ops ? sm[(size_t)(t_ + 64u) * sub_rows] : 0u;
        *b0 = __ballot((w0 >> sx_) & 1u);
        *b1 = __ballot((w1 >> sx_) & 1u);
    };
    uint2 hdr = make_uint2(0u, 0u);
    if (fold) {
        unsigned long long b0, b1;
        op_bits(&b0, &b1);
        hdr.y = (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
    } else {
        /* (requested beside the job record, not behind it) a tile without ops is plain canvas (drawer.rs:60-131 with no
         * areas): what its headers hold is never looked at — a scene without any op has no list kernel launch at all */
        hdr = a.hdr[(size_t)tile * nsub + sub];
        if (job.n_ops == 0u) hdr = make_uint2(0u, 0u);
    }
#if defined(OSMT_ABL) && OSMT_ABL == 8
    const uint32_t n_ent = hdr.y > 0xFFFFFFF0u ? 1u : 0u; /* ablation: the list is not even staged */
#else
    const uint32_t n_ent = hdr.y;
#endif
    const osmt_ent* OSMT_R my_ent = g_ent + hdr.x;

    for (uint32_t base = 0; base < n_ent; base += OPCHUNK) {
        const uint32_t total = min((uint32_t)OPCHUNK, n_ent - base);
        const bool hit = lane < total;
        osmt_ent e = {};
        if (FOLD && late_args()->jobs[tile].n_ops <= OSMT_FOLD_MAX_OPS) { /* (asked again: a flag kept across the loop is a register) */
            osmt_ent* const tmp = reinterpret_cast<osmt_ent*>(sh.seg); /* the previous chunk's records are consumed */
            static_assert(sizeof(osmt_ent) * OPCHUNK <= sizeof(sh.seg), "a chunk's entries fit the record array");
            const osmt_opinfo* OSMT_R g_info = late_args()->info + late_args()->jobs[tile].op_off;
            unsigned long long bb[2];
            op_bits(&bb[0], &bb[1]);
            const uint32_t n0 = (uint32_t)__popcll(bb[0]);
#pragma unroll
            for (uint32_t half = 0; half < 2u; ++half) {
                const uint32_t t_ = fresh_lane();
                const uint32_t r = (half ? n0 : 0u) + (uint32_t)__popcll(bb[half] & ((1ull << t_) - 1ull));
                if (((bb[half] >> t_) & 1ull) && r >= base && r < base + (uint32_t)OPCHUNK) {
                    const osmt_opinfo* OSMT_R hi = &g_info[t_ + 64u * half];
                    const uint32_t kind = hi->kind;
                    const bool strk = kind == OSMT_OP_STROKE;
                    const uint32_t geom = hi->fill_geom, arena0 = hi->arena_off;
                    osmt_ent t;
                    t.kind_color = kind | ((uint32_t)hi->color[0] << 8) | ((uint32_t)hi->color[1] << 16) | ((uint32_t)hi->color[2] << 24);
                    t.opacity = hi->opacity;
                    t.aux = strk ? hi->aux : hi->image_id;
                    t.nv = strk ? hi->rec_cap : 0u;
                    /* FILL: word index of this sub-tile's 16 rows (sr0 | c0 << 8 | ncols << 16); STROKE: the op's first slot */
                    t.arena = strk ? arena0
                                   : (arena0 + (sub / subs_per_row - (geom & 255u)) * ((geom >> 16) & 255u) + (sub % subs_per_row - ((geom >> 8) & 255u))) * SUBH;
                    t.stage = 0u;
                    t._pad = 0u;
                    tmp[r - base] = t;
                }
            }
            __syncthreads();
            if (hit) e = tmp[fresh_lane()];
        } else if (hit) {
            e = my_ent[base + lane];
        }
        const uint32_t e_kind = e.kind_color & 255u;
        const bool is_stroke = hit && e_kind == OSMT_OP_STROKE;
        const unsigned long long sbal = __ballot(is_stroke);
#if OSMT_STAGE_UNION
        const uint32_t my_stage = fresh_lane(); /* one staging slot per entry of the chunk */
#else
        const unsigned long long fbal = __ballot(hit && !is_stroke);
        const uint32_t my_stage = (uint32_t)__popcll((is_stroke ? sbal : fbal) & ((1ull << fresh_lane()) - 1ull));
#endif
        /* slots of the stroke entries, prefix-summed over the chunk's lanes: the groups of the filter passes are cut out
         * of this scan with a ballot instead of a scalar loop over the entries (clamped: only "more than a pass" matters) */
        const uint32_t nv_incl = wave_incl_scan(is_stroke ? min(e.nv, 1u << 20) : 0u);
        __syncthreads(); /* the previous chunk's list is consumed */
        if (hit) {
            StagedEnt se;
            const double cr = k_u8_over_255[(e.kind_color >> 8) & 255u], cg = k_u8_over_255[(e.kind_color >> 16) & 255u],
                         cb = k_u8_over_255[e.kind_color >> 24];
            const bool fc = e_kind == OSMT_OP_FILL_COLOR;
            se.c0 = fc ? e.opacity * cr : cr; /* from_color: o * (c / 255) */
            se.c1 = fc ? e.opacity * cg : cg;
            se.c2 = fc ? e.opacity * cb : cb;
            se.op = fc ? 1.0 - e.opacity : e.opacity;
            se.arena = e.arena;
            se.kind_stage = e_kind | ((my_stage < (uint32_t)STAGECAP ? my_stage : 255u) << 8);
            se.aux = e.aux;
            se.nv = e.nv;
            sh.ent[lane] = se;
            if (!is_stroke && my_stage < (uint32_t)STAGECAP) {
                /* the fill's 16 coverage words (one 64-byte line), requested by the lane that staged the entry — beside the
                 * strokes' constants, not a round trip behind them */
                const uint4* OSMT_R src = reinterpret_cast<const uint4*>(g_fmask + e.arena);
                uint4* dst = reinterpret_cast<uint4*>(&SH_FMASK(sh, my_stage)[0]);
                const uint4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
                dst[0] = w0;
                dst[1] = w1;
                dst[2] = w2;
                dst[3] = w3;
            }
            if (is_stroke && my_stage < (uint32_t)STAGECAP) {
                /* constants of the across test (second round trip, in parallel for all strokes of the chunk) */
                const osmt_stroke_aux* __restrict__ sa = &g_aux[e.aux];
                StrokeConst kc;
                kc.ff0 = sa->ff0;
                kc.ft0 = sa->ft0;
                kc.fd0 = sa->fd0;
                kc.rfd0 = sa->rfd0;
                kc.mul0 = sa->mul0;
                kc.flags = stroke_flags(sa);
                kc._pad = 0;
                SH_SCONST(sh, my_stage) = kc;
            }
        }
        const bool any_stroke = sbal != 0ull;
        if (any_stroke && !plane_clean) {
            static_assert((PLANE_STRIDE * SUBH) % NTHREADS == 0, "the plane is cleared in whole wave strides");
            const uint32_t t_ = fresh_lane();
#pragma unroll
            for (uint32_t i = 0; i < PLANE_STRIDE * SUBH; i += NTHREADS) sh.plane[i + t_] = 0ull;
            plane_clean = true;
        }
        __syncthreads();

        uint32_t g0 = 0;
#if defined(OSMT_ABL) && OSMT_ABL == 7
        if (total < 1000u) g0 = total; /* ablation: the chunk is staged, no group is filtered or drawn */
#endif
        while (g0 < total) {
        /* ---- group = consecutive list entries whose stroke slots fit in the SEGCAP lanes of ONE filter pass; an op with
         * more slots forms a group of its own and is filtered SEGCAP slots at a time ---- */
        uint32_t gend = total, V = 0;
        bool big = false;
        uint32_t s_before = 0u;           /* slots of the chunk's entries in front of the group */
        if (any_stroke) {
            s_before = g0 ? (uint32_t)__builtin_amdgcn_readlane((int)nv_incl, (int)g0 - 1) : 0u;
            const unsigned long long over = __ballot(lane >= g0 && lane < total && nv_incl - s_before > (uint32_t)FILTCAP);
            if (over) gend = (uint32_t)__builtin_ctzll(over);
            if (gend == g0) { /* the first entry alone has more slots than one filter pass looks at */
                OSMT_DBG(if (lane == 0) sh.dbg[7] += 1u;)
                big = true;
                gend = g0 + 1u;
            } else {
                V = (uint32_t)__builtin_amdgcn_readlane((int)nv_incl, (int)gend - 1) - s_before;
            }
        }
        if (V) {
            OSMT_DBG(if (lane == 0) sh.dbg[3] += 1u;)
            /* ---- filter pass of the GROUP: every slot of every stroke entry of the group is looked at once, FILTCAP slots
             * (four rounds of 64 lanes, all key loads in flight together); the records of THIS sub-tile are compacted in
             * slot order (= op order, segment order).  Round 3 looked at SEGCAP slots per pass: an op of 90 slots — five
             * edges and two stubs with windows of a dozen sub-tiles — took three passes (key -> record -> barrier each)
             * and walked the two or three records it kept in up to three under-filled walks. ---- */
            uint8_t* const mark = reinterpret_cast<uint8_t*>(sh.seg);   /* mark[s]: list entry whose slots start at virtual slot s */
            uint32_t* const tmp = reinterpret_cast<uint32_t*>(sh.der);  /* [c]: arena slot of kept record c, [SEGCAP + c]: its item count | cap flag */
            const uint32_t t_ = fresh_lane();
#pragma unroll
            for (uint32_t i = 0; i < FILTCAP; i += 256u) reinterpret_cast<uint32_t*>(mark + i)[t_] = 0xFFFFFFFFu;
            __syncthreads();
            /* exclusive prefix = the inclusive one of the lane below (wave_shr:1; lane 0 keeps the 0) */
            const uint32_t excl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nv_incl, 0x138, 0xF, 0xF, false);
            const bool mine = t_ >= g0 && t_ < gend && nv_incl != excl;
            if (mine) mark[excl - s_before] = (uint8_t)t_;
            __syncthreads();
            constexpr uint32_t NR = FILTCAP / 64u;
            uint32_t ridx[NR];
            uint2 key[NR];
            unsigned long long bal[NR];
            uint32_t carry_pos = 0u, carry_ent = g0; /* the last entry start in the rounds so far */
#pragma unroll
            for (uint32_t r = 0; r < NR; ++r) {
                ridx[r] = 0xFFFFFFFFu;
                key[r] = make_uint2(0xFFFFFFFFu, 0u);
                if (r * 64u < V) { /* uniform */
                    const uint32_t vs = r * 64u + t_;
                    const unsigned long long st = __ballot(mark[vs] != 0xFFu);
                    const unsigned long long upto = (t_ == 63u) ? ~0ull : ((2ull << t_) - 1ull);
                    const unsigned long long m = st & upto;
                    uint32_t s0 = carry_pos, en = carry_ent;
                    if (m) {
                        s0 = r * 64u + 63u - (uint32_t)__builtin_clzll(m);
                        en = mark[s0];
                    }
                    if (vs < V) {
                        ridx[r] = sh.ent[en].arena + (vs - s0);
                        key[r] = g_skey[ridx[r]]; /* (sub-tile, item count | cap flag << 31); hole: sub-tile 0xFFFFFFFF */
                    }
                    if (st) {
                        carry_pos = r * 64u + 63u - (uint32_t)__builtin_clzll(st);
                        carry_ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)mark[carry_pos]);
                    }
                }
            }
            uint32_t kept = 0u;
#pragma unroll
            for (uint32_t r = 0; r < NR; ++r) {
                bal[r] = 0ull;
                if (r * 64u >= V) continue; /* uniform */
                const bool keep = key[r].x == sub && (key[r].y & 0x7FFFFFFFu) != 0u;
                bal[r] = __ballot(keep);
                const uint32_t c = kept + (uint32_t)__popcll(bal[r] & ((1ull << fresh_lane()) - 1ull));
                if (keep && c < (uint32_t)SEGCAP) {
                    tmp[c] = ridx[r];
                    tmp[(uint32_t)SEGCAP + c] = key[r].y;
                }
                kept += (uint32_t)__popcll(bal[r]);
            }
            /* lane e: kept records in front of entry e's slots, and among them.  A bound lies in one round: that round's
             * ballot and the count of the rounds before it are selected per lane (a popcount per bound, not one per round) */
            uint32_t slot0_v, nslot_v;
            {
                auto kept_below = [&](uint32_t x) { /* kept records among virtual slots [0, x), x <= V */
                    const uint32_t rr = x >> 6;
                    unsigned long long bsel = 0ull;
                    uint32_t csel = kept, cum = 0u;
#pragma unroll
                    for (uint32_t r = 0; r < NR; ++r) {
                        if (rr == r) {
                            bsel = bal[r];
                            csel = cum;
                        }
                        cum += (uint32_t)__popcll(bal[r]);
                    }
                    return csel + (uint32_t)__popcll(bsel & ((1ull << (x & 63u)) - 1ull));
                };
                const uint32_t na = kept_below(mine ? excl - s_before : 0u);
                const uint32_t nb = kept_below(mine ? nv_incl - s_before : 0u);
                slot0_v = na;
                nslot_v = nb - na;
            }
            if (kept > (uint32_t)SEGCAP) { /* more records than the LDS holds: the group ends in front of the entry that does not fit */
                const unsigned long long ov = __ballot(slot0_v + nslot_v > (uint32_t)SEGCAP);
                const uint32_t e_ov = (uint32_t)__builtin_ctzll(ov);
                if (e_ov == g0) { /* an op with more than SEGCAP records in ONE sub-tile: filtered and walked SEGCAP slots at a time */
                    OSMT_DBG(if (lane == 0) sh.dbg[6] += 1u;)
                    big = true;
                    gend = g0 + 1u;
                    kept = 0u;
                } else {
                    OSMT_DBG(if (lane == 0) sh.dbg[4] += 1u;)
                    gend = e_ov;
                    kept = (uint32_t)__builtin_amdgcn_readlane((int)slot0_v, (int)e_ov);
                }
            }
            /* where an entry's records sit among the compacted ones replaces its arena position, which nothing needs any more */
            if (mine && t_ < gend && !big) {
                sh.ent[t_].arena = slot0_v;
                sh.ent[t_].nv = nslot_v;
            }
            __syncthreads();
            {
                const uint32_t c = fresh_lane();
                uint32_t ky = 0u, at = 0u;
                if (c < kept) {
                    ky = tmp[(uint32_t)SEGCAP + c];
                    at = tmp[c];
                }
                const uint32_t incl = wave_incl_scan(ky & 0x7FFFFFFFu); /* inclusive prefix of the item counts */
                __syncthreads(); /* tmp and mark are read: their memory takes the records now */
                if (c < kept) {
                    sh.pre[c] = incl;
                    const osmt_srec r = g_srec[at];
                    sh.seg[c] = r;
                    sh.der[c] = seg_derive(r, (ky >> 31) != 0u);
                }
            }
            __syncthreads();
        }
        unsigned long long gbal = 0ull; /* big ops: lanes of the filter pass holding a record of this sub-tile */

        for (uint32_t li = g0; li < gend; ++li) {
            const StagedEnt& en = sh.ent[li];
            const uint32_t ks = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.kind_stage);
            const uint32_t kind = ks & 255u, stage = ks >> 8;
            const double cop = en.op; /* a uniform value in a vector register */
#if defined(OSMT_ABL) && OSMT_ABL == 6
            if (kind != 77u) continue; /* ablation: lists are staged, nothing is drawn */
#endif
#if defined(OSMT_ABL) && OSMT_ABL == 3
            if (kind == OSMT_OP_STROKE) continue; /* ablation: no stroke work at all */
#endif
#if defined(OSMT_ABL) && OSMT_ABL == 4
            if (kind != OSMT_OP_STROKE) continue; /* ablation: no fill work */
#endif
            if (kind == OSMT_OP_STROKE) {
                /* ---------------- draw_lines (line.rs:9-61) ---------------- */
                OSMT_DBG(if (lane == 0) sh.dbg[0] += 1u;)
                const uint32_t aux_i = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.aux);
                const osmt_stroke_aux* __restrict__ sa = &g_aux[aux_i];
                StrokeConst kc;
                if (stage != 255u) {
                    kc = SH_SCONST(sh, stage);
                } else { /* more than STAGECAP strokes in one chunk of one sub-tile: read the table directly */
                    kc.ff0 = sa->ff0;
                    kc.ft0 = sa->ft0;
                    kc.fd0 = sa->fd0;
                    kc.rfd0 = sa->rfd0;
                    kc.mul0 = sa->mul0;
                    kc.flags = stroke_flags(sa);
                    kc._pad = 0;
                }
                const uint32_t sflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)kc.flags);
                if (sflags & STROKE_DASH_FAST) { /* the op's DashSegments into LDS: the walk reads them per pixel */
                    const double* OSMT_R src = reinterpret_cast<const double*>(late_args()->dseg + (size_t)aux_i * OSMT_MAX_DASH_SEGS);
                    static_assert(sizeof(osmt_dash_seg) == 9 * sizeof(double), "nine doubles per DashSegment");
                    const uint32_t nd = (uint32_t)sa->main_n_segs * (uint32_t)DTAB_F;
                    for (uint32_t l = fresh_lane(); l < nd; l += 64u) {
                        const uint32_t sg = l / (uint32_t)DTAB_F, f = l % (uint32_t)DTAB_F;
                        (&sh.dtab[0][0])[l] = src[sg * 9u + (f < 5u ? f : f + 2u)];
                    }
                    __syncthreads();
                }
                uint32_t n_rounds = 1u, big_cap = 0u, arena = 0u;
                if (big) {
                    big_cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.nv);
                    arena = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.arena);
                    n_rounds = (big_cap + (uint32_t)SEGCAP - 1u) / (uint32_t)SEGCAP;
                }
                for (uint32_t round = 0; round < n_rounds; ++round) {
                    uint32_t slot0 = 0u, nslot = 0u;
                    if (big) {
                        /* ---- filter pass of ONE big op, SEGCAP of its slots per round ---- */
                        __syncthreads(); /* previous round's records are consumed */
                        uint32_t ridx = 0xFFFFFFFFu;
                        const uint32_t v = round * (uint32_t)SEGCAP + lane;
                        if (lane < (uint32_t)SEGCAP && v < big_cap) ridx = arena + v;
                        uint32_t cnt = 0, is_cap = 0;
                        if (ridx != 0xFFFFFFFFu) {
                            const uint2 key = g_skey[ridx];
                            if (key.x == sub) {
                                cnt = key.y & 0x7FFFFFFFu;
                                is_cap = key.y >> 31;
                            }
                        }
                        gbal = __ballot(cnt > 0u);
                        const uint32_t incl = wave_incl_scan(cnt);
                        if (cnt > 0u) {
                            const uint32_t slot = (uint32_t)__popcll(gbal & ((1ull << fresh_lane()) - 1ull));
                            const osmt_srec r = g_srec[ridx];
                            sh.seg[slot] = r;
                            sh.der[slot] = seg_derive(r, is_cap != 0u);
                            sh.pre[slot] = incl;
                        }
                        __syncthreads();
                        nslot = (uint32_t)__popcll(gbal);
                    } else {
                        slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.arena);
                        nslot = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.nv);
                    }
                    {
                        if (nslot) {
                            const uint32_t item_lo = slot0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.pre[slot0 - 1u]) : 0u;
                            const uint32_t item_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.pre[slot0 + nslot - 1u]);
#if defined(OSMT_ABL) && (OSMT_ABL == 1 || OSMT_ABL == 2)
                            (void)item_lo; (void)item_hi; /* ablation: the runs are not walked */
#else
                            walk_items(sh, lane, slot0, nslot, item_lo, item_lo, item_hi, kc, sflags, sa, cop, rc);
#endif
                        }
                    }
                }
                __syncthreads();
                /* blend this generation's pending pixels (tile_pixels.rs:205-223) */
                /* an untouched cell holds alpha = +0.0, and 0*c + (1 - 0)*old == old exactly: blend_stroke_masked */
#if defined(OSMT_ABL) && OSMT_ABL == 2
                if (false) /* ablation: neither walked nor blended */
#endif
                {
                    /* the colour is read only now: held across the walk it costs six registers where the kernel has none to spare */
                    const StagedEnt& eb = sh.ent[late_index(li)];
                    const double c0 = eb.c0, c1 = eb.c1, c2 = eb.c2;
                    const uint32_t t_ = fresh_lane();
                    const uint32_t cell0 = (t_ / SUB) * PLANE_STRIDE + (t_ & (SUB - 1));
#pragma unroll
                    for (int j = 0; j < PXT; ++j) {
                        const uint32_t idx = cell0 + (uint32_t)j * ROWSTEP * PLANE_STRIDE;
                        const double al = __longlong_as_double((long long)sh.plane[idx]);
                        sh.plane[idx] = 0ull;
                        blend_stroke_masked(acc[AJ(j)][0], acc[AJ(j)][1], acc[AJ(j)][2], al, c0, c1, c2);
                    }
                }
                __syncthreads(); /* the plane is reused by the next op */
            }
            /* NOT an else: with two arms that both define all 48 accumulators the register allocator keeps the incoming
             * values alive through the first arm (the structurised flow runs the arms one after the other) and copies
             * them — 24 moves per fill visit in round 3's kernel.  Two independent if-blocks update them in place; the
             * laundered copy of `kind` keeps the compiler from fusing the blocks again. */
            if (late_index(kind) != OSMT_OP_STROKE) {
                /* ---------------- fill_contour (fill.rs:16-47): coverage words from k_fill_rows ----------------
                 * pixel j of lane t is column t % 32 of row t / 32 + 2j: the lanes that own a covered pixel j are the bits
                 * of (word of row 2j) | (word of row 2j + 1) << 32 — a ready-made execution mask */
                OSMT_DBG(if (lane == 0) sh.dbg[5] += 1u;)
                uint32_t w_[SUBH];
                if (stage != 255u) {
                    const uint4* OSMT_R lw = reinterpret_cast<const uint4*>(&SH_FMASK(sh, stage)[0]);
#pragma unroll
                    for (int q = 0; q < SUBH / 4; ++q) {
                        const uint4 v = lw[q];
                        w_[4 * q + 0] = v.x;
                        w_[4 * q + 1] = v.y;
                        w_[4 * q + 2] = v.z;
                        w_[4 * q + 3] = v.w;
                    }
                } else { /* more than STAGECAP fills in one chunk of one sub-tile: the arena directly (a uniform address) */
                    const uint32_t* OSMT_R mw = g_fmask + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)en.arena);
#pragma unroll
                    for (int q = 0; q < SUBH; ++q) w_[q] = mw[q];
                }
                unsigned long long m_[PXT];
#pragma unroll
                for (int j = 0; j < PXT; ++j)
                    m_[j] = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)w_[2 * j]) |
                            ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)w_[2 * j + 1]) << 32);
                /* Filler::Color: the wave-uniform colour the staging lane prepared */
                const bool image = kind != OSMT_OP_FILL_COLOR;
                if (!image) {
                    const double c0 = en.c0, c1 = en.c1, c2 = en.c2;
#pragma unroll
                    for (int j = 0; j < PXT; ++j) blend_masked(acc[AJ(j)][0], acc[AJ(j)][1], acc[AJ(j)][2], c0, c1, c2, cop, m_[j]);
                }
                /* Filler::Image: icon.get(x % w, y % h) per pixel, the opacity ignored (fill.rs:36-40) — again its own
                 * if-block, through the same masked blend */
                if (late_index((uint32_t)__builtin_amdgcn_readfirstlane(image ? 1 : 0))) {
                    const uint32_t img = (uint32_t)__builtin_amdgcn_readfirstlane((int)en.aux);
                    const osmt_raster_args* la = late_args();
                    const uint32_t n_images = (uint32_t)__builtin_amdgcn_readfirstlane((int)la->n_images);
                    if (img < n_images) { /* an unknown image id draws nothing */
                        const osmt_image_desc im = la->images[img];
                        const double4* __restrict__ ipx = la->image_pool + im.offset;
                        const uint32_t t_ = fresh_lane();
                        const uint32_t ix = ((uint32_t)rc.x0 + (t_ & (SUB - 1))) % im.width;
                        const uint32_t iy0 = (uint32_t)rc.y0 + t_ / SUB;
#pragma unroll
                        for (int j = 0; j < PXT; ++j) {
                            /* one pixel at a time (the scheduling barrier keeps the eight icon loads from being hoisted
                             * together: eight addresses and pixels in flight cost the whole kernel ~45 registers — a wave
                             * per SIMD — for a rare op) */
                            double s0 = 0.0, s1 = 0.0, s2 = 0.0, kk = 1.0;
                            if (__builtin_amdgcn_inverse_ballot_w64(m_[j])) {
                                const double4 c = ipx[(size_t)((iy0 + (uint32_t)j * (uint32_t)ROWSTEP) % im.height) * im.width + ix];
                                s0 = c.x;
                                s1 = c.y;
                                s2 = c.z;
                                kk = 1.0 - c.w;
                            }
                            blend_masked(acc[AJ(j)][0], acc[AJ(j)][1], acc[AJ(j)][2], s0, s1, s2, kk, m_[j]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
        __syncthreads(); /* the group's records are rewritten by the next group */
        g0 = gend;
        }
    }

    /* ---- label pass, blend_unfinished_pixels(true) (tile_pixels.rs:154-158,205-223) ----------
     * k_label_resolve has decided which labels succeeded; succeeded labels never share a pixel
     * (set_label_pixel refuses the second one), so every pixel is blended at most once and the
     * order of the loop does not matter.  Inside one label the text's pixels (total > 0) were
     * written after the icon's and replace them (labeler.rs:29-31). */
    if (LABELS) {
        const osmt_raster_args* la = late_args();
        const osmt_labelinfo* OSMT_R g_lab = la->labels.info;
        const double* OSMT_R g_lab_plane = la->labels.plane;
        const double4* OSMT_R g_image_pool = la->image_pool;
        const osmt_tile_label* OSMT_R tl = la->labels.tile_labels + la->labels.job_label_off[tile];
        const uint32_t n_tl = la->labels.tile_label_cnt[tile];
        /* the tile's survivors are tested against the sub-tile 64 at a time (one load, one ballot): only the few
         * that reach into it are walked */
        for (uint32_t kb = 0; kb < n_tl; kb += 64u) {
          const uint32_t kk = kb + lane;
          osmt_tile_label e = {};
          if (kk < n_tl) e = tl[kk];
          unsigned long long hm = __ballot(kk < n_tl && !(e.x0 > rc.x1 || e.x1 < rc.x0 || e.y0 > rc.y1 || e.y1 < rc.y0));
          while (hm) {
            const uint32_t hj = (uint32_t)__builtin_ctzll(hm);
            hm &= hm - 1ull;
            const osmt_labelinfo* OSMT_R li = g_lab + (uint32_t)__builtin_amdgcn_readlane((int)e.label, (int)hj);
            const int32_t ry0 = li->ry0, ry1 = li->ry1, cx0 = li->cx0;
            const int32_t cx1 = cx0 + (int32_t)li->cols - 1;
            const bool text_hit = li->has_text && ry0 <= rc.y1 && ry1 >= rc.y0 && cx0 <= rc.x1 && cx1 >= rc.x0;
            const int32_t ix0 = li->icon_x, iy0 = li->icon_y;
            const int32_t iw = (int32_t)li->icon_w, ih = (int32_t)li->icon_h;
            const bool icon_hit = iw > 0 && ix0 <= rc.x1 && ix0 + iw - 1 >= rc.x0 && iy0 <= rc.y1 && iy0 + ih - 1 >= rc.y0;
            if (!text_hit && !icon_hit) continue;
            const double cr = (double)li->color[0] / 255.0, cg = (double)li->color[1] / 255.0,
                         cb = (double)li->color[2] / 255.0;
            const double* OSMT_R plane = g_lab_plane + li->plane_off;
            const uint32_t cols = li->cols;
            const double4* OSMT_R ipx = g_image_pool + li->icon_off;
            const uint32_t lx = fresh_lane() & (SUB - 1), ly0 = fresh_lane() / SUB;
            const int32_t x = rc.x0 + (int32_t)lx;
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const int32_t y = rc.y0 + (int32_t)(ly0 + (uint32_t)j * ROWSTEP);
                double t = 0.0;
                if (text_hit && y >= ry0 && y <= ry1 && x >= cx0 && x <= cx1)
                    t = plane[(size_t)(y - ry0) * cols + (uint32_t)(x - cx0)];
                if (t > 0.0) { /* RgbaColor::from_color(&self.color, total) (rasterizer.rs:140) */
                    blend_rgb(acc[AJ(j)], t * cr, t * cg, t * cb, t);
                } else if (icon_hit && x >= ix0 && x < ix0 + iw && y >= iy0 && y < iy0 + ih) {
                    const double4 c = ipx[(size_t)(y - iy0) * (uint32_t)iw + (uint32_t)(x - ix0)];
                    blend_rgb(acc[AJ(j)], c.x, c.y, c.z, c.w);
                }
            }
          }
        }
    }

    /* ---- to_rgb_triples (tile_pixels.rs:164-181) / raw canvas ---------------- */
    void* const g_out = late_args()->out;
    const size_t g_out_tile_stride = late_args()->out_tile_stride;
    const uint32_t t_out = fresh_lane();
    const uint32_t lx_o = t_out & (SUB - 1), ly_o = t_out / SUB;
    if (!OUT_F64 && late_args()->out_rgb8) {
        /* packed RGB8: the sub-tile's 16 rows of 96 bytes go through LDS (the alpha plane is free now) — three byte
         * stores per pixel in, six dwords per lane out, every row a contiguous 96-byte piece of the tile's row — instead
         * of a second kernel that reads the RGBA8 framebuffers back and packs them */
        uint8_t* const stg = reinterpret_cast<uint8_t*>(sh.plane);
        static_assert(sizeof(sh.plane) >= (size_t)SUB * SUBH * 3, "RGB8 staging fits the alpha plane");
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const uint32_t row = ly_o + (uint32_t)j * ROWSTEP;
            uint8_t* px = stg + (row * SUB + lx_o) * 3u;
            px[0] = (uint8_t)f64_as_u8(255.0 * acc[AJ(j)][0]);
            px[1] = (uint8_t)f64_as_u8(255.0 * acc[AJ(j)][1]);
            px[2] = (uint8_t)f64_as_u8(255.0 * acc[AJ(j)][2]);
        }
        __syncthreads();
        constexpr uint32_t ROW_DW = SUB * 3 / 4; /* 24 dwords per row */
        uint8_t* const tile_out = reinterpret_cast<uint8_t*>(g_out) + (size_t)tile * g_out_tile_stride;
#pragma unroll
        for (uint32_t q = 0; q < ROW_DW * SUBH / NTHREADS; ++q) {
            const uint32_t d = q * NTHREADS + t_out;
            const uint32_t row = d / ROW_DW, k = d % ROW_DW;
            const uint32_t v = reinterpret_cast<const uint32_t*>(stg)[d];
            __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(tile_out + ((size_t)(rc.y0 + (int32_t)row) * W + (size_t)rc.x0) * 3u + 4u * k));
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const uint32_t row = ly_o + (uint32_t)j * ROWSTEP;
        const size_t px = (size_t)(rc.y0 + (int32_t)row) * W + (size_t)(rc.x0 + (int32_t)lx_o);
        if (OUT_F64) {
            double4* out = reinterpret_cast<double4*>(g_out) + (size_t)tile * W * W + px;
            *out = make_double4(acc[AJ(j)][0], acc[AJ(j)][1], acc[AJ(j)][2], 1.0);
        } else {
            /* postdivide (tile_pixels.rs:171-175) with p.a == 1.0: val / 1.0 == val */
            const uint32_t v = f64_as_u8(255.0 * acc[AJ(j)][0]) | (f64_as_u8(255.0 * acc[AJ(j)][1]) << 8) |
                               (f64_as_u8(255.0 * acc[AJ(j)][2]) << 16) | 0xFF000000u;
            uint32_t* out = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(g_out) +
                                                        (size_t)tile * g_out_tile_stride) + px;
            /* written once, read by nobody on the device: a non-temporal store keeps the 268 MB of a launch's pixels from
             * pushing the lists, keys and records of the tiles still being drawn out of the L2s (0.644 -> 0.614 ms on config 2;
             * the same hint on the LOADS of the read-once list entries and coverage words costs 0.01 ms instead) */
#ifdef OSMT_V_PLAIN_STORE
            *out = v;
#else
            __builtin_nontemporal_store(v, out);
#endif
            OSMT_DBG(__syncthreads(); if (j == 0 && ly_o == 0 && lx_o < 8) *out = sh.dbg[lx_o];)
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Layer compositing (tile_pixels.rs:205-223 over L resident layers, then :164-181).
 * Pure HBM stream: 32*L bytes in, 4 bytes out per pixel.
 *
 * Access pattern: a wave owns 64 consecutive pixels = one 2 KiB run per layer.  Every load is
 * a fully coalesced 16 B-per-lane instruction (lane i reads bytes [16i, 16i+16) of the first
 * or the second KiB), so lane pair (2j, 2j+1) holds the two halves of pixel j (first KiB) and
 * of pixel 32+j (second KiB).  One quad_perm DPP swap per half gives lane 2j all of pixel j
 * and lane 2j+1 all of pixel 32+j; from there each lane blends one pixel strictly in layer
 * order (blend_pixel is not commutative) and stores one RGBA8 word. */
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double dpp_swap_pair(double x) {
    /* quad_perm:[1,0,3,2] = 0xB1: exchange with the neighbouring lane (lane ^ 1) */
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0xB1, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int LT, bool NT>
__global__ __launch_bounds__(256) void k_composite(const v2d* __restrict__ planes, double4 canvas, uint32_t n,
                                                   uint32_t L, uint32_t npx, uint32_t* __restrict__ out) {
    /* npx is a multiple of 64 (checked by the launcher), so a 64-pixel run never straddles tiles */
    const size_t total_runs = (size_t)n * (npx / 64u);
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave0 = ((size_t)blockIdx.x * 256u + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256u) >> 6;
    const bool odd = lane & 1u;
    for (size_t run = wave0; run < total_runs; run += nwaves) {
        const size_t t = run / (npx / 64u);
        const size_t q0 = (run - t * (npx / 64u)) * 64u; /* first pixel of the run inside tile t */
        const v2d* src = planes + ((t * (size_t)L) * npx + q0) * 2u + lane;
        const size_t layer_stride = (size_t)npx * 2u; /* v2d units */
        double d[4] = {canvas.x, canvas.y, canvas.z, canvas.w};
        auto blend_layer = [&](v2d h0, v2d h1) {
            /* even lane: h0 = (r,g) of pixel j, h1 = (r,g) of pixel 32+j
             * odd  lane: h0 = (b,a) of pixel j, h1 = (b,a) of pixel 32+j */
            const v2d give = odd ? h0 : h1; /* what the partner lane needs */
            v2d got;
            got.x = dpp_swap_pair(give.x);
            got.y = dpp_swap_pair(give.y);
            const v2d rg = odd ? got : h0;
            const v2d ba = odd ? h1 : got;
            blend_px(d, rg.x, rg.y, ba.x, ba.y);
        };
        if (LT > 0) {
            v2d h0[LT > 0 ? LT : 1], h1[LT > 0 ? LT : 1];
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                if (NT) {
                    h0[l] = __builtin_nontemporal_load(src + (size_t)l * layer_stride);
                    h1[l] = __builtin_nontemporal_load(src + (size_t)l * layer_stride + 64);
                } else {
                    h0[l] = src[(size_t)l * layer_stride];
                    h1[l] = src[(size_t)l * layer_stride + 64];
                }
            }
#pragma unroll
            for (int l = 0; l < LT; ++l) blend_layer(h0[l], h1[l]);
        } else {
            for (uint32_t l = 0; l < L; ++l) {
                const v2d a0 = __builtin_nontemporal_load(src + (size_t)l * layer_stride);
                const v2d a1 = __builtin_nontemporal_load(src + (size_t)l * layer_stride + 64);
                blend_layer(a0, a1);
            }
        }
        const double a = d[3];
        const double mr = (a == 0.0) ? 0.0 : d[0] / a;
        const double mg = (a == 0.0) ? 0.0 : d[1] / a;
        const double mb = (a == 0.0) ? 0.0 : d[2] / a;
        const size_t px = t * npx + q0 + (lane >> 1) + (odd ? 32u : 0u);
        out[px] = f64_as_u8(255.0 * mr) | (f64_as_u8(255.0 * mg) << 8) | (f64_as_u8(255.0 * mb) << 16) | 0xFF000000u;
    }
}

/* osmt_hbm_copy_probe: the plain streaming copy every HBM fraction is compared with */
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
/* four independent 16-byte loads in flight per lane and trip; blocks own contiguous 16 KiB pieces */
__global__ __launch_bounds__(256) void k_copy16(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 1024u;
    size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x;
    for (; i + 768u < n; i += stride) {
        const v4u a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256u);
        const v4u c = __builtin_nontemporal_load(src + i + 512u), d = __builtin_nontemporal_load(src + i + 768u);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + 256u);
        __builtin_nontemporal_store(c, dst + i + 512u);
        __builtin_nontemporal_store(d, dst + i + 768u);
    }
    for (uint32_t k = 0; k < 4u; ++k)
        if (i + 256u * k < n && i + 768u >= n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i + 256u * k), dst + i + 256u * k);
}
/* the read half alone: every lane folds what it loads into one word and the block leaves a single store behind */
__global__ __launch_bounds__(256) void k_read16(const v4u* __restrict__ src, uint32_t* __restrict__ sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * 1024u;
    v4u acc = {0u, 0u, 0u, 0u};
    size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x;
    for (; i + 768u < n; i += stride) {
        const v4u a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256u);
        const v4u c = __builtin_nontemporal_load(src + i + 512u), d = __builtin_nontemporal_load(src + i + 768u);
        acc ^= a ^ b ^ c ^ d;
    }
    for (uint32_t k = 0; k < 4u; ++k)
        if (i + 256u * k < n && i + 768u >= n) acc ^= __builtin_nontemporal_load(src + i + 256u * k);
    const uint32_t w = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (w == 0x9E3779B9u) sink[blockIdx.x] = w; /* data-dependent, practically never taken: keeps the loads alive */
}

}  // namespace

hipError_t osmt_launch_copy16(const void* src, void* dst, size_t n16, bool read_only, hipStream_t st) {
    if (n16 == 0) return hipSuccess;
    const size_t blocks = (n16 + 1023) / 1024;
    const uint32_t grid = (uint32_t)(blocks < 256 * 16 ? blocks : 256 * 16); /* 256 CUs x 16 blocks, grid-stride beyond */
    if (read_only)
        hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, st, reinterpret_cast<const v4u*>(src), reinterpret_cast<uint32_t*>(dst), n16);
    else
        hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, reinterpret_cast<const v4u*>(src), reinterpret_cast<v4u*>(dst), n16);
    return hipGetLastError();
}

/* ---- launchers (C++ internal interface, see osmt_internal.h) ---------------- */
hipError_t osmt_launch_project(const osmt_tile_job* jobs, const uint32_t* pt_job, const double* latlon, const uint32_t* refs,
                               uint32_t n_pts, double scale, int32_t* pts, hipStream_t st, uint32_t* zero, size_t n_zero) {
    const size_t n_thr = n_pts > n_zero ? n_pts : n_zero;
    if (n_thr == 0) return hipSuccess;
    hipLaunchKernelGGL(k_project, dim3((uint32_t)((n_thr + 255u) / 256u)), dim3(256), 0, st, jobs, pt_job,
                       reinterpret_cast<const double2*>(latlon), refs, n_pts, scale, reinterpret_cast<int2*>(pts), zero, (uint32_t)n_zero);
    return hipGetLastError();
}

size_t osmt_prepass_zero_words(const osmt_prepass_args& a) {
    const uint32_t Wt = OSMT_TILE_SIZE * a.scale;
    const size_t n_cnt = (a.fmask_cap || a.srec_cap) ? (size_t)a.n_jobs * (Wt / SUB) * (Wt / SUBH) : 0;
    return 4 * sizeof(unsigned long long) / sizeof(uint32_t) + n_cnt;
}

hipError_t osmt_launch_ptjob(const osmt_tile_job* jobs, uint32_t n_jobs, uint32_t* pt_job, uint32_t n_pts, hipStream_t st) {
    if (n_pts == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(pt_job, 0xFF, (size_t)n_pts * 4, st);
    if (e != hipSuccess) return e;
    if (n_jobs) hipLaunchKernelGGL(k_ptjob, dim3(n_jobs), dim3(256), 0, st, jobs, pt_job);
    return hipGetLastError();
}

hipError_t osmt_launch_project_single(const double* latlon, uint32_t n, uint32_t zoom, uint32_t tx, uint32_t ty,
                                      double scale, int32_t* pts, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_project_single, dim3((n + 255u) / 256u), dim3(256), 0, st,
                       reinterpret_cast<const double2*>(latlon), n, zoom, tx, ty, scale, reinterpret_cast<int2*>(pts));
    return hipGetLastError();
}

hipError_t osmt_launch_prepass(const osmt_prepass_args& a, hipStream_t st, bool zeroed) {
    /* the three cursors and, right behind them, the per-sub-tile list counts (zeroed: k_project of this step has done it) */
    if (!zeroed) {
        const hipError_t e = hipMemsetAsync(a.cursors, 0, osmt_prepass_zero_words(a) * sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
    }
    if (a.n_ops)
        hipLaunchKernelGGL(k_opinfo, dim3((a.n_ops + OPINFO_THREADS * OPINFO_WAVES - 1u) / (OPINFO_THREADS * OPINFO_WAVES)),
                           dim3(OPINFO_THREADS * OPINFO_WAVES), 0, st, a);
    if (a.fmask_cap == 0 && a.srec_cap == 0) return hipGetLastError(); /* sizing pass */
    const uint32_t n_vblk = (a.n_vsegs + 63u) / 64u;
    if (a.n_ops)
        hipLaunchKernelGGL(k_prebin, dim3(n_vblk + (a.n_ops + FILL_GROUP - 1u) / FILL_GROUP), dim3(64), 0, st, a.ops, a.n_ops, a.info, a.rings, a.pts,
                           a.op_blk, a.blk, a.n_vsegs, n_vblk,
                           a.scale, a.sub_rows, a.submask, a.fmask, a.srec, a.skey, a.op_job, a.cnt, a.vseg);
    /* lists only for tiles with more than OSMT_FOLD_MAX_OPS ops (k_raster's waves put the others' together themselves) */
    if (a.n_jobs && (a.fold_max_ops == 0u || a.max_job_ops > a.fold_max_ops))
        hipLaunchKernelGGL(k_sublist, dim3(a.n_jobs), dim3(SUBLIST_THREADS), 0, st, a.jobs, a.scale, a.info, a.submask, a.sub_rows, a.cnt,
                           a.cursors + 2, a.hdr, a.ent, a.ent_cap, a.err, a.fold_max_ops);
    return hipGetLastError();
}

hipError_t osmt_launch_raster(const osmt_raster_args& a, bool out_f64, hipStream_t st) {
    if (a.n_jobs == 0) return hipSuccess;
    const uint32_t W = OSMT_TILE_SIZE * a.scale;
    const uint32_t nsub = (W / SUB) * (W / SUBH);
    const uint32_t groups = (a.n_jobs + 7u) / 8u;
    const dim3 grid(groups * 8u * nsub);
#define OSMT_LAUNCH_RASTER(F64, LAB)                                                                \
    do {                                                                                            \
        if (a.fold_max_ops)                                                                         \
            hipLaunchKernelGGL((k_raster<F64, LAB, true>), grid, dim3(NTHREADS), 0, st, a);  \
        else                                                                                        \
            hipLaunchKernelGGL((k_raster<F64, LAB, false>), grid, dim3(NTHREADS), 0, st, a); \
    } while (0)
    if (out_f64) /* the raw canvas is the one BEFORE labels (osmt_render_scene_f64) */
        OSMT_LAUNCH_RASTER(true, false);
    else if (a.labels.info)
        OSMT_LAUNCH_RASTER(false, true);
    else
        OSMT_LAUNCH_RASTER(false, false);
#undef OSMT_LAUNCH_RASTER
    return hipGetLastError();
}

#ifndef OSMT_V_COMP_NT
#define OSMT_V_COMP_NT 1
#endif
#ifndef OSMT_V_COMP_BLOCKS
#define OSMT_V_COMP_BLOCKS 16
#endif
hipError_t osmt_launch_composite(const void* planes, const double canvas[4], uint32_t n, uint32_t L, uint32_t npx,
                                 void* out, hipStream_t st) {
    const size_t total = (size_t)n * npx;
    if (total == 0) return hipSuccess;
    if (npx % 64u) return hipErrorInvalidValue; /* W*H must be a multiple of the wave size */
    const double4 cv = make_double4(canvas[0], canvas[1], canvas[2], canvas[3]);
    size_t blocks = (total + 255) / 256;
    const size_t cap = 256 * OSMT_V_COMP_BLOCKS; /* 256 CUs x resident blocks, grid-stride beyond */
    if (blocks > cap) blocks = cap;
    const v2d* p = reinterpret_cast<const v2d*>(planes);
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    constexpr bool NT = OSMT_V_COMP_NT != 0;
    if (L == 8)
        hipLaunchKernelGGL((k_composite<8, NT>), dim3((uint32_t)blocks), dim3(256), 0, st, p, cv, n, L, npx, o);
    else if (L == 4)
        hipLaunchKernelGGL((k_composite<4, NT>), dim3((uint32_t)blocks), dim3(256), 0, st, p, cv, n, L, npx, o);
    else
        hipLaunchKernelGGL((k_composite<0, NT>), dim3((uint32_t)blocks), dim3(256), 0, st, p, cv, n, L, npx, o);
    return hipGetLastError();
}
