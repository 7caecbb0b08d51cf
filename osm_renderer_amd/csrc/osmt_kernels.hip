/*
 * osmt_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the tile hot path.
 *
 *   k_project    Point::from_node                 (src/tile.rs:88-106, src/draw/point.rs:11-19)
 *   k_opinfo     per-op pixel extents, traveled distances and dash tables
 *                (src/draw/line.rs:21-33, src/draw/opacity_calculator.rs:16-30,98-143)
 *   k_raster     fill_contour + draw_lines + set_pixel/blend + to_rgb_triples, fused per
 *                32x32-pixel sub-tile (src/draw/fill.rs, line.rs, opacity_calculator.rs,
 *                tile_pixels.rs, drawer.rs:133-219)
 *   k_composite  blend_pixel over L resident layers + to_rgb_triples (tile_pixels.rs:205-223,164-181)
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
__device__ __forceinline__ uint32_t f64_as_u8(double v) { return (uint32_t)(int32_t)fmin(fmax(v, 0.0), 255.0); }

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
                                                 uint32_t n_pts, double scale, int2* __restrict__ pts) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
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

/* opacity_calculator.rs:98-143 compute_segments, for one calculator. */
__device__ void compute_segments(double hlw, const double* __restrict__ dashes, int n_dashes, int cap,
                                 osmt_dash_table* t) {
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
        t->segs[n++] = s;
    }
    t->n_segs = n;
    t->has_orig = (cap == OSMT_CAP_ROUND) ? 1 : 0;
    t->total_len = len_before;
}

/* Marks the sub-tiles whose rectangle grown by g pixels intersects the segment a-b
 * (separating axes: x, y and the segment's normal; exact in int64). */
__device__ void mark_segment(uint32_t* __restrict__ sm, int32_t n_sub_x, int32_t n_sub_y, int32_t ax, int32_t ay,
                             int32_t bx, int32_t by, int32_t g) {
    const int32_t sx0 = max((min(ax, bx) - g) >> 5, 0), sx1 = min((max(ax, bx) + g) >> 5, n_sub_x - 1);
    const int32_t sy0 = max((min(ay, by) - g) >> OSMT_SUB_H_LOG2, 0);
    const int32_t sy1 = min((max(ay, by) + g) >> OSMT_SUB_H_LOG2, n_sub_y - 1);
    if (sx0 > sx1 || sy0 > sy1) return;
    const int64_t dx = (int64_t)bx - ax, dy = (int64_t)by - ay;
    for (int32_t sy = sy0; sy <= sy1; ++sy) {
        const int64_t y0 = (int64_t)sy * OSMT_SUB_H - g - ay, y1 = (int64_t)sy * OSMT_SUB_H + (OSMT_SUB_H - 1) + g - ay;
        uint32_t bits = 0u;
        for (int32_t sx = sx0; sx <= sx1; ++sx) {
            const int64_t x0 = (int64_t)sx * 32 - g - ax, x1 = (int64_t)sx * 32 + 31 + g - ax;
            /* cross(d, corner - a) for the four corners */
            const int64_t c00 = dx * y0 - dy * x0, c10 = dx * y0 - dy * x1;
            const int64_t c01 = dx * y1 - dy * x0, c11 = dx * y1 - dy * x1;
            const bool all_pos = c00 > 0 && c10 > 0 && c01 > 0 && c11 > 0;
            const bool all_neg = c00 < 0 && c10 < 0 && c01 < 0 && c11 < 0;
            if (!(all_pos || all_neg)) bits |= 1u << sx;
        }
        sm[sy] |= bits;
    }
}

/* Per-op pre-pass: pixel extents (for sub-tile culling), traveled distance before every
 * edge of a stroke (line.rs:31: add_traveled_distance, summed in edge order), and the two
 * dash tables of draw_lines (line.rs:21-22). One thread per op. */
__global__ __launch_bounds__(64) void k_opinfo(const osmt_op* __restrict__ ops, uint32_t n_ops,
                                               const osmt_ring* __restrict__ rings, const int2* __restrict__ pts,
                                               const double* __restrict__ dashes, const uint32_t* __restrict__ op_aux,
                                               osmt_opinfo* __restrict__ info, double* __restrict__ trav,
                                               double* __restrict__ den, osmt_stroke_aux* __restrict__ aux,
                                               uint8_t* __restrict__ opnv, const uint32_t* __restrict__ op_blk,
                                               osmt_blk_bbox* __restrict__ blk, uint32_t* __restrict__ submask,
                                               uint32_t sub_rows) {
    const uint32_t o = blockIdx.x * 64u + threadIdx.x;
    if (o >= n_ops) return;
    const osmt_op op = ops[o];
    uint32_t* __restrict__ sm = submask + (size_t)o * sub_rows;
    for (uint32_t r = 0; r < sub_rows; ++r) sm[r] = 0u;
    osmt_opinfo oi;
    oi.x0 = oi.y0 = INT32_MAX;
    oi.x1 = oi.y1 = INT32_MIN;
    oi.aux = op_aux[o];
    oi.n_edges = 0;
    oi.reach_major = 0;
    opnv[o] = 0;
    if (op.kind == OSMT_OP_NONE) {
        info[o] = oi;
        return;
    }
    double traveled = 0.0;
    uint32_t n_edges = 0;
    /* bounding boxes of the 64-edge blocks (ops with more than 64 edges only) */
    const uint32_t blk_off = op_blk[o];
    uint32_t cur_blk = 0xFFFFFFFFu;
    osmt_blk_bbox bb = {INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN};
    for (uint32_t r = 0; r < op.n_rings; ++r) {
        const osmt_ring ring = rings[op.ring_off + r];
        int2 prev = make_int2(0, 0);
        for (uint32_t i = 0; i < ring.n_pts; ++i) {
            const int2 p = pts[ring.first_pt + i];
            if (blk_off != 0xFFFFFFFFu && i > 0) {
                const uint32_t b_ = (n_edges + i - 1u) >> 6; /* running edge index / 64 */
                if (b_ != cur_blk) {
                    if (cur_blk != 0xFFFFFFFFu) blk[blk_off + cur_blk] = bb;
                    cur_blk = b_;
                    bb = {INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN};
                }
                bb.x0 = min(bb.x0, min(prev.x, p.x));
                bb.y0 = min(bb.y0, min(prev.y, p.y));
                bb.x1 = max(bb.x1, max(prev.x, p.x));
                bb.y1 = max(bb.y1, max(prev.y, p.y));
            }
            oi.x0 = min(oi.x0, p.x);
            oi.x1 = max(oi.x1, p.x);
            oi.y0 = min(oi.y0, p.y);
            oi.y1 = max(oi.y1, p.y);
            if (op.kind == OSMT_OP_STROKE) {
                if (i > 0) {
                    /* |p2 - p1| is both the traveled increment (line.rs:31) and center_dist_denom
                     * (line.rs:104: sqrt(dy*dy + dx*dx) of the absolute deltas — the same f64) */
                    const double len = point_dist(prev.x, prev.y, p.x, p.y);
                    den[ring.first_pt + i - 1] = len;
                    traveled += len;
                }
                trav[ring.first_pt + i] = traveled; /* traveled before the edge that STARTS at point i */
            }
            prev = p;
        }
        if (ring.n_pts >= 2) n_edges += ring.n_pts - 1;
    }
    if (cur_blk != 0xFFFFFFFFu) blk[blk_off + cur_blk] = bb;
    oi.n_edges = n_edges;
    if (op.kind == OSMT_OP_STROKE) {
        const double hw = op.width / 2.0;
        const double ft = fmax(hw + 0.5, 1.0);
        /* How far a SET pixel can be from the Bresenham centre its perpendicular starts at.  At walk
         * step t the pixel is t px along the minor axis and cc_t <= t*a/b + 1 px along the major
         * axis from the centre; its distance from the ideal line is >= t*len/b - a/len - |d0|, with
         * |d0| <= 0.5 for a main perpendicular and <= 1.5 for the extra one of line.rs:152-154; it
         * is set only while that distance is < ft' <= ft.  Hence t < (ft + 2.21) * b/len <= ft + 2.21
         * and cc_t < (a/len)*(ft + 2.21) + 1 <= 0.7072*(ft + 2.21) + 1. */
        int32_t reach = (int32_t)fmin(ceil(ft + 2.21), 1.0e6);
        int32_t reach_major = (int32_t)fmin(ceil(0.7072 * (ft + 2.21)), 1.0e6) + 1;
        oi.reach_major = reach_major;
        const bool caps = (op.cap == OSMT_CAP_ROUND || op.cap == OSMT_CAP_SQUARE);
        int32_t cap_reach = caps ? (int32_t)fmin(ceil(fabs(hw)), 1.0e6) + 1 : 0;
        oi.reach = reach;
        if (oi.x0 <= oi.x1) {
            oi.x0 -= reach + cap_reach;
            oi.y0 -= reach + cap_reach;
            oi.x1 += reach + cap_reach;
            oi.y1 += reach + cap_reach;
        }
        /* sub-tile coverage: a sub-tile is marked when its rectangle grown by reach+1 (the
         * Bresenham centre is within 0.5 px of the ideal segment) meets the segment; the cap
         * stubs (length <= hw + 1) are covered by growing the first/last edge's test further */
        {
            const int32_t n_sub_y = (int32_t)sub_rows, n_sub_x = (int32_t)(sub_rows * OSMT_SUB_H / OSMT_SUB_W);
            uint32_t e_seen = 0;
            for (uint32_t r = 0; r < op.n_rings; ++r) {
                const osmt_ring ring = rings[op.ring_off + r];
                for (uint32_t i = 1; i < ring.n_pts; ++i) {
                    const int2 a = pts[ring.first_pt + i - 1];
                    const int2 b = pts[ring.first_pt + i];
                    ++e_seen;
                    const bool endcap = caps && (e_seen == 1 || e_seen == n_edges);
                    const int32_t g = reach + 1 + (endcap ? cap_reach : 0);
                    mark_segment(sm, n_sub_x, n_sub_y, a.x, a.y, b.x, b.y, g);
                }
            }
        }
        {
            const uint32_t nv = n_edges + (caps ? 2u : 0u);
            opnv[o] = (op.n_rings == 1u && nv <= 64u) ? (uint8_t)nv : (uint8_t)255;
        }
        osmt_stroke_aux* sa = &aux[oi.aux];
        sa->half_width = hw;
        /* cap stubs (line.rs:33-57): only for the first / last iterated edge, only if it is not
         * degenerate (`first` is consumed by a degenerate first edge) */
        {
            osmt_cap_seg c0 = {0, 0, 0, 0, 0, 0, 1.0}, c1 = {0, 0, 0, 0, 0, 0, 1.0};
            uint32_t seen = 0;
            for (uint32_t r = 0; r < op.n_rings && caps; ++r) {
                const osmt_ring ring = rings[op.ring_off + r];
                for (uint32_t i = 1; i < ring.n_pts; ++i) {
                    ++seen;
                    if (seen != 1 && seen != n_edges) continue;
                    const int2 a = pts[ring.first_pt + i - 1];
                    const int2 b = pts[ring.first_pt + i];
                    if (a.x == b.x && a.y == b.y) continue;
                    if (seen == 1) {
                        const int2 ce = push_away_from(a, b, hw);
                        c0 = {a.x, a.y, ce.x, ce.y, 1, 0, point_dist(ce.x, ce.y, a.x, a.y)};
                    }
                    if (seen == n_edges) {
                        const int2 ce = push_away_from(b, a, hw);
                        c1 = {b.x, b.y, ce.x, ce.y, 1, 0, point_dist(ce.x, ce.y, b.x, b.y)};
                    }
                }
            }
            sa->cap_seg[0] = c0;
            sa->cap_seg[1] = c1;
        }
        sa->hlw0 = sqrt(hw * hw - 0.0 * 0.0);
        sa->ff0 = fmax(sa->hlw0 - 0.5, 0.0);
        sa->ft0 = fmax(sa->hlw0 + 0.5, 1.0);
        sa->fd0 = sa->ft0 - sa->ff0;
        sa->mul0 = fmin(2.0 * sa->hlw0, 1.0);
        const int cap_for_dashes = op.use_caps_for_dashes ? op.cap : OSMT_CAP_NONE;
        if (op.has_dashes) {
            compute_segments(hw, dashes + op.dashes_off, (int)op.n_dashes, cap_for_dashes, &sa->main);
        } else {
            sa->main.n_segs = 0;
            sa->main.has_orig = 0;
            sa->main.total_len = 0.0;
        }
        const double zero = 0.0;
        compute_segments(hw, &zero, 1, op.cap, &sa->caps);
        /* the chained (0..1) pass pushes the same segment twice; max/min over two equal
         * entries equals one entry, keep one */
        sa->caps.n_segs = 1;
    } else {
        oi.reach = 0;
        /* fills: every sub-tile of the extent (rows ytop+1..ybot only carry records) */
        if (oi.x0 <= oi.x1) {
            const int32_t n_sub_y = (int32_t)sub_rows, n_sub_x = (int32_t)(sub_rows * OSMT_SUB_H / OSMT_SUB_W);
            const int32_t sx0 = max(oi.x0 >> 5, 0), sx1 = min(oi.x1 >> 5, n_sub_x - 1);
            const int32_t sy0 = max((oi.y0 + 1) >> OSMT_SUB_H_LOG2, 0), sy1 = min(oi.y1 >> OSMT_SUB_H_LOG2, n_sub_y - 1);
            if (sx0 <= sx1) {
                const uint32_t bits = (uint32_t)((((uint64_t)1 << (sx1 - sx0 + 1)) - 1) << sx0);
                for (int32_t sy = sy0; sy <= sy1; ++sy) sm[sy] |= bits;
            }
        }
    }
    info[o] = oi;
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

/* x / d with the (very common) d == 1.0 short-cut: x / 1.0 == x exactly */
__device__ __forceinline__ double div_or_same(double x, double d) { return d == 1.0 ? x : x / d; }

/* opacity_calculator.rs:32-80 calculate (+ get_opacity_by_start_distance).  `sa` carries the
 * per-op constants of the cap_dist == 0 case (half_line_width = sqrt(h*h - 0*0) and its
 * feather terms), which is every pixel unless a Round cap shrinks the line. */
__device__ __forceinline__ bool opacity_calculate(const osmt_dash_table* __restrict__ t,
                                                  const osmt_stroke_aux* __restrict__ sa, double traveled, double cd,
                                                  double sd, double* opacity) {
    double sd_op = 1.0;
    double cap_dist = 0.0;
    const int n = t->n_segs;
    if (n > 0) {
        double dist_rem = traveled + sd;
        const double total = t->total_len;
        if (total > 0.0) dist_rem = osmt_fmod_pos(dist_rem, total); /* dist_rem >= 0: exact `%` */
        sd_op = 0.0;
        bool has = false;
        double dic = 0.0;
        const int has_orig = t->has_orig;
        for (int i = 0; i < n; ++i) {
            const osmt_dash_seg* s = &t->segs[i];
            /* :145-157 */
            if (dist_rem < s->start_from || dist_rem > s->end_to) continue;
            double base;
            if (dist_rem <= s->start_to)
                base = div_or_same(dist_rem - s->start_from, s->start_to - s->start_from);
            else if (dist_rem < s->end_from)
                base = 1.0;
            else
                base = div_or_same(s->end_to - dist_rem, s->end_to - s->end_from);
            sd_op = fmax(sd_op, s->opacity_mul * base);
            if (has_orig) { /* :159-169 */
                double d;
                if (dist_rem < s->orig_a)
                    d = s->orig_a - dist_rem;
                else if (dist_rem <= s->orig_b)
                    d = 0.0;
                else
                    d = dist_rem - s->orig_b;
                if (!has || d < dic) {
                    has = true;
                    dic = d;
                }
            }
        }
        cap_dist = has ? dic : 0.0;
    }
    double cdop;
    if (cap_dist == 0.0) {
        /* sqrt(h*h - 0*0) and its feather terms: the per-op constants (opacity_calculator.rs:36,171-176) */
        double v;
        if (cd < sa->ff0)
            v = 1.0;
        else if (cd < sa->ft0)
            v = div_or_same(sa->ft0 - cd, sa->fd0);
        else
            v = 0.0;
        cdop = sa->mul0 * v;
    } else {
        const double hw = sa->half_width;
        cdop = opacity_by_center_distance(cd, sqrt(hw * hw - cap_dist * cap_dist));
    }
    *opacity = fmin(sd_op, cdop);
    return cdop > 0.0;
}

/* ---- the fused raster kernel ---------------------------------------------- */
constexpr int SUB = OSMT_SUB_W;    /* sub-tile width in pixels (one 32-bit coverage word per row) */
constexpr int SUBH = OSMT_SUB_H;   /* sub-tile height */
#ifndef OSMT_V_NTHREADS
#define OSMT_V_NTHREADS 64
#endif
constexpr int NTHREADS = OSMT_V_NTHREADS; /* 64 = one wave per sub-tile: no cross-wave barrier anywhere */
static_assert(NTHREADS == 64, "the stroke path packs items with wave-level scans");
constexpr int PXT = SUB * SUBH / NTHREADS; /* pixels per thread */
constexpr int NBUF = NTHREADS > 64 ? 2 : 1; /* multi-wave groups double-buffer planes/masks to save a barrier */
constexpr int ROWSTEP = NTHREADS / SUB;     /* rows between a thread's consecutive pixels */
#ifndef OSMT_V_ROWCAP
#define OSMT_V_ROWCAP 16
#endif
constexpr int ROWCAP = OSMT_V_ROWCAP; /* crossing records kept per row before the slow path */

constexpr int OPCHUNK = NTHREADS;  /* ops culled per pass */

struct RowRec {
    int32_t x_min, x_max;
    uint32_t edge;
};

/* One stroke segment (an edge or a cap stub) that survived the sub-tile cull, with the step
 * ranges of its two perpendicular sides; the items of all records of an op are walked together. */
struct SegRec {
    int32_t p1x, p1y, p2x, p2y;
    double traveled;
    double denom;         /* center_dist_denom (line.rs:104) */
    int32_t k_lo0, k_n0, k_lo1, k_n1; /* main perpendiculars: steps [k_lo, k_lo + k_n) per side */
    int32_t m_lo0, n_x0, m_lo1, n_x1; /* extra perpendiculars (line.rs:152-154): events [m_lo, m_lo + n_x) per side */
    uint32_t caps_table;  /* 1: opacity_calculator_for_outer_caps (line.rs:22) */
    uint32_t count;
};
constexpr int SEGCAP = 64;

struct RasterShared {
    SegRec seg[SEGCAP];
    uint8_t opnv[OPCHUNK];          /* g_opnv of the compacted list entries */
    uint32_t grp_base[OPCHUNK + 1]; /* first virtual-segment lane of every list entry of a group */
    unsigned long long plane[NBUF][SUB * SUBH]; /* generation alpha planes (f64 bit patterns) */
    uint32_t mask[NBUF][SUBH];                  /* fill coverage per row */
    RowRec rec[SUBH][ROWCAP];
    uint32_t rowcnt[SUBH];
    uint32_t oplist[OPCHUNK];
    uint32_t wcount[(NTHREADS + 63) / 64];
};

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

/* One perpendicular run (line.rs:108-137).  PLAIN = the calculator has no dash segments
 * (get_opacity_by_start_distance returns (1.0, None) without looking at the distance,
 * opacity_calculator.rs:50-55), so long_start_dist / short_start_dist are dead values and
 * the feather terms are the per-op constants of osmt_stroke_aux. */
__device__ __forceinline__ void walk_perpendicular(const bool PLAIN, const osmt_seg& s,
                                                   const osmt_stroke_aux* __restrict__ sa,
                                                   const osmt_dash_table* __restrict__ tab, double traveled,
                                                   double initial_opacity, int32_t mn, int32_t mx, int32_t p_error,
                                                   int32_t mul, const SubRect& rc,
                                                   unsigned long long* __restrict__ plane) {
    int32_t p_mn = mx;
    int32_t p_mx = mn;
    int32_t err = mul * p_error;
    const int32_t two_a = 2 * s.a, two_b = 2 * s.b;
    int32_t px = s.swap ? p_mn : p_mx;
    int32_t py = s.swap ? p_mx : p_mn;
    /* center_dist_raw (line.rs:116-117) kept incrementally: exact int64 arithmetic */
    int64_t raw = s.numer_const + (s.sdy * (int64_t)px - s.sdx * (int64_t)py);
    const int32_t step_mx = mul * s.mn_inc;  /* p_mx += */
    const int32_t step_mn = -mul * s.mx_inc; /* p_mn += (when corrected) */
    const int64_t raw_step = s.swap ? -s.sdx * step_mx : s.sdy * step_mx;
    const int64_t raw_corr = s.swap ? s.sdy * step_mn : -s.sdx * step_mn;
    const double ff0 = sa->ff0, ft0 = sa->ft0, fd0 = sa->fd0, mul0 = sa->mul0;
    for (;;) {
        const double cd = fabs((double)raw) / s.denom;
        double op;
        bool in_line;
        if (PLAIN) {
            /* opacity_calculator.rs:171-185 with half_line_width = sqrt(h*h - 0*0) */
            double v;
            if (cd < ff0)
                v = 1.0;
            else if (cd < ft0)
                v = div_or_same(ft0 - cd, fd0);
            else
                v = 0.0;
            const double cdop = mul0 * v;
            op = fmin(1.0, cdop);
            in_line = cdop > 0.0;
        } else {
            const double ld = point_dist(px, py, s.p1x, s.p1y);
            const double sd = sqrt(fmax(ld * ld - cd * cd, 0.0));
            in_line = opacity_calculate(tab, sa, traveled, cd, sd, &op);
        }
        if (!in_line) break;
        if (px >= rc.x0 && px <= rc.x1 && py >= rc.y0 && py <= rc.y1) {
            const double alpha = initial_opacity * op;
            /* set_pixel inside one generation keeps the larger alpha (tile_pixels.rs:114-118);
             * alpha >= +0, so the u64 order of the bit pattern is the f64 order */
            atomicMax(&plane[(py - rc.y0) * SUB + (px - rc.x0)], (unsigned long long)__double_as_longlong(alpha));
        }
        /* update_error (line.rs:91-100) */
        if (err + two_a > s.b) {
            err -= two_b;
            if (s.swap) px += step_mn; else py += step_mn;
            raw += raw_corr;
        }
        err += two_a;
        if (s.swap) py += step_mx; else px += step_mx;
        raw += raw_step;
    }
}

/* Items of segment p1->p2 for this sub-tile: per side the main-axis steps [k_lo, k_lo + k_n)
 * whose perpendicular run can reach the sub-tile, and the extra-perpendicular events that fire on
 * those steps; returns the total item count (0 when culled).  The run on side `mul` moves
 * mul*mn_inc per step along the minor axis and -mul*mx_inc per correction along the major axis,
 * so the major-axis test is one-sided.  Every item is exactly ONE perpendicular run. */
__device__ __forceinline__ uint32_t seg_ranges(int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, int32_t reach,
                                               int32_t reach_major, const SubRect& rc, SegRec* q) {
    q->k_lo0 = q->k_n0 = q->k_lo1 = q->k_n1 = 0;
    q->m_lo0 = q->n_x0 = q->m_lo1 = q->n_x1 = 0;
    if (p1x == p2x && p1y == p2y) return 0u; /* line.rs:73-75 */
    /* every set pixel lies within `reach` (per axis) of the segment's box */
    if (max(p1x, p2x) + reach < rc.x0 || min(p1x, p2x) - reach > rc.x1 || max(p1y, p2y) + reach < rc.y0 ||
        min(p1y, p2y) - reach > rc.y1)
        return 0u;
    const int32_t dx = abs(p2x - p1x), dy = abs(p2y - p1y);
    const bool swap = dx > dy;
    const int32_t mx0 = swap ? p1x : p1y;
    const int32_t bmax = swap ? dx : dy, amin = swap ? dy : dx;
    const int32_t mx_inc = swap ? (p1x <= p2x ? 1 : -1) : (p1y <= p2y ? 1 : -1);
    const int32_t LO = swap ? rc.x0 : rc.y0, HI = swap ? rc.x1 : rc.y1;
    uint32_t total = 0;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int32_t mul = side ? -1 : 1;
        int32_t lo = LO, hi = HI; /* pixel major = mx_k - mul*mx_inc*cc, 0 <= cc <= reach_major */
        if (mul * mx_inc > 0) hi += reach_major; else lo -= reach_major;
        int32_t a, b;
        if (mx_inc > 0) {
            a = lo - mx0;
            b = hi - mx0;
        } else {
            a = mx0 - hi;
            b = mx0 - lo;
        }
        a = max(a, 0);
        b = min(b, bmax);
        const int32_t n = max(b - a + 1, 0);
        int32_t m_lo = 0, n_x = 0;
        if (n > 0) { /* events on steps a .. min(b, bmax-1): E(min(b, bmax-1) + 1) - E(a) */
            const int32_t e0 = osmt_extra_count(amin, bmax, a);
            const int32_t e1 = osmt_extra_count(amin, bmax, min(b, bmax - 1) + 1);
            m_lo = e0 + 1;
            n_x = max(e1 - e0, 0);
        }
        if (side == 0) {
            q->k_lo0 = a; q->k_n0 = n; q->m_lo0 = m_lo; q->n_x0 = n_x;
        } else {
            q->k_lo1 = a; q->k_n1 = n; q->m_lo1 = m_lo; q->n_x1 = n_x;
        }
        total += (uint32_t)(n + n_x);
    }
    return total;
}

/* One item of a segment record = one perpendicular run (line.rs:108-137): items [0, k_n0 + k_n1)
 * are the main perpendiculars of steps on side +1 then -1, the rest are the extra perpendiculars
 * of line.rs:152-154, located directly by osmt_extra_event. */
__device__ __forceinline__ void walk_item(const SegRec& r, uint32_t local, const bool plain_main,
                                          const osmt_stroke_aux* __restrict__ sa, double initial_opacity,
                                          int32_t reach, const SubRect& rc, unsigned long long* __restrict__ plane) {
    osmt_seg s;
    osmt_seg_setup(&s, r.p1x, r.p1y, r.p2x, r.p2y, r.denom);
    const uint32_t n_main = (uint32_t)(r.k_n0 + r.k_n1);
    int32_t k, c, pe, mul;
    if (local < n_main) {
        const bool side1 = local >= (uint32_t)r.k_n0;
        k = side1 ? r.k_lo1 + (int32_t)(local - (uint32_t)r.k_n0) : r.k_lo0 + (int32_t)local;
        mul = side1 ? -1 : 1;
        osmt_stroke_main(s.a, s.b, k, &c, &pe);
    } else {
        const uint32_t x = local - n_main;
        const bool side1 = x >= (uint32_t)r.n_x0;
        const int32_t m = side1 ? r.m_lo1 + (int32_t)(x - (uint32_t)r.n_x0) : r.m_lo0 + (int32_t)x;
        mul = side1 ? -1 : 1;
        osmt_extra_event(s.a, s.b, m, &c, &k, &pe);
    }
    const int32_t mx = s.mx0 + k * s.mx_inc;
    const int32_t mn = s.mn0 + c * s.mn_inc;
    /* pixel minor = mn + mul*mn_inc*t, 0 <= t <= reach */
    int32_t mlo = s.swap ? rc.y0 : rc.x0, mhi = s.swap ? rc.y1 : rc.x1;
    if (mul * s.mn_inc > 0) mlo -= reach; else mhi += reach;
    const bool use_caps = r.caps_table != 0u;
    if (mn >= mlo && mn <= mhi)
        walk_perpendicular(plain_main && !use_caps, s, sa, use_caps ? &sa->caps : &sa->main, r.traveled,
                           initial_opacity, mn, mx, pe, mul, rc, plane);
}

/* fill.rs:23-45 for ONE row without storing its records: stream them in (x_min, edge) order by
 * repeated minimum search and OR the paired spans that fall into [x0, x1].  Used only for rows
 * with more than ROWCAP crossings (cold; deliberately not inlined). */
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
#define OSMT_V_WAVES 3
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

/* BLOCKS: the scene has ops with more than 64 edges (osmt_blk_bbox culling compiled in); scenes
 * made of short ways only (all named configs) run the leaner instantiation. */
template <bool OUT_F64, bool BLOCKS, bool LABELS>
__global__ OSMT_RASTER_BOUNDS void k_raster(
    /* separate __restrict__ const pointers (not a struct): lets the compiler prove the display
     * list is read-only and fetch wave-uniform records with scalar loads */
    const osmt_tile_job* OSMT_R g_jobs, uint32_t g_n_jobs, uint32_t g_scale, const osmt_op* OSMT_R g_ops,
    const osmt_opinfo* OSMT_R g_info, const osmt_ring* OSMT_R g_rings, const int2* OSMT_R g_pts,
    const double* OSMT_R g_trav, const double* OSMT_R g_den, const osmt_stroke_aux* OSMT_R g_aux,
    const uint8_t* OSMT_R g_opnv, const uint32_t* OSMT_R g_op_blk, const osmt_blk_bbox* OSMT_R g_blk,
    const uint32_t* OSMT_R g_submask, uint32_t g_sub_rows, const osmt_image_desc* OSMT_R g_images,
    const double4* OSMT_R g_image_pool, uint32_t g_n_images, void* OSMT_R g_out,
    size_t g_out_tile_stride, const osmt_labelinfo* OSMT_R g_lab, const uint32_t* OSMT_R g_job_label_off,
    const osmt_tile_label* OSMT_R g_tl, const uint32_t* OSMT_R g_tl_cnt, const double* OSMT_R g_lab_plane) {
    __shared__ RasterShared sh;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t W = OSMT_TILE_SIZE * g_scale;
    const uint32_t subs_per_row = W / SUB;
    const uint32_t nsub = subs_per_row * (W / SUBH);

    /* XCD-aware block -> (tile, sub-tile): blocks b, b+8, b+16.. land on one XCD, so give
     * them the sub-tiles of the same tiles (they share that tile's display list in L2). */
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u;
    const uint32_t rest = b >> 3;
    const uint32_t tile = (rest / nsub) * 8u + xcd;
    const uint32_t sub = rest % nsub;
    if (tile >= g_n_jobs) return;

    const osmt_tile_job job = g_jobs[tile];
    SubRect rc;
    const uint32_t sub_x = sub % subs_per_row, sub_y = sub / subs_per_row;
    rc.x0 = (int32_t)(sub_x * SUB);
    rc.y0 = (int32_t)(sub_y * SUBH);
    rc.x1 = rc.x0 + SUB - 1;
    rc.y1 = rc.y0 + SUBH - 1;

    /* thread -> pixels: column lx, rows ly0 + 8*j; a wave covers two full 128-byte rows */
    const uint32_t lx = tid & (SUB - 1);
    const uint32_t ly0 = tid / SUB;

    /* tile_pixels.rs:89-93 reset.  Only r,g,b are carried: the canvas alpha starts at 1.0 and
     * blend_pixel keeps it at exactly 1.0 — fl(a + fl(1-a)*1.0) == 1.0 for every alpha in
     * [0, 2^52] (1-a is exact for a >= 0.5; below, the rounding error of 1-a is <= 2^-54 and
     * 1 + e rounds to 1.0) — so it is a constant, checked bit-for-bit by the f64 parity tests. */
    double acc[PXT][3];
    {
        double r = 0.0, g = 0.0, bl = 0.0;
        if (job.has_canvas) {
            r = 1.0 * ((double)job.canvas_rgb[0] / 255.0);
            g = 1.0 * ((double)job.canvas_rgb[1] / 255.0);
            bl = 1.0 * ((double)job.canvas_rgb[2] / 255.0);
        }
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            acc[j][0] = r;
            acc[j][1] = g;
            acc[j][2] = bl;
        }
    }
    bool plane_clean = false; /* the alpha plane is cleared when the first stroke op shows up */
    if (tid < SUBH) sh.rowcnt[tid] = 0u;
    __syncthreads();

    uint32_t buf = 0;
    for (uint32_t base = 0; base < job.n_ops; base += OPCHUNK) {
        /* ---- ordered compaction of the ops whose extent touches this sub-tile ---- */
        const uint32_t oi_idx = base + tid;
        bool hit = false;
        uint32_t my_nv = 0;
        if (oi_idx < job.n_ops) {
            hit = (g_submask[(size_t)(job.op_off + oi_idx) * g_sub_rows + sub_y] >> sub_x) & 1u;
            if (hit) my_nv = g_opnv[job.op_off + oi_idx];
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) sh.wcount[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < (NTHREADS + 63) / 64; ++w) {
            const uint32_t cnt = sh.wcount[w];
            if (w < wave) off += cnt;
            total += cnt;
        }
        if (hit) {
            const uint32_t pos = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            sh.oplist[pos] = oi_idx;
            sh.opnv[pos] = (uint8_t)my_nv;
        }
        const bool any_stroke = __ballot(hit && my_nv != 0u) != 0ull;
        if (any_stroke && !plane_clean) {
            for (uint32_t i = tid; i < NBUF * SUB * SUBH; i += NTHREADS) (&sh.plane[0][0])[i] = 0ull;
            plane_clean = true;
        }
        __syncthreads();
        total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);

        uint32_t g0 = 0;
        while (g0 < total) {
        /* ---- group = consecutive list entries whose stroke segments (edges + cap stubs) fit in the
         * 64 lanes of ONE record pass; a stroke op with more segments (or several rings) forms a
         * group of its own and takes the chunked per-op path below ------------------------------ */
        uint32_t gend = g0, V = 0;
        bool big = false;
        if (!any_stroke) {
            gend = total; /* fills only: one group, nothing to lay out */
        } else {
            for (; gend < total; ++gend) {
                const uint32_t nv = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.opnv[gend]);
                if (nv > 64u) {
                    if (gend == g0) {
                        big = true;
                        ++gend;
                    }
                    break;
                }
                if (V + nv > 64u) break;
                if (lane == 0) sh.grp_base[gend - g0] = V;
                V += nv;
            }
            if (lane == 0) sh.grp_base[gend - g0] = V;
            __syncthreads();
        }
        /* A normal group's records are produced by ONE pass (at its first stroke op) in which lane
         * -> (list entry, virtual segment) through grp_base; a big op runs one pass per 64 of its
         * own virtual segments.  Same code, one call site. */
        unsigned long long gbal = 0ull;
        uint32_t gincl = 0;
        bool records_ready = false;

        for (uint32_t li = g0; li < gend; ++li) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)(job.op_off + sh.oplist[li]));
            const osmt_op* __restrict__ op = &g_ops[o];
            const uint32_t kind = op->kind;
            if (kind == OSMT_OP_STROKE) {
                /* ---------------- draw_lines (line.rs:9-61) ---------------- */
                const osmt_opinfo* __restrict__ oi = &g_info[o];
                const osmt_stroke_aux* __restrict__ sa = &g_aux[oi->aux];
                const bool plain_main = sa->main.n_segs == 0;
                const double initial_opacity = op->opacity;
                const int32_t reach = oi->reach;
                unsigned long long* plane = sh.plane[buf];
                const bool has_caps = (op->cap == OSMT_CAP_ROUND || op->cap == OSMT_CAP_SQUARE);
                const uint32_t nv_op = oi->n_edges + (has_caps ? 2u : 0u);
                const uint32_t n_rounds = big ? (nv_op + 63u) / 64u : 1u;
                const uint32_t blk_off = (BLOCKS && big) ? g_op_blk[o] : 0xFFFFFFFFu;
                for (uint32_t round = 0; round < n_rounds; ++round) {
                    if (blk_off != 0xFFFFFFFFu && (round + 1u) * 64u <= oi->n_edges) {
                        /* a round made of edges only (the cap stubs live in the last round): skip it when
                         * the block's box, grown by the reach of a run, misses the sub-tile */
                        const osmt_blk_bbox bb = g_blk[blk_off + round];
                        if (bb.x1 + reach < rc.x0 || bb.x0 - reach > rc.x1 || bb.y1 + reach < rc.y0 || bb.y0 - reach > rc.y1)
                            continue;
                    }
                    if (big || !records_ready) {
                        /* ---- record pass: one lane per virtual segment (an edge, or one of the two cap
                         * stubs precomputed by k_opinfo, line.rs:33-57): cull, item ranges, record ---- */
                        if (big) __syncthreads(); /* previous round's records are consumed */
                        SegRec rec;
                        rec.count = 0;
                        rec.p1x = rec.p1y = rec.p2x = rec.p2y = 0;
                        rec.traveled = 0.0;
                        rec.denom = 1.0;
                        rec.caps_table = 0u;
                        uint32_t lo = o, v = round * 64u + lane; /* big: this op, its round-th 64 segments */
                        bool valid = v < nv_op;
                        if (!big) {
                            valid = lane < V;
                            uint32_t j = 0;
                            const uint32_t gn = gend - g0;
                            while (j + 1u < gn && sh.grp_base[j + 1u] <= lane) ++j; /* last entry with base <= lane */
                            lo = job.op_off + sh.oplist[g0 + j];
                            v = lane - sh.grp_base[j];
                        }
                        if (valid) {
                            const osmt_opinfo loi = g_info[lo];
                            const osmt_op* lop = &g_ops[lo];
                            const uint32_t ne_all = loi.n_edges;
                            bool live = false;
                            if (v < ne_all) {
                                /* (ring, edge) of running edge index v (point_pairs.rs:36-40) */
                                uint32_t r = 0, e = v;
                                osmt_ring ring = g_rings[lop->ring_off];
                                while ((ring.n_pts < 2u || e >= ring.n_pts - 1u) && r + 1u < lop->n_rings) {
                                    if (ring.n_pts >= 2u) e -= ring.n_pts - 1u;
                                    ring = g_rings[lop->ring_off + ++r];
                                }
                                const int2 p1 = g_pts[ring.first_pt + e];
                                const int2 p2 = g_pts[ring.first_pt + e + 1];
                                rec.p1x = p1.x; rec.p1y = p1.y; rec.p2x = p2.x; rec.p2y = p2.y;
                                rec.traveled = g_trav[ring.first_pt + e];
                                rec.denom = g_den[ring.first_pt + e];
                                live = true;
                            } else {
                                const osmt_cap_seg cs = g_aux[loi.aux].cap_seg[v - ne_all];
                                rec.p1x = cs.p1x; rec.p1y = cs.p1y; rec.p2x = cs.p2x; rec.p2y = cs.p2y;
                                rec.denom = cs.denom;
                                rec.caps_table = 1u;
                                live = cs.valid != 0;
                            }
                            if (live)
                                rec.count = seg_ranges(rec.p1x, rec.p1y, rec.p2x, rec.p2y, loi.reach, loi.reach_major, rc, &rec);
                        }
                        gbal = __ballot(rec.count > 0u);
                        if (rec.count > 0u) sh.seg[__popcll(gbal & ((1ull << lane) - 1ull))] = rec;
                        gincl = rec.count; /* inclusive prefix of the item counts over the lanes */
#pragma unroll
                        for (uint32_t d = 1; d < 64u; d <<= 1) {
                            const uint32_t y = __shfl_up(gincl, d);
                            if (lane >= d) gincl += y;
                        }
                        records_ready = true;
                        __syncthreads();
                    }
                    /* the op's records are lanes [va, vb) of the record pass */
                    uint32_t va = 0, vb = min(64u, nv_op - round * 64u);
                    if (!big) {
                        va = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.grp_base[li - g0]);
                        vb = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.grp_base[li - g0 + 1u]);
                    }
                    if (vb > va) {
                        const unsigned long long lanes_ab =
                            ((vb >= 64u) ? ~0ull : ((1ull << vb) - 1ull)) & ~((1ull << va) - 1ull);
                        const unsigned long long mask = gbal & lanes_ab;
                        const uint32_t item_lo = va ? (uint32_t)__builtin_amdgcn_readlane((int)gincl, (int)va - 1) : 0u;
                        const uint32_t item_hi = (uint32_t)__builtin_amdgcn_readlane((int)gincl, (int)vb - 1);
                        const uint32_t slot0 = (uint32_t)__popcll(gbal & ((1ull << va) - 1ull));
                        /* all (record, item) pairs of this op, lanes packed */
                        for (uint32_t it = item_lo + lane; it < item_hi; it += 64u) {
                            /* record of item `it`: walk the (wave-uniform) record lanes of this op and
                             * compare with each one's inclusive item prefix, read with v_readlane */
                            uint32_t j = slot0, base_items = item_lo;
                            for (unsigned long long rem = mask; rem != 0ull; rem &= rem - 1ull) {
                                const uint32_t pv = (uint32_t)__builtin_amdgcn_readlane((int)gincl, __builtin_ctzll(rem));
                                if (it >= pv) {
                                    ++j;
                                    base_items = pv;
                                }
                            }
                            const SegRec q = sh.seg[j];
                            walk_item(q, it - base_items, plain_main, sa, initial_opacity, reach, rc, plane);
                        }
                    }
                }
                __syncthreads();
                /* blend this generation's pending pixels (tile_pixels.rs:205-223) */
                const double cr = (double)op->color[0] / 255.0;
                const double cg = (double)op->color[1] / 255.0;
                const double cb = (double)op->color[2] / 255.0;
#pragma unroll
                for (int j = 0; j < PXT; ++j) {
                    const uint32_t idx = (ly0 + (uint32_t)j * ROWSTEP) * SUB + lx;
                    const unsigned long long bits = plane[idx];
                    if (bits != 0ull) {
                        plane[idx] = 0ull;
                        const double a = __longlong_as_double((long long)bits);
                        blend_rgb(acc[j], a * cr, a * cg, a * cb, a); /* from_color: o * (c/255) */
                    }
                }
                if (NBUF == 1) __syncthreads(); /* plane/mask reused by the next op */
                buf = (buf + 1u) % NBUF;
            } else {
                /* ---------------- fill_contour (fill.rs:16-47) ---------------- */
                /* A: every (edge, row) pair -> un-poisoned Edge{x_min,x_max} record of that row.  Ops
                 * with more than 64 edges are walked block by block (64 running edge indices), skipping
                 * the blocks whose rows cannot meet the sub-tile's rows (x does not matter: crossings
                 * left or right of the sub-tile still decide the parity). */
                /* only ops with more than 64 edges have blocks: do not even load the offset for a
                 * single short ring (every polygon of the named configs) */
                const bool maybe_long = BLOCKS && (op->n_rings > 1u || g_rings[op->ring_off].n_pts > 65u);
                const uint32_t fblk_off = maybe_long ? g_op_blk[o] : 0xFFFFFFFFu;
                uint32_t e_base = 0;
                for (uint32_t r = 0; r < op->n_rings; ++r) {
                    const osmt_ring ring = g_rings[op->ring_off + r];
                    if (ring.n_pts < 2) continue;
                    const uint32_t ne = ring.n_pts - 1;
                    /* pieces of this ring: the whole ring (short ops), or its intersections with the
                     * 64-edge blocks that can meet this sub-tile's rows (long ops) */
                    uint32_t b_first = 0, b_last = 0;
                    if (BLOCKS && fblk_off != 0xFFFFFFFFu) {
                        b_first = e_base >> 6;
                        b_last = (e_base + ne - 1u) >> 6;
                    }
                    for (uint32_t bk = b_first; bk <= b_last; ++bk) {
                        uint32_t e_lo = 0, n_items = ne * SUBH; /* ring-local first edge, (edge,row) items */
                        if (BLOCKS && fblk_off != 0xFFFFFFFFu) {
                            const osmt_blk_bbox bb = g_blk[fblk_off + bk];
                            /* rows carrying records of an edge: ytop < y <= ybot */
                            if (bb.y1 < rc.y0 || bb.y0 >= rc.y1) continue;
                            e_lo = max(bk << 6, e_base) - e_base;
                            n_items = (min((bk + 1u) << 6, e_base + ne) - e_base - e_lo) * SUBH;
                        }
                        for (uint32_t it = tid; it < n_items; it += NTHREADS) {
                            const uint32_t e = e_lo + it / SUBH, row = it % SUBH;
                            const int2 p1 = g_pts[ring.first_pt + e];
                            const int2 p2 = g_pts[ring.first_pt + e + 1];
                            int32_t xmn, xmx;
                            if (osmt_fill_row_extent(p1.x, p1.y, p2.x, p2.y, rc.y0 + (int32_t)row, &xmn, &xmx)) {
                                const uint32_t slot = atomicAdd(&sh.rowcnt[row], 1u);
                                if (slot < ROWCAP) {
                                    sh.rec[row][slot].x_min = xmn;
                                    sh.rec[row][slot].x_max = xmx;
                                    sh.rec[row][slot].edge = e_base + e;
                                }
                            }
                        }
                    }
                    e_base += ne;
                }
                __syncthreads();
                /* B: per row: order by (x_min, edge index) == stable sort_by_key(x_min) of records
                 * inserted in edge order (fill.rs:24-25), pair (0,1),(2,3).., OR the spans */
                if (tid < SUBH) {
                    const uint32_t row = tid;
                    const uint32_t n = sh.rowcnt[row];
                    uint32_t m = 0u;
                    if (n <= ROWCAP) {
                        RowRec* rr = sh.rec[row];
                        for (uint32_t i = 1; i < n; ++i) {
                            const RowRec key = rr[i];
                            int32_t j = (int32_t)i - 1;
                            while (j >= 0 && (rr[j].x_min > key.x_min ||
                                              (rr[j].x_min == key.x_min && rr[j].edge > key.edge))) {
                                rr[j + 1] = rr[j];
                                --j;
                            }
                            rr[j + 1] = key;
                        }
                        for (uint32_t k = 0; k + 1 < n; k += 2) {
                            const int32_t from = max(rr[k].x_min, rc.x0);
                            const int32_t to = min(rr[k + 1].x_max, rc.x1);
                            if (from <= to) {
                                const uint32_t len = (uint32_t)(to - from + 1);
                                const uint32_t bits = (len >= 32u) ? 0xFFFFFFFFu : ((1u << len) - 1u);
                                m |= bits << (uint32_t)(from - rc.x0);
                            }
                        }
                    } else {
                        /* slow path (more than ROWCAP crossings on a row): kept out of line — it is cold,
                         * and inlined it would only add code and register pressure to the hot path */
                        m = fill_row_streaming(g_rings, g_pts, op->ring_off, op->n_rings, rc.y0 + (int32_t)row, rc.x0, rc.x1);
                    }
                    sh.mask[buf][row] = m;
                    sh.rowcnt[row] = 0u;
                }
                __syncthreads();
                /* C: set_pixel + blend of the covered pixels */
                if (kind == OSMT_OP_FILL_COLOR) {
                    const double o_ = op->opacity;
                    const double sr = o_ * ((double)op->color[0] / 255.0);
                    const double sg = o_ * ((double)op->color[1] / 255.0);
                    const double sb = o_ * ((double)op->color[2] / 255.0);
#pragma unroll
                    for (int j = 0; j < PXT; ++j) {
                        const uint32_t row = ly0 + (uint32_t)j * ROWSTEP;
                        if ((sh.mask[buf][row] >> lx) & 1u) blend_rgb(acc[j], sr, sg, sb, o_);
                    }
                } else { /* Filler::Image: icon.get(x % w, y % h), opacity ignored (fill.rs:36-40) */
                    const uint32_t img = op->image_id;
                    if (img < g_n_images) {
                        const osmt_image_desc im = g_images[img];
                        const double4* __restrict__ ipx = g_image_pool + im.offset;
#pragma unroll
                        for (int j = 0; j < PXT; ++j) {
                            const uint32_t row = ly0 + (uint32_t)j * ROWSTEP;
                            if ((sh.mask[buf][row] >> lx) & 1u) {
                                const uint32_t ix = (uint32_t)(rc.x0 + (int32_t)lx) % im.width;
                                const uint32_t iy = (uint32_t)(rc.y0 + (int32_t)row) % im.height;
                                const double4 c = ipx[(size_t)iy * im.width + ix];
                                blend_rgb(acc[j], c.x, c.y, c.z, c.w);
                            }
                        }
                    }
                }
                if (NBUF == 1) __syncthreads(); /* plane/mask reused by the next op */
                buf = (buf + 1u) % NBUF;
            }
        }
        __syncthreads(); /* the group's records / bases are rewritten by the next group */
        g0 = gend;
        }
        __syncthreads(); /* oplist is rewritten by the next chunk */
    }

    /* ---- label pass, blend_unfinished_pixels(true) (tile_pixels.rs:154-158,205-223) ----------
     * k_label_resolve has decided which labels succeeded; succeeded labels never share a pixel
     * (set_label_pixel refuses the second one), so every pixel is blended at most once and the
     * order of the loop does not matter.  Inside one label the text's pixels (total > 0) were
     * written after the icon's and replace them (labeler.rs:29-31). */
    if (LABELS) {
        const osmt_tile_label* OSMT_R tl = g_tl + g_job_label_off[tile];
        const uint32_t n_tl = g_tl_cnt[tile];
        for (uint32_t k = 0; k < n_tl; ++k) {
            const osmt_tile_label e = tl[k];
            if (e.x0 > rc.x1 || e.x1 < rc.x0 || e.y0 > rc.y1 || e.y1 < rc.y0) continue;
            const osmt_labelinfo* OSMT_R li = g_lab + e.label;
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
            const int32_t x = rc.x0 + (int32_t)lx;
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const int32_t y = rc.y0 + (int32_t)(ly0 + (uint32_t)j * ROWSTEP);
                double t = 0.0;
                if (text_hit && y >= ry0 && y <= ry1 && x >= cx0 && x <= cx1)
                    t = plane[(size_t)(y - ry0) * cols + (uint32_t)(x - cx0)];
                if (t > 0.0) { /* RgbaColor::from_color(&self.color, total) (rasterizer.rs:140) */
                    blend_rgb(acc[j], t * cr, t * cg, t * cb, t);
                } else if (icon_hit && x >= ix0 && x < ix0 + iw && y >= iy0 && y < iy0 + ih) {
                    const double4 c = ipx[(size_t)(y - iy0) * (uint32_t)iw + (uint32_t)(x - ix0)];
                    blend_rgb(acc[j], c.x, c.y, c.z, c.w);
                }
            }
        }
    }

    /* ---- to_rgb_triples (tile_pixels.rs:164-181) / raw canvas ---------------- */
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const uint32_t row = ly0 + (uint32_t)j * ROWSTEP;
        const size_t px = (size_t)(rc.y0 + (int32_t)row) * W + (size_t)(rc.x0 + (int32_t)lx);
        if (OUT_F64) {
            double4* out = reinterpret_cast<double4*>(g_out) + (size_t)tile * W * W + px;
            *out = make_double4(acc[j][0], acc[j][1], acc[j][2], 1.0);
        } else {
            /* postdivide (tile_pixels.rs:171-175) with p.a == 1.0: val / 1.0 == val */
            const uint32_t v = f64_as_u8(255.0 * acc[j][0]) | (f64_as_u8(255.0 * acc[j][1]) << 8) |
                               (f64_as_u8(255.0 * acc[j][2]) << 16) | 0xFF000000u;
            uint32_t* out = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(g_out) +
                                                        (size_t)tile * g_out_tile_stride) + px;
            *out = v;
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

}  // namespace

/* ---- launchers (C++ internal interface, see osmt_internal.h) ---------------- */
/* ------------------------------------------------------------------------- */
/* Label pass: font/rasterizer.rs + tile_pixels.rs:131-162 + labeler.rs:91-106.
 *
 * k_label_cover    one wave per label.  Lane = one stripe y of the label's window; the wave digests the
 *                  draw_line calls 64 at a time (one call per lane: the y-independent part of draw_line,
 *                  two f64 divisions) into LDS, then every lane walks the calls that cross the band IN
 *                  CALL ORDER and adds those that cross its stripe into its own row of the LDS-resident
 *                  A / S accumulators — the per-key f64 sums therefore happen in exactly the reference's
 *                  order (BTreeMap entry += ..., :77,:80) with no atomics.  Then the lane runs
 *                  save_to_figure's scan over [x_min, x_max] of its stripe (:121-143) and the band is
 *                  copied out coalesced: total = min(a + s_acc, 1.0) per cell, 0 where the stripe has no key.
 * k_label_resolve  one workgroup per tile, labels strictly in draw order: a label succeeds iff none of
 *                  the pixels it would set (icon rectangle, then cells with total > 0) inside labels_bb
 *                  belongs to an earlier SUCCEEDED label (set_label_pixel, tile_pixels.rs:131-148;
 *                  pixels of failed labels are overwritten freely); succeeded labels mark their pixels
 *                  in a (3W)^2-bit ownership map.  The early `return false` of draw_icon /
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

/* A draw_line call parks its sums in its lane's registers, one per CHANNEL = (column parity, A/S kind, stripe
 * parity): the cells one short call touches always fall into different channels, and a given cell always falls
 * into the same one, so "consecutive calls adding to the same cell" is simply "consecutive lanes with the same
 * key in that channel".  Calls whose cells collide in a channel (three cells wide, ...) are replayed stripe by
 * stripe by the row owners instead. */
#define LC_CH 8
#define LC_NOCOL 0xFFFFFFFFu

__device__ __forceinline__ double readlane_f64(double v, uint32_t j) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, (int)j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), (int)j);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

__global__ __launch_bounds__(64) void k_label_cover(const osmt_labelinfo* __restrict__ g_lab, uint32_t n_labels,
                                                    const double4* __restrict__ g_seg, double* __restrict__ g_a,
                                                    uint32_t* g_err) {
    __shared__ double sh_a[LC_CELLS];
    __shared__ double sh_s[LC_CELLS];
    const uint32_t l = blockIdx.x;
    if (l >= n_labels) return;
    const osmt_labelinfo* __restrict__ li = g_lab + l;
    if (!li->has_text || li->ry0 > li->ry1 || li->cols == 0 || li->cols > LC_CELLS) return;
    const uint32_t lane = threadIdx.x;
    const int32_t ry0 = li->ry0, cx0 = li->cx0;
    const uint32_t R = (uint32_t)(li->ry1 - ry0 + 1), cols = li->cols;
    const uint32_t n_segs = li->n_segs;
    const double4* __restrict__ segs = g_seg + li->seg_off;
    double* __restrict__ A = g_a + li->plane_off;
    const uint32_t band_rows = min(64u, LC_CELLS / cols);
    bool oob = false;
    for (uint32_t rbase = 0; rbase < R; rbase += band_rows) {
        const uint32_t nrow = min(band_rows, R - rbase);
        const uint32_t cnt = nrow * cols;
        for (uint32_t i = lane; i < cnt; i += 64u) {
            sh_a[i] = 0.0;
            sh_s[i] = 0.0;
        }
        __syncthreads();
        const bool active = lane < nrow;
        const int32_t y = ry0 + (int32_t)(rbase + lane);
        double* a_row = sh_a + (active ? lane * cols : 0u);
        double* s_row = sh_s + (active ? lane * cols : 0u);
        /* the stripe owner keeps the cell it is adding to in a register (consecutive calls of a curve land in
         * the same cell): LDS is touched only when the cell changes */
        uint32_t a_col = LC_NOCOL, s_col = LC_NOCOL;
        double a_val = 0.0, s_val = 0.0;
        uint32_t c_min = 0xFFFFFFFFu, c_max = 0u; /* columns of the stripe's keys (x - cx0) */
        const int32_t band0 = ry0 + (int32_t)rbase, band1 = band0 + (int32_t)nrow - 1;
        for (uint32_t base = 0; base < n_segs; base += 64u) {
            /* ---- phase 1, lane = draw_line call: all the f64 work of the call's stripes inside the band ---- */
            const uint32_t i = base + lane;
            bool overlaps = false, slow = false;
            osmt_label_seg sg;
            uint32_t chmask = 0u;
            uint32_t ekey[LC_CH]; /* kind << 31 | local stripe << 20 | column */
            double eval[LC_CH];
#pragma unroll
            for (int k = 0; k < LC_CH; ++k) {
                ekey[k] = 0xFFFFFFFFu;
                eval[k] = 0.0;
            }
            if (i < n_segs) {
                sg = label_seg_prep(segs[i]);
                overlaps = sg.yl >= band0 && sg.yf <= band1; /* also drops delta == 0 (yf > yl) */
                if (overlaps) {
                    const int32_t ya = max(sg.yf, band0), yb = min(sg.yl, band1);
                    auto emit = [&](uint32_t kind, uint32_t row, uint32_t col, double val) {
                        const uint32_t ch = (col & 1u) | (kind << 1) | ((row & 1u) << 2);
                        if ((chmask >> ch) & 1u) slow = true;
                        chmask |= 1u << ch;
                        const uint32_t key = (kind << 31) | (row << 20) | col;
#pragma unroll
                        for (int k = 0; k < LC_CH; ++k)
                            if ((uint32_t)k == ch) {
                                ekey[k] = key;
                                eval[k] = val;
                            }
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
            }
            /* ---- phase 2, lane = stripe: the parked sums are applied strictly in call order ---- */
            unsigned long long rest = __ballot(overlaps);
            const unsigned long long slowm = __ballot(overlaps && slow);
            while (rest) {
                /* calls before the next replayed one form a segment whose channels can be handled one by one: sums to
                 * different cells are independent, sums to one cell (one channel, equal keys) stay in call order */
                const unsigned long long sl_rest = slowm & rest;
                const uint32_t sl = sl_rest ? (uint32_t)__builtin_ctzll(sl_rest) : 64u;
                const unsigned long long seg = sl < 64u ? (rest & ((1ull << sl) - 1ull)) : rest;
                const bool in_seg = (seg >> lane) & 1ull;
#pragma unroll
                for (int ch = 0; ch < LC_CH; ++ch) {
                    const bool valid = in_seg && ((chmask >> ch) & 1u);
                    const unsigned long long vm = __ballot(valid);
                    if (!vm) continue;
                    const uint32_t key = ekey[ch];
                    const uint32_t pkey = (uint32_t)__shfl_up((int)key, 1);
                    const bool pvalid = lane != 0u && ((vm >> (lane - 1u)) & 1ull);
                    const bool head = valid && !(pvalid && pkey == key);
                    unsigned long long hm = __ballot(head);
                    const unsigned long long cont = vm & ~hm; /* lanes continuing their predecessor's run */
                    while (hm) {
                        const uint32_t h = (uint32_t)__builtin_ctzll(hm);
                        hm &= hm - 1ull;
                        const uint32_t run = 1u + (uint32_t)__builtin_ctzll(~((cont >> 1) >> h));
                        const uint32_t K = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)h);
                        const uint32_t col = K & 0xFFFFFu;
                        if (((K >> 20) & 0x7FFu) == lane) { /* the stripe's owner; v_readlane below ignores EXEC */
                            uint32_t src = h, left = run;
                            if (K >> 31) {
                                if (col != s_col) {
                                    if (s_col != LC_NOCOL) s_row[s_col] = s_val;
                                    s_val = s_row[col];
                                    s_col = col;
                                }
                                do {
                                    s_val += readlane_f64(eval[ch], src);
                                    ++src;
                                } while (--left);
                            } else {
                                if (col != a_col) {
                                    if (a_col != LC_NOCOL) a_row[a_col] = a_val;
                                    a_val = a_row[col];
                                    a_col = col;
                                }
                                do {
                                    a_val += readlane_f64(eval[ch], src);
                                    ++src;
                                } while (--left);
                            }
                            c_min = min(c_min, col);
                            c_max = max(c_max, col);
                        }
                    }
                }
                if (sl >= 64u) break;
                { /* the replayed call works on LDS directly: write the cached cells back first */
                    const uint32_t j = sl;
                    osmt_label_seg q;
                    q.x0 = readlane_f64(sg.x0, j);
                    q.y0 = readlane_f64(sg.y0, j);
                    q.slope = readlane_f64(sg.slope, j);
                    q.slope_recip = readlane_f64(sg.slope_recip, j);
                    q.y_min = readlane_f64(sg.y_min, j);
                    q.y_max = readlane_f64(sg.y_max, j);
                    q.sign = readlane_f64(sg.sign, j);
                    q.yf = __builtin_amdgcn_readlane(sg.yf, (int)j);
                    q.yl = __builtin_amdgcn_readlane(sg.yl, (int)j);
                    if (a_col != LC_NOCOL) a_row[a_col] = a_val;
                    if (s_col != LC_NOCOL) s_row[s_col] = s_val;
                    a_col = s_col = LC_NOCOL;
                    if (active && y >= q.yf && y <= q.yl) {
                        int32_t x_min = INT32_MAX, x_max = INT32_MIN;
                        oob |= !label_stripe(q, y, cx0, cols, a_row, s_row, x_min, x_max);
                        if (x_min <= x_max) {
                            c_min = min(c_min, (uint32_t)(x_min - cx0));
                            c_max = max(c_max, (uint32_t)(x_max - cx0));
                        }
                    }
                }
                rest &= ~((2ull << sl) - 1ull);
            }
        }
        if (a_col != LC_NOCOL) a_row[a_col] = a_val;
        if (s_col != LC_NOCOL) s_row[s_col] = s_val;
        /* save_to_figure (:115-147) for this stripe: keys span [c_min, c_max]; the rest of the row stays 0 */
        if (active && c_min <= c_max) {
            double s_acc = 0.0;
            for (uint32_t c = c_min; c <= c_max; ++c) {
                s_acc += s_row[c];
                a_row[c] = fmin(a_row[c] + s_acc, 1.0);
            }
        }
        __syncthreads();
        double* __restrict__ dst = A + (size_t)rbase * cols;
        for (uint32_t i = lane; i < cnt; i += 64u) dst[i] = sh_a[i];
        __syncthreads();
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

#define OSMT_LABEL_RESOLVE_THREADS 256
/* LDS_BM: the (3W)^2-bit ownership map lives in LDS (scale 1: 72 KB); otherwise in global memory. */
template <bool LDS_BM>
__global__ __launch_bounds__(OSMT_LABEL_RESOLVE_THREADS) void k_label_resolve(
    const osmt_labelinfo* __restrict__ g_lab, const uint32_t* __restrict__ g_job_label_off, uint32_t n_jobs, uint32_t scale,
    const double* __restrict__ g_a, uint32_t* g_bitmap, uint8_t* g_ok, osmt_tile_label* __restrict__ g_tl,
    uint32_t* __restrict__ g_tl_cnt) {
    extern __shared__ uint32_t sh_bm[];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_jobs) return;
    const uint32_t tid = threadIdx.x;
    const int32_t W = (int32_t)(OSMT_TILE_SIZE * scale);
    const uint32_t EW = 3u * (uint32_t)W; /* labels_bb is the 3x3-tile square [-W, 2W) (tile_pixels.rs:67-72) */
    const size_t words = ((size_t)EW * EW + 31u) / 32u;
    uint32_t* bm = LDS_BM ? sh_bm : g_bitmap + (size_t)tile * words;
    for (size_t i = tid; i < words; i += OSMT_LABEL_RESOLVE_THREADS) bm[i] = 0u;
    if (!LDS_BM) __threadfence();
    __syncthreads();
    auto test = [&](uint32_t bit) -> bool {
        if (LDS_BM) return (bm[bit >> 5] >> (bit & 31u)) & 1u;
        return (__hip_atomic_load(bm + (bit >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (bit & 31u)) & 1u;
    };
    const uint32_t l0 = g_job_label_off[tile], l1 = g_job_label_off[tile + 1];
    uint32_t n_out = 0; /* thread 0: succeeded labels that reach into the tile itself */
    for (uint32_t l = l0; l < l1; ++l) {
        const osmt_labelinfo* __restrict__ li = g_lab + l;
        const int32_t ix0 = li->icon_x, iy0 = li->icon_y;
        const uint32_t iw = li->icon_w, ih = li->icon_h;
        const bool has_cells = li->has_text && li->ry0 <= li->ry1 && li->cols > 0;
        const int32_t ry0 = li->ry0, cx0 = li->cx0;
        const uint32_t cols = li->cols;
        const uint32_t n_cells = has_cells ? (uint32_t)(li->ry1 - ry0 + 1) * cols : 0u;
        const double* __restrict__ A = g_a + li->plane_off;
        bool failed = false;
        for (int pass = 0; pass < 2; ++pass) { /* 0: collide with earlier succeeded labels, 1: take ownership */
            bool hit = false;
            for (uint32_t i = tid; i < iw * ih; i += OSMT_LABEL_RESOLVE_THREADS) {
                const int32_t x = ix0 + (int32_t)(i % iw), y = iy0 + (int32_t)(i / iw);
                if (x < -W || x >= 2 * W || y < -W || y >= 2 * W) continue; /* set_label_pixel: outside labels_bb -> true */
                const uint32_t bit = (uint32_t)(y + W) * EW + (uint32_t)(x + W);
                if (pass == 0)
                    hit |= test(bit);
                else
                    atomicOr(bm + (bit >> 5), 1u << (bit & 31u));
            }
            for (uint32_t i = tid; i < n_cells; i += OSMT_LABEL_RESOLVE_THREADS) {
                if (!(A[i] > 0.0)) continue;
                const int32_t x = cx0 + (int32_t)(i % cols), y = ry0 + (int32_t)(i / cols);
                if (x < -W || x >= 2 * W) continue; /* rows are clipped already */
                const uint32_t bit = (uint32_t)(y + W) * EW + (uint32_t)(x + W);
                if (pass == 0)
                    hit |= test(bit);
                else
                    atomicOr(bm + (bit >> 5), 1u << (bit & 31u));
            }
            if (pass == 0) {
                failed = __syncthreads_or(hit ? 1 : 0) != 0;
                if (tid == 0) g_ok[l] = failed ? 0 : 1; /* bump_label_generation(succeeded) */
                if (failed) break;
            } else {
                if (!LDS_BM) __threadfence();
                __syncthreads();
            }
        }
        if (!failed && tid == 0) {
            /* what k_raster has to look at: the label's pixels clipped to the tile [0, W)^2 */
            int32_t bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = INT32_MIN, by1 = INT32_MIN;
            if (has_cells) {
                bx0 = cx0, bx1 = cx0 + (int32_t)cols - 1, by0 = ry0, by1 = li->ry1;
            }
            if (iw) {
                bx0 = min(bx0, ix0), bx1 = max(bx1, ix0 + (int32_t)iw - 1);
                by0 = min(by0, iy0), by1 = max(by1, iy0 + (int32_t)ih - 1);
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
    }
    if (tid == 0) g_tl_cnt[tile] = n_out;
}

/* ------------------------------------------------------------------------- */
/* PNG encoding on the GPU (rgb_triples_to_png, png_writer.rs:4-21): one wave per tile
 * turns an RGBA8 framebuffer into a complete RGB8 PNG file — Paeth-filtered rows, ONE fixed-Huffman deflate
 * block whose only matches are distance-1 runs, Adler-32, chunk CRCs — so that a serving pipeline moves
 * ~50 KB per tile over PCIe instead of 256 KB and the host does no zlib work.  The reference's tests compare
 * decoded pixels only (tests/test_rendering.rs:15-23), so the encoder is free to differ from the png crate.
 *
 * Per row (3W+1 filtered bytes in LDS, one wave): lanes own contiguous byte spans; a byte starts a run when it differs
 * from its predecessor; a run (v, L) becomes literal(v), matches(len <= 258, dist 1) for the other L-1 bytes,
 * and at most two trailing literals.  Bit counts are prefix-summed across lanes and tokens are OR-ed into an LDS
 * bit buffer; rows are sized first so that every row knows its bit position in the file (see k_png_encode). */
#define PNG_HDR_BYTES 43u /* 8 signature + 25 IHDR + 4 IDAT length + 4 "IDAT" + 2 zlib header */

__device__ __forceinline__ void png_lit(uint32_t v, uint32_t& bits, uint32_t& n) {
    if (v < 144u) {
        bits = __brev(0x30u + v) >> 24;
        n = 8u;
    } else {
        bits = __brev(0x190u + (v - 144u)) >> 23;
        n = 9u;
    }
}
/* match of length L (3..258) at distance 1: length code + extra bits + the 5-bit distance code 0 */
__device__ __forceinline__ void png_run(uint32_t L, uint32_t& bits, uint32_t& n) {
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
    uint32_t hb, hn;
    if (idx <= 22u) { /* codes 257..279: 7 bits */
        hb = __brev(idx + 1u) >> 25;
        hn = 7u;
    } else { /* 280..285: 8 bits */
        hb = __brev(0xC0u + idx - 23u) >> 24;
        hn = 8u;
    }
    bits = hb | (ev << hn);
    n = hn + eb + 5u;
}
/* bits of the tokens of run (v, L) */
__device__ __forceinline__ uint32_t png_run_bits(uint32_t v, uint32_t L) {
    const uint32_t ln = v < 144u ? 8u : 9u;
    uint32_t total = ln, R = L - 1u;
    while (R >= 258u) { /* code 285: 8 + 0 + 5 bits; at most 11 rounds per 1024-px row (no integer division) */
        total += 13u;
        R -= 258u;
    }
    if (R >= 3u) {
        uint32_t b, n;
        png_run(R, b, n);
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
__device__ __forceinline__ uint32_t png_row_tokens(const uint8_t* f, uint32_t NB, uint32_t lane, uint32_t* bits, uint32_t& adler1,
                                                   uint32_t& adler2) {
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
    uint32_t my_bits = lane == 0 ? 8u : 0u; /* the filter-type byte: literal(4) */
    for (uint32_t k = s0; k < s1;) {
        const uint32_t v = f[k];
        const bool is_start = k == 1u || v != f[k - 1u];
        uint32_t e = k + 1u;
        while (e < s1 && f[e] == v) ++e;
        if (is_start) my_bits += png_run_bits(v, ((e == s1) ? nxt_pos : e) - k);
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
        png_lit(4u, b, n);
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
            png_lit(v, lb, ln);
            png_put(bits, pos, lb, ln);
            uint32_t R = end - k - 1u;
            while (R >= 3u) {
                const uint32_t m = min(R, 258u);
                uint32_t b, n;
                png_run(m, b, n);
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
    __shared__ uint32_t sh_bits[PNG_WAVES][(9u * (3u * PNG_MAX_W + 1u)) / 32u + 4u];
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
    /* ---- pass 1: size every row ---- */
    for (uint32_t y = wave; y < H; y += PNG_WAVES) {
        png_filter_row(src, W, y, lane, sh_f[wave]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t a1, a2;
        const uint32_t rb = png_row_tokens<false>(sh_f[wave], NB, lane, nullptr, a1, a2);
        if (lane == 0) {
            sh_rowpos[y] = rb;
            sh_a1[y] = a1;
            sh_a2[y] = a2;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    /* ---- row positions (bit 0 of the deflate stream = byte 43, after the 3 block-header bits) + Adler-32 ---- */
    if (tid == 0) {
        uint32_t pos = PNG_HDR_BYTES * 8u + 3u;
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
    /* words shared by two rows (and the word the stream ends in) start from zero; word 10 carries 'T', the zlib
     * header and the block header */
    for (uint32_t y = tid; y <= H; y += 64u * PNG_WAVES) {
        const uint32_t w = sh_rowpos[y] >> 5;
        if (w != 10u) out_w[w] = 0u;
        if (y == H) out_w[w + 1u] = 0u; /* the 7 EOB bits may spill into the next word */
    }
    if (tid == 0) out_w[10] = 0x00017854u | (3u << 24);
    __threadfence_block();
    __syncthreads();
    /* ---- pass 2: emit ---- */
    const uint32_t nwords_row = (9u * NB) / 32u + 2u;
    for (uint32_t y = wave; y < H; y += PNG_WAVES) {
        png_filter_row(src, W, y, lane, sh_f[wave]);
        for (uint32_t i = lane; i < nwords_row; i += 64u) sh_bits[wave][i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t a1, a2;
        const uint32_t row_bits = png_row_tokens<true>(sh_f[wave], NB, lane, sh_bits[wave], a1, a2);
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
    /* end of block (7 zero bits: already there), pad to a byte, Adler-32, IDAT length */
    const uint32_t gend = sh_rowpos[H] + 7u;
    const uint32_t endb = (gend + 7u) >> 3; /* first byte after the deflate stream */
    if (tid == 0) {
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
 * Wave w owns the band of rows [w*H/4, (w+1)*H/4) and walks it top to bottom; a lane owns PX consecutive pixels,
 * whose raw values, the row above (carried in registers from the previous iteration) and the 3*PX filtered bytes
 * all live in registers — run starts, run ends inside the lane and token sizes are straight-line code, the next
 * row's pixels are fetched while the current one is tokenised.  Band 0 appends its rows directly behind the
 * file header; bands 1..3 append into staging areas further up the tile's slot (bit 0 of a word), and once the
 * band lengths are known they are moved down, bit-shifted, behind their predecessors (dst <= src, chunked
 * read-then-write).  Adler-32 per band, combined like zlib's adler32_combine. */
template <int PX>
__global__ __launch_bounds__(256) void k_png_encode_fast(const uint8_t* __restrict__ g_rgba, size_t tile_stride, uint32_t n_tiles,
                                                         uint32_t H, uint32_t ihdr_crc, uint8_t* g_out, size_t out_stride,
                                                         uint32_t band_cap_words, uint32_t* __restrict__ g_len) {
    constexpr uint32_t W = 64u * PX, NB = 3u * W + 1u, NBY = 3u * PX; /* bytes per lane */
    constexpr uint32_t ROWW = (9u * NB) / 32u + 3u;
    __shared__ uint32_t sh_bits[4][ROWW];
    __shared__ uint32_t sh_crc_tab[256];
    __shared__ uint32_t sh_col[32];
    __shared__ uint32_t sh_raw[256];
    __shared__ uint32_t sh_carry[4];
    __shared__ uint32_t sh_band_bits[4], sh_band_a[4], sh_band_b[4];
    __shared__ uint32_t sh_move[256 + 1];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint8_t* __restrict__ src = g_rgba + (size_t)tile * tile_stride;
    uint8_t* out = g_out + (size_t)tile * out_stride;
    uint32_t* out_w = reinterpret_cast<uint32_t*>(out);
    for (uint32_t i = tid; i < 256u; i += 256u) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        sh_crc_tab[i] = c;
    }
    const uint32_t rows = H / 4u, y_begin = wave * rows, y_end = y_begin + rows;
    /* where this band's bits go while it is being produced */
    const uint32_t stage_w = wave == 0u ? 0u : 11u + wave * band_cap_words; /* band 0: the file itself */
    uint32_t gbit = wave == 0u ? PNG_HDR_BYTES * 8u + 3u : 0u;               /* bit cursor relative to out_w[stage_w] */
    uint32_t carry = wave == 0u ? (0x00017854u | (3u << 24)) : 0u;           /* band 0 continues word 10 */
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
        const uint32_t first_start = startmask ? base + (uint32_t)__builtin_ctz(startmask) : 0xFFFFFFFFu;
        const unsigned long long has = __ballot(startmask != 0u);
        const unsigned long long later = lane < 63u ? (has >> (lane + 1u)) : 0ull;
        const int nxt_lane = later ? (int)lane + 1 + __builtin_ctzll(later) : (int)lane;
        const uint32_t nxt_pos_raw = (uint32_t)__shfl((int)first_start, nxt_lane);
        const uint32_t nxt_pos = later ? nxt_pos_raw : NB;
        /* ---- size, prefix, emit ---- */
        uint32_t my_bits = lane == 0u ? 8u : 0u, a1 = lane == 0u ? 4u : 0u, a2 = lane == 0u ? NB * 4u : 0u;
#pragma unroll
        for (int i = 0; i < (int)NBY; ++i) {
            a1 += fb[i];
            a2 += (NB - (base + (uint32_t)i)) * fb[i];
            if ((startmask >> i) & 1u) {
                const uint32_t rest = startmask >> (i + 1); /* i + 1 < 32 always: NBY <= 24 */
                const uint32_t end = rest ? base + (uint32_t)i + 1u + (uint32_t)__builtin_ctz(rest) : nxt_pos;
                my_bits += png_run_bits(fb[i], end - (base + (uint32_t)i));
            }
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
            png_lit(4u, b, n);
            png_put(bits, pos, b, n);
        }
#pragma unroll
        for (int i = 0; i < (int)NBY; ++i) {
            if ((startmask >> i) & 1u) {
                const uint32_t rest = startmask >> (i + 1);
                const uint32_t end = rest ? base + (uint32_t)i + 1u + (uint32_t)__builtin_ctz(rest) : nxt_pos;
                uint32_t lb, ln;
                png_lit(fb[i], lb, ln);
                png_put(bits, pos, lb, ln);
                uint32_t R = end - (base + (uint32_t)i) - 1u;
                while (R >= 3u) {
                    const uint32_t m = min(R, 258u);
                    uint32_t b, n;
                    png_run(m, b, n);
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
        sh_band_bits[wave] = wave == 0u ? gbit - (PNG_HDR_BYTES * 8u + 3u) : gbit;
        sh_band_a[wave] = adler_a;
        sh_band_b[wave] = adler_b;
    }
    __threadfence_block();
    __syncthreads();
    /* ---- move bands 1..3 down behind their predecessors ---- */
    uint32_t endpos = PNG_HDR_BYTES * 8u + 3u + sh_band_bits[0];
    for (uint32_t b = 1; b < 4u; ++b) {
        const uint32_t L = sh_band_bits[b];
        const uint32_t sw = 11u + b * band_cap_words; /* staging: bit 0 of out_w[sw] */
        const uint32_t sh = endpos & 31u, wb = endpos >> 5;
        const uint32_t n_src = (L + 31u) >> 5;
        const uint32_t n_dst = ((endpos + L + 31u) >> 5) - wb; /* destination words holding bits of this band */
        for (uint32_t c = 0; c < n_dst; c += 256u) {
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
    /* end of block: 7 zero bits (the word after the last one may receive some of them) */
    if (tid == 0) {
        if (((endpos + 7u) >> 5) != (endpos >> 5) || !(endpos & 31u)) out_w[(endpos + 7u) >> 5] = 0u;
        /* Adler-32 of the concatenation (zlib's adler32_combine): A = A1 + A2 - 1, B = B1 + B2 + len2 * (A1 - 1) */
        unsigned long long A = sh_band_a[0], B = sh_band_b[0];
        const unsigned long long len2 = (unsigned long long)rows * NB;
        for (uint32_t b = 1; b < 4u; ++b) {
            const unsigned long long A2 = sh_band_a[b], B2 = sh_band_b[b];
            B = (B + B2 + (len2 % 65521ull) * ((A + 65520ull) % 65521ull)) % 65521ull;
            A = (A + A2 + 65520ull) % 65521ull;
        }
        sh_move[0] = (uint32_t)((B << 16) | A);
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t gend = endpos + 7u;
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
    const uint32_t blk = (c1 - c0 + 255u) / 256u;
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
        for (uint32_t i = 0; i < 256u; ++i) {
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

hipError_t osmt_launch_project(const osmt_tile_job* jobs, const uint32_t* pt_job, const double* latlon, const uint32_t* refs,
                               uint32_t n_pts, double scale, int32_t* pts, hipStream_t st) {
    if (n_pts == 0) return hipSuccess;
    hipLaunchKernelGGL(k_project, dim3((n_pts + 255u) / 256u), dim3(256), 0, st, jobs, pt_job,
                       reinterpret_cast<const double2*>(latlon), refs, n_pts, scale, reinterpret_cast<int2*>(pts));
    return hipGetLastError();
}

hipError_t osmt_launch_project_single(const double* latlon, uint32_t n, uint32_t zoom, uint32_t tx, uint32_t ty,
                                      double scale, int32_t* pts, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_project_single, dim3((n + 255u) / 256u), dim3(256), 0, st,
                       reinterpret_cast<const double2*>(latlon), n, zoom, tx, ty, scale, reinterpret_cast<int2*>(pts));
    return hipGetLastError();
}

hipError_t osmt_launch_opinfo(const osmt_op* ops, uint32_t n_ops, const osmt_ring* rings, const int32_t* pts,
                              const double* dashes, const uint32_t* op_aux, osmt_opinfo* info, double* trav,
                              double* den, osmt_stroke_aux* aux, uint8_t* opnv, const uint32_t* op_blk, osmt_blk_bbox* blk,
                              uint32_t* submask, uint32_t sub_rows, hipStream_t st) {
    if (n_ops == 0) return hipSuccess;
    hipLaunchKernelGGL(k_opinfo, dim3((n_ops + 63u) / 64u), dim3(64), 0, st, ops, n_ops, rings,
                       reinterpret_cast<const int2*>(pts), dashes, op_aux, info, trav, den, aux, opnv, op_blk, blk, submask, sub_rows);
    return hipGetLastError();
}

hipError_t osmt_launch_raster(const osmt_raster_args& a, bool out_f64, hipStream_t st) {
    if (a.n_jobs == 0) return hipSuccess;
    const uint32_t W = OSMT_TILE_SIZE * a.scale;
    const uint32_t nsub = (W / SUB) * (W / SUBH);
    const uint32_t groups = (a.n_jobs + 7u) / 8u;
    const dim3 grid(groups * 8u * nsub);
#define OSMT_LAUNCH_RASTER(F64, BLK, LAB)                                                                             \
    hipLaunchKernelGGL((k_raster<F64, BLK, LAB>), grid, dim3(NTHREADS), 0, st, a.jobs, a.n_jobs, a.scale, a.ops, a.info, \
                       a.rings, a.pts, a.trav, a.den, a.aux, a.opnv, a.op_blk, a.blk, a.submask, a.sub_rows, a.images,    \
                       a.image_pool, a.n_images, a.out, a.out_tile_stride, a.labels.info, a.labels.job_label_off,         \
                       a.labels.tile_labels, a.labels.tile_label_cnt, a.labels.plane)
    if (out_f64) { /* the raw canvas is the one BEFORE labels (osmt_render_scene_f64) */
        if (a.has_blocks) OSMT_LAUNCH_RASTER(true, true, false); else OSMT_LAUNCH_RASTER(true, false, false);
    } else if (a.labels.info) {
        if (a.has_blocks) OSMT_LAUNCH_RASTER(false, true, true); else OSMT_LAUNCH_RASTER(false, false, true);
    } else {
        if (a.has_blocks) OSMT_LAUNCH_RASTER(false, true, false); else OSMT_LAUNCH_RASTER(false, false, false);
    }
#undef OSMT_LAUNCH_RASTER
    return hipGetLastError();
}

hipError_t osmt_launch_labels(const osmt_label_launch& a, hipStream_t st) {
    if (a.n_labels == 0 || a.n_jobs == 0) return hipSuccess;
    const double4* segs = reinterpret_cast<const double4*>(a.segs);
    hipLaunchKernelGGL(k_label_cover, dim3(a.n_labels), dim3(64), 0, st, a.info, a.n_labels, segs, a.plane_a, a.err);
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
        hipLaunchKernelGGL(k_label_resolve<true>, dim3(a.n_jobs), dim3(OSMT_LABEL_RESOLVE_THREADS), bm_bytes, st, a.info,
                           a.job_label_off, a.n_jobs, a.scale, a.plane_a, a.bitmap, a.ok, a.tile_labels, a.tile_label_cnt);
    } else {
        hipLaunchKernelGGL(k_label_resolve<false>, dim3(a.n_jobs), dim3(OSMT_LABEL_RESOLVE_THREADS), 0, st, a.info,
                           a.job_label_off, a.n_jobs, a.scale, a.plane_a, a.bitmap, a.ok, a.tile_labels, a.tile_label_cnt);
    }
    return hipGetLastError();
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
    if ((W == 256u || W == 512u) && (H % 4u) == 0u && H >= 4u) {
        /* staging capacity of one band: H/4 rows of at most 9 bits per filtered byte */
        const uint32_t band_cap_words = (uint32_t)(((size_t)(H / 4u) * (3u * W + 1u) * 9u + 31u) / 32u + 2u);
        if ((size_t)(11u + 4u * band_cap_words) * 4u + 64u <= out_stride) {
            if (W == 256u)
                hipLaunchKernelGGL((k_png_encode_fast<4>), dim3(n), dim3(256), 0, st, reinterpret_cast<const uint8_t*>(rgba), tile_stride, n, H,
                                   ihdr_crc, reinterpret_cast<uint8_t*>(out), out_stride, band_cap_words, out_len);
            else
                hipLaunchKernelGGL((k_png_encode_fast<8>), dim3(n), dim3(256), 0, st, reinterpret_cast<const uint8_t*>(rgba), tile_stride, n, H,
                                   ihdr_crc, reinterpret_cast<uint8_t*>(out), out_stride, band_cap_words, out_len);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(k_png_encode, dim3(n), dim3(64 * PNG_WAVES), 0, st, reinterpret_cast<const uint8_t*>(rgba), tile_stride, n, W, H, ihdr_crc,
                       reinterpret_cast<uint8_t*>(out), out_stride, out_len);
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
