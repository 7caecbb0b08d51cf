/* osmt_internal.h — device-side records shared by the kernels and the host library. */
#ifndef OSMT_INTERNAL_H
#define OSMT_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/osmtile.h"

/* Sub-tile geometry of k_raster: OSMT_SUB_W x OSMT_SUB_H pixels per workgroup.  Sub-tile
 * coverage masks (osmt_raster_args.submask) have one 32-bit word per sub-tile ROW. */
#define OSMT_SUB_W 32
#ifndef OSMT_SUB_H_LOG2
#define OSMT_SUB_H_LOG2 4
#endif
#define OSMT_SUB_H (1 << OSMT_SUB_H_LOG2)

/* even dash indices (<= 8) + the first dash repeated (opacity_calculator.rs:105-106) */
#define OSMT_MAX_DASH_SEGS 10

/* DashSegment, opacity_calculator.rs:88-96 */
struct osmt_dash_seg {
    double start_from, start_to, end_from, end_to, opacity_mul;
    double orig_a, orig_b; /* original_endpoints (valid when table.has_orig) */
    double r_start, r_end; /* RN(1 / (start_to - start_from)), RN(1 / (end_to - end_from)): the two ramp divisions as osmt_div_exact */
};

/* A cap stub of draw_lines (line.rs:33-57): p1 -> p1.push_away_from(p2, half_width) of the first edge, p2 ->
 * p2.push_away_from(p1, half_width) of the last one; k_opinfo's working record (the stubs reach the binning kernel as the op's
 * last two virtual segments). */
struct osmt_cap_seg {
    int32_t p1x, p1y, p2x, p2y;
    int32_t valid; /* the edge is not degenerate and the cap is Round/Square */
    uint32_t cand_off; /* first slot of the stub's sub-tile window in the op's slice of the stroke arena */
    double denom;  /* center_dist_denom of the stub */
};

/* The per-op constants of a STROKE op (k_opinfo -> k_raster): 192 bytes, written as three whole 64-byte lines.
 * draw_lines builds two OpacityCalculators (line.rs:21-22): `main` for the edges — its DashSegments (up to
 * OSMT_MAX_DASH_SEGS, only when the op has dashes) live in a table of their own, osmt_*_args::dseg[aux index][..], so that an
 * un-dashed op writes and reads nothing of it — and the one for the outer cap stubs, which always has exactly ONE segment
 * (compute_segments over dashes = [0.0]) and total_len 0.  (Until round 4 the record held both tables in full, 1608 bytes,
 * and the cap stubs a second time: k_opinfo wrote five partial cache lines per stroke op — most of its 1.9 GB on config 5.) */
struct alignas(64) osmt_stroke_aux {
    double half_width;
    /* get_opacity_by_center_distance terms for cap_dist == 0 (opacity_calculator.rs:36,171-176):
     * hlw0 = sqrt(h*h - 0*0), feather_from/to/dist and opacity_mul of hlw0 */
    double hlw0, ff0, ft0, fd0, mul0;
    double rfd0; /* RN(1 / fd0): the feather division of the walk becomes osmt_div_exact */
    /* OpacityCalculator `main` minus half_line_width / traveled (opacity_calculator.rs:3-8) */
    int32_t main_n_segs;
    int32_t main_has_orig; /* line cap is Round and use_caps_for_dashes: original_endpoints = Some(..) */
    double main_total_len;
    double main_r_total;   /* RN(1 / total_len): quotient estimate of the exact `dist_rem % total_len` */
    /* opacity_calculator_for_outer_caps: n_segs == 1, total_len == 0 */
    int32_t caps_has_orig;
    int32_t _pad;
    osmt_dash_seg caps_seg;
};
static_assert(sizeof(osmt_stroke_aux) == 192, "three 64-byte lines");

/* Everything k_raster needs to start on an op, in ONE 64-byte record (one s_load_dwordx16): the op header fields it
 * uses and the results of the pre-pass — k_raster never touches ops, rings or points. */
struct osmt_opinfo {
    int32_t x0, y0, x1, y1; /* inclusive extent of the op's points (empty: x0 > x1) */
    uint32_t aux;           /* STROKE: index into the stroke_aux table */
    uint32_t n_edges;       /* total edges over all rings */
    uint32_t first_pt;      /* first point of the op's FIRST ring: a one-ring polygon is binned without touching osmt_op / osmt_ring */
    uint32_t n_rings;       /* osmt_op.n_rings */
    uint8_t kind, cap, color[3], _pad[3]; /* osmt_op.kind / cap / color */
    /* FILL: first 64-byte group (16 row words of one sub-tile) of the op's coverage masks in the fill arena;
     * STROKE: first record of the op in the stroke-record arena */
    uint32_t arena_off;
    uint32_t rec_cap;   /* STROKE: slots of the op in the stroke arena = sum over its virtual segments of the sub-tiles
                         * in each one's window; slot order = segment order (edges, then the two cap stubs), window row-major */
    union {
        struct {
            /* FILL: sub-tile window the masks cover: sr0 | c0 << 8 | ncols << 16 | nsr << 24 (nsr == 0: no covered row inside the tile) */
            uint32_t fill_geom;
            uint32_t image_id; /* FILL_IMAGE: osmt_op.image_id */
        };
        double stroke_ft; /* STROKE: max(|half_width| + 0.5, 1.0), the feather_to the binning bounds a run's reach with */
    };
    double opacity;     /* osmt_op.opacity */
};
static_assert(sizeof(osmt_opinfo) == 64, "osmt_opinfo must be one 64-byte record");

/* ---- pre-pass products (implementation traffic, not algorithmic bytes) --------------------------
 * Fill arena: for every FILL op, for every sub-tile (sr, c) of its window, 16 words = the coverage of rows
 * 16*sr .. 16*sr+15 restricted to columns 32*c .. 32*c+31 (bit x of the word of row y = fill_contour sets pixel (x, y),
 * fill.rs:23-45).  Group index = arena_off + (sr - sr0) * ncols + (c - c0).  Written by k_fill_rows once per op — the
 * rows of an op are evaluated ONCE per tile, not once per sub-tile column.
 *
 * Stroke arena: one SLOT per (virtual segment, sub-tile of its window), written by k_stroke_bin: the step ranges of
 * the segment's perpendicular runs for that sub-tile when it can draw there, a hole otherwise.  The slots of one op
 * are contiguous (arena_off .. + rec_cap) and in segment order — no atomics, the layout is a pure function of the
 * scene; `key` = the sub-tile of a slot (0xFFFFFFFF: hole), kept in its own array so that a wave filters 64 slots with
 * one coalesced load. */
struct alignas(16) osmt_srec {
    int32_t p1x, p1y, p2x, p2y;
    double traveled;      /* line.rs:31, before this edge (0 for a cap stub) */
    double denom;         /* center_dist_denom (line.rs:104) */
    double rdenom;        /* 1 / denom, correctly rounded (exact-division shortcut of the walk) */
    int32_t k_lo0, k_lo1; /* main perpendiculars: steps [k_lo, k_lo + k_n) per side */
    int32_t m_lo0, m_lo1; /* extra perpendiculars (line.rs:152-154): events [m_lo, m_lo + n_x) per side */
    uint16_t k_n0, k_n1, n_x0, n_x1; /* <= sub-tile extent + OSMT_REACH_MAX */
};
static_assert(sizeof(osmt_srec) == 64, "osmt_srec is one 64-byte line: four 16-byte stores by k_stroke_bin, four loads by k_raster");

/* What k_opinfo leaves per VIRTUAL SEGMENT (an edge of a stroke op, or one of its two cap stubs; index = op_vseg[op] + running
 * edge, the stubs last) for the binning kernel, as ONE 48-byte record: three 16-byte stores by the op's lane, three loads by the
 * segment's lane.  (Round 6: with k_opinfo no longer queueing behind its atomics, the bytes it writes show — pre-pass of 256
 * config-5 tiles 2.77 -> 2.73 ms, 64 tiles 0.935 -> 0.906, config 2 0.207 -> 0.2055; profiles/r06_n_vseg_aos_stage_times.txt.) */
struct alignas(16) osmt_vseg {
    int32_t p1x, p1y, p2x, p2y; /* p1 == p2: draws nothing (a degenerate edge, line.rs:73-75, or an invalid stub) */
    double trav;                /* traveled before the edge (line.rs:31); 0 for a stub */
    double den, rden;           /* |p2 - p1| = center_dist_denom (line.rs:104) and its correctly rounded reciprocal */
    uint32_t cand_off;          /* first slot (relative to the op) of the segment's sub-tile window */
    uint32_t vop;               /* the op; bit 31: the segment is a cap stub */
};
static_assert(sizeof(osmt_vseg) == 48, "three 16-byte words");

/* Ops with more than 64 edges get one bounding box per block of 64 consecutive edges (running
 * edge index over all rings): k_fill_rows skips the blocks whose rows miss the rows it is working on. */
struct osmt_blk_bbox {
    int32_t x0, y0, x1, y1; /* over the end points of the block's edges (empty block: x0 > x1) */
};

/* ---- per-sub-tile display lists (k_sublist -> k_raster) -----------------------------------------------------------
 * After the binning kernels have set the exact "op draws into sub-tile" bits, k_sublist turns them round: for every
 * (tile, sub-tile) the ops that draw there, in display-list order (= generation order, drawer.rs:218), each with what
 * k_raster needs to start on it resolved for THAT sub-tile.  k_raster then streams its own short list instead of
 * scanning the op bits of the whole tile (config 5: 9000 bits for ~100 drawing ops) and never touches osmt_opinfo. */
struct alignas(16) osmt_ent {
    uint32_t arena;      /* FILL: first word of the 16 coverage words of THIS sub-tile; STROKE: first slot of the op */
    uint32_t kind_color; /* kind | r << 8 | g << 16 | b << 24 */
    double opacity;
    uint32_t aux;        /* STROKE: index into the stroke_aux table; FILL_IMAGE: image id */
    uint32_t nv;         /* STROKE: slots of the op in the stroke arena (rec_cap) */
    uint32_t stage;      /* k_raster's own use while the entry sits in LDS */
    uint32_t _pad;
};
static_assert(sizeof(osmt_ent) == 32, "osmt_ent is two 16-byte loads");

struct osmt_image_desc {
    uint64_t offset; /* first pixel in the image pool (double4 units) */
    uint32_t width, height;
};

/* ---- label pass ------------------------------------------------------ */
/* One Rasterizer::draw_line call, pre-digested (label_seg_prep, font/rasterizer.rs:27-41):
 * everything that does not depend on the stripe y. */
struct osmt_label_seg {
    double x0, y0;
    double slope, slope_recip; /* (x1 - x0) / delta and its f64::recip */
    double y_min, y_max;
    double sign;               /* +1.0 / -1.0 */
    int32_t yf, yl;            /* floor(y_min) as i32 ..= floor(y_max) as i32; yf > yl: delta == 0.0, no-op */
};
static_assert(sizeof(osmt_label_seg) == 64, "osmt_label_seg must be one 64-byte record");

/* One Labeler::label_entity call with its coverage plane: the dense window
 * [cx0, cx0 + cols) x [ry0, ry1] of the Rasterizer's stripes (rows clipped to labels_bb,
 * tile_pixels.rs:67-72; columns cover every key the clipped rows can receive). */
struct osmt_labelinfo {
    uint32_t seg_off, n_segs;
    int32_t ry0, ry1; /* empty (ry0 > ry1): no text pixels can land inside labels_bb */
    int32_t cx0;
    uint32_t cols;
    uint64_t plane_off; /* first cell of the window in the A pool */
    int32_t icon_x, icon_y; /* get_start_coord (labeler.rs:92-95) */
    uint32_t icon_w, icon_h; /* 0 x 0: no icon */
    uint64_t icon_off;       /* first pixel in the image pool (double4 units) */
    uint8_t has_text, color[3];
    /* windows wider than the LDS band: first cell of a 64-stripe S scratch (k_label_cover_wide); all others: first
     * 64-bit word of the label's coverage bit streams (one per band, see osmt_label_band) */
    uint32_t wide_off;
};
static_assert(sizeof(osmt_labelinfo) == 64, "osmt_labelinfo must be one 64-byte record");

/* Accumulator cells (A and S each) one k_label_cover workgroup keeps in LDS: a label's window is processed in
 * bands of OSMT_LABEL_LDS_CELLS / cols stripes; windows with more columns take k_label_cover_wide.  576: with the
 * hand-over buffers a workgroup takes 20 336 B, eight fit a CU (640 cells: seven; measured 448 .. 896, profiles/). */
#ifndef OSMT_LABEL_LDS_CELLS
#define OSMT_LABEL_LDS_CELLS 576
#endif

/* k_label_resolve -> k_raster: a succeeded label that reaches into the tile, with the box to test sub-tiles against */
struct osmt_tile_label {
    int16_t x0, y0, x1, y1;
    uint32_t label;
    uint32_t _pad;
};

/* One k_label_cover wave: OSMT_LABEL_LDS_CELLS / cols stripes of one label's window, starting at stripe `rbase`.  Stripes
 * never share a cell, so the bands of a label are independent waves; the list is ordered by falling work
 * (draw_line calls to scan), so that the longest bands start first and the kernel does not end on one straggler. */
struct osmt_label_band {
    uint32_t label, rbase;
};
/* Besides the f64 totals a band leaves ONE BIT per cell (total > 0) for k_label_resolve, in cell order, 64 cells
 * per word; band b of a label starts at word wide_off + b * osmt_label_band_words(cols). */
__host__ __device__ static inline uint32_t osmt_label_band_rows(uint32_t cols) {
    const uint32_t r = OSMT_LABEL_LDS_CELLS / cols;
    return r < 64u ? r : 64u;
}
__host__ __device__ static inline uint32_t osmt_label_band_words(uint32_t cols) { return (osmt_label_band_rows(cols) * cols + 63u) / 64u; }

struct osmt_label_launch {
    const osmt_labelinfo* info;
    const osmt_label_band* bands;
    uint32_t n_bands;
    uint32_t n_labels, n_jobs, scale, n_wide;
    const uint32_t* job_label_off;
    const double* segs;
    const uint32_t* wide; /* labels that need k_label_cover_wide */
    double* plane_a;
    unsigned long long* cell_bits; /* coverage bit streams (osmt_label_band_words) */
    double* plane_s_wide;
    uint32_t* bitmap; /* scale > 1 only */
    uint8_t* ok;
    uint32_t* err;
    osmt_tile_label* tile_labels;
    uint32_t* tile_label_cnt;
};

struct osmt_label_args {
    const osmt_labelinfo* info; /* [n_labels] */
    uint32_t n_labels;
    const uint32_t* job_label_off; /* [n_jobs + 1] */
    const osmt_tile_label* tile_labels; /* [n_labels], tile i's entries start at job_label_off[i] (k_label_resolve) */
    const uint32_t* tile_label_cnt;     /* [n_jobs] */
    const double* plane;           /* A pool after k_label_cover: min(a + s_acc, 1.0) per cell, 0 where no key */
};

/* SMALL batches (the one-tile request, a gathered worker group): a tile with at most this many ops gets no per-sub-tile lists
 * from k_sublist — one word of op bits per lane (two rounds) and a ballot give a sub-tile wave its list in op order directly,
 * and a launch (8 us of a 105 us request) is saved.  Big batches keep the lists: there the scattered reads of the op bits by
 * 131 072 waves cost k_raster more (0.615 -> 0.670 ms on config 2) than the list kernel does (0.032 ms). */
#define OSMT_FOLD_MAX_OPS 128u
#define OSMT_FOLD_MAX_JOBS 64u

struct osmt_raster_args {
    const osmt_tile_job* jobs;
    uint32_t n_jobs;
    uint32_t scale;
    const osmt_stroke_aux* aux;
    const osmt_dash_seg* dseg; /* [stroke][OSMT_MAX_DASH_SEGS]: DashSegments of the `main` calculators (dashed ops only) */
    const uint2* hdr;        /* [n_jobs][nsub]: (first entry, entry count) of the sub-tile's list (k_sublist) */
    const osmt_ent* ent;     /* the lists */
    const uint32_t* fmask;   /* fill arena (words) */
    const osmt_srec* srec;   /* stroke arena */
    /* tiles of at most fold_max_ops ops have no lists: their sub-tile waves read the op bits and the op records themselves */
    uint32_t fold_max_ops;   /* 0: every tile has lists */
    uint32_t _pad1;
    const osmt_opinfo* info;
    const uint32_t* submask; /* [op][sub-tile row]: bit sx = the op draws into sub-tile (sx, row) */
    const uint2* skey;       /* per stroke slot: (its sub-tile sy * subs_per_row + sx, or 0xFFFFFFFF for a hole; item count | cap flag << 31) */
    const osmt_image_desc* images;
    const double4* image_pool;
    uint32_t n_images;
    uint32_t out_rgb8;       /* 1: packed RGB8 (the memory of to_rgb_triples' Vec<(u8, u8, u8)>) instead of RGBA8; tile stride a multiple of 4 */
    void* out;
    size_t out_tile_stride; /* bytes (RGBA8 / RGB8 output) */
    osmt_label_args labels;  /* info == NULL: no label pass */
};

/* The per-op pre-pass and the two binning kernels (stage 2). */
struct osmt_prepass_args {
    const osmt_tile_job* jobs;
    uint32_t n_jobs;
    const osmt_op* ops;
    uint32_t n_ops;
    const osmt_ring* rings;
    const int2* pts;
    const double* dashes;
    const uint32_t* op_aux;   /* op -> stroke slot */
    const uint32_t* op_job;   /* op -> job */
    const uint32_t* op_blk;   /* op -> first 64-edge block bbox (0xFFFFFFFF: none) */
    const uint32_t* op_vseg;  /* op -> its first virtual segment (stroke ops that have segments) */
    uint32_t n_vsegs;
    uint32_t scale;
    uint32_t sub_rows;
    uint32_t max_job_ops;  /* most ops of any tile */
    uint32_t fold_max_ops; /* tiles of at most this many ops get no lists (0: all do); max_job_ops <= fold_max_ops = no k_sublist launch */
    osmt_opinfo* info;
    osmt_stroke_aux* aux;
    osmt_dash_seg* dseg; /* [stroke][OSMT_MAX_DASH_SEGS] */
    osmt_blk_bbox* blk;
    uint32_t* submask;
    /* per VIRTUAL SEGMENT (index = op_vseg[op] + running edge index, so ops that share rings do not collide): what the binning
     * kernel needs of an edge / cap stub in ONE record, one level of loads (round 3 went vseg -> table slot -> op -> opinfo ->
     * osmt_op -> ring -> points; rounds 4-5 kept six arrays: six scattered partial-line stores per edge) */
    osmt_vseg* vseg;
    unsigned long long* cursors; /* [0] fill arena (64-byte groups), [1] stroke arena (records), [2] list entries; zeroed by the launcher */
    uint32_t* cnt;      /* [n_jobs][nsub], right behind the cursors (zeroed with them): ops that draw into the sub-tile */
    uint2* hdr;         /* [n_jobs][nsub]: k_sublist's (first entry, count) */
    osmt_ent* ent;      /* list arena */
    unsigned long long ent_cap;
    uint32_t* fmask;
    osmt_srec* srec;
    uint2* skey;
    unsigned long long fmask_cap, srec_cap; /* arena capacities (groups / records); 0 = sizing pass: only the cursors are produced */
    /* host-mapped (pinned, coherent) word of the scene, or NULL: a kernel whose arena reservation does not fit — it cannot,
     * the arenas are sized by the same code; a future change to the binning that breaks the invariant must not show as
     * silently blank tiles — stores an OSMT_PREPASS_ERR_* code here, the host looks at it after its own synchronisation */
    uint32_t* err;
};
#define OSMT_PREPASS_ERR_FILL_ARENA 1u
#define OSMT_PREPASS_ERR_STROKE_ARENA 2u
#define OSMT_PREPASS_ERR_LIST_ARENA 4u

/* zero / n_zero (optional): 32-bit words the kernel clears on the way — the cursors and list counts of the pre-pass that follows */
hipError_t osmt_launch_project(const osmt_tile_job* jobs, const uint32_t* pt_job, const double* latlon, const uint32_t* refs,
                               uint32_t n_pts, double scale, int32_t* pts, hipStream_t st, uint32_t* zero = nullptr, size_t n_zero = 0);
size_t osmt_prepass_zero_words(const osmt_prepass_args& a);
/* point -> job table on the device: pt_job[i] = j for the points of job j, 0xFFFFFFFF for points no job owns */
hipError_t osmt_launch_ptjob(const osmt_tile_job* jobs, uint32_t n_jobs, uint32_t* pt_job, uint32_t n_pts, hipStream_t st);
hipError_t osmt_launch_project_single(const double* latlon, uint32_t n, uint32_t zoom, uint32_t tx, uint32_t ty,
                                      double scale, int32_t* pts, hipStream_t st);
/* k_opinfo -> k_fill_rows -> k_stroke_bin on `st` (sizing pass: k_opinfo only) */
hipError_t osmt_launch_prepass(const osmt_prepass_args& a, hipStream_t st, bool zeroed = false);
hipError_t osmt_launch_raster(const osmt_raster_args& a, bool out_f64, hipStream_t st);
/* label pass: cover (one wave per label) -> resolve (one workgroup per tile, labels in order) */
hipError_t osmt_launch_labels(const osmt_label_launch& a, hipStream_t st);
/* RGBA8 framebuffers -> complete RGB8 PNG files, one per tile, out_len[i] bytes at out + i * out_stride */
hipError_t osmt_launch_png(const void* rgba, size_t tile_stride, uint32_t n, uint32_t W, uint32_t H, uint32_t ihdr_crc, void* out,
                           size_t out_stride, uint32_t* out_len, hipStream_t st);
hipError_t osmt_launch_png_compact(const void* slots, size_t slot_stride, const uint32_t* len, const unsigned long long* off, uint32_t n,
                                   void* blob, hipStream_t st);
hipError_t osmt_launch_copy16(const void* src, void* dst, size_t n16, bool read_only, hipStream_t st);
hipError_t osmt_launch_composite(const void* planes, const double canvas[4], uint32_t n, uint32_t L, uint32_t npx,
                                 void* out, hipStream_t st);

#endif
