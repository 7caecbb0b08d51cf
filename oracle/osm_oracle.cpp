/*
 * osm_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar C++17 restatement of the hot path of dfyz/osm-renderer (Rust), one
 * function per reference function, same operation order, f64 everywhere,
 * compiled with -ffp-contract=off so no FMA is ever formed.  Each function
 * cites the reference lines (relative to /root/reference) it follows.
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg — as the checker / the timed CPU baseline.  libosmtile.so never links it.
 *
 * PARITY PIN STATUS
 *   - projection (tile.rs:88-106): PINNED by the reference's own doctest known-answer
 *     values (src/tile.rs:26-28, 83-86) — tests/test_oracle_kat.py.
 *   - fill / stroke / blend / RGB: PINNED by six crops of the reference's REAL golden images
 *     (tests/rendered/17_expected.png, 18_expected.png) that the oracle reproduces with ZERO
 *     differing pixels from stylesheet parameters + fitted integer vertices
 *     (tests/golden/ref_golden_patches.json, tests/test_reference_golden_patches.py):
 *       stub     852 px   thick AA stroke walk, across-feather, Round caps, 2-generation over, u8
 *       dashed  1239 px   dash pattern 6,8 with Round caps for dashes (use_caps_for_dashes),
 *                         traveled phase, 0.5-opacity third generation (161 colours)
 *       building 990 px   fill-opacity 0.9 polygon + 0.2-px outline of a closed 8-vertex ring
 *       wood    3411 px   opaque 16-vertex polygon: fat Bresenham extents, top-row exclusion, pairing
 *       subway  2416 px   dashes 5,3 with NO caps on a width-2 line, phase carried across a vertex
 *       courtyard 3193 px multipolygon building with a hole: ONE edge table over both rings,
 *                         even-odd pairing in x_min order (outer ring = stand-in rectangle outside
 *                         the crop, inner ring fitted; the hole drawn as a separate op differs)
 *     Inputs were FITTED (the .osm is missing); +-1 px / reversed inputs do not match — see
 *     tests/golden/make_ref_patches.py for what that does and does not prove.
 *     Round 6, from the goldens BELOW z17 (tests/rendered/14_expected.png, 15_expected.png; tests/golden/ref_river_patches.json,
 *     tests/test_reference_golden_rivers.py) — one waterway=river at two zooms, widths 5 and 6, Round caps:
 *       river15_end   1743 px  a free end and its Round cap stub at z15
 *       river14_end    582 px  the same end one zoom lower
 *       river14_bends 4026 px  six vertices, four of them inside the window: draw_lines' join rule (NO joins: consecutive
 *                              segments overlap, the larger alpha of the generation wins) and the walk's direction (drawn the
 *                              other way one pixel differs)
 *   - label pass (font/rasterizer.rs, tile_pixels.rs:131-162 + the for_labels blend, labeler.rs:91-106):
 *     PINNED by the metro-station label "Арбатская" of tests/rendered/17_expected.png — icon + 9 glyphs,
 *     1135 compared pixels, 0 differ — with NOTHING fitted but the node's integer position (read off the
 *     icon): outlines and metrics come from the reference's own font through a restatement of the
 *     stb_truetype calls text_placer.rs makes (tests/golden/make_ref_label_patches.py,
 *     tests/test_reference_golden_labels.py).  A 0.1-px text offset no longer matches.  A second crop (z14,
 *     font-size 9) shows the same label hanging into the tile below its node's tile.  Not covered by a
 *     reference output: labels that COLLIDE (hand-derived cases in tests/test_labels_oracle.py, from the
 *     reference source) and TextPosition::Line placement (host side, outside the oracle).
 *   - image fills (Filler::Image, fill.rs:36-40): PINNED by the landuse=cemetery strip of 18_expected.png
 *     (tests/golden/ref_image_fill_patch.json, tests/test_reference_golden_image_fill.py: 2324 pattern pixels, 0 differ).
 *   - NOT pinned by any reference output (the reference cannot be built here — no rustc/cargo,
 *     crates not vendored — and tests/osm/nano_moscow.osm is absent), one sentence each on why no golden can:
 *       Square / Butt caps        the stylesheet has two `linecap: square` rules, both on man_made=cutline
 *                                 (mapnik.mapcss:319-330, colour #f2efe9): no golden at any zoom (14-18) holds a connected
 *                                 component of that colour larger than 3 px; `butt` occurs in no rule at all;
 *       use_caps_for_dashes=false the styler sets it per style TYPE (styler.rs:95) and every golden was rendered with the
 *                                 Josm type (= true): the other branch never ran for them;
 *       colliding labels          which labels WOULD have been drawn is not recoverable from pixels (needs the .osm);
 *       anything at @2x           tests/rendered/18_2x_expected.png is missing from the mount (.MISSING_LARGE_BLOBS:2).
 *     For these the oracle is checked only against the hand-derived vectors K1..K8 (tests/golden/kat.json,
 *     derived from the reference SOURCE) and against a second, independent Python restatement
 *     (tests/_py_area_model.py).  Status of those parts: PARITY UNPINNED.
 *
 * Rust -> C++ semantics kept on purpose:
 *   f64::round -> std::round (half away from zero); `as i32` / `as u8` ->
 *   saturating, NaN -> 0; f64::max/min -> fmax/fmin (NaN-ignoring); `%` on f64 ->
 *   fmod; powi(2) -> x*x; to_radians -> x * (PI/180); sort_by_key -> stable_sort;
 *   IndexMap -> insertion-ordered containers; i64 for the cross product.
 */
#include "osm_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

constexpr double PI = 3.14159265358979323846264338327950288; /* std::f64::consts::PI */
constexpr uint32_t TILE_SIZE = 256;                          /* tile.rs:6 */
constexpr uint8_t MAX_ZOOM = 18;                             /* tile.rs:5 */

/* Rust `f64 as i32`: saturating, NaN -> 0. */
inline int32_t f64_as_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return (int32_t)v;
}
/* Rust `f64 as u8`: saturating truncation, NaN -> 0. */
inline uint8_t f64_as_u8(double v) {
    if (v != v) return 0;
    if (v >= 255.0) return 255;
    if (v <= 0.0) return 0;
    return (uint8_t)v;
}
/* Rust `f64 as u32` */
inline uint32_t f64_as_u32(double v) {
    if (v != v) return 0;
    if (v >= 4294967295.0) return UINT32_MAX;
    if (v <= 0.0) return 0;
    return (uint32_t)v;
}
/* wrapping i32 arithmetic (Rust release build) */
inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t wmul2(int32_t a) { return (int32_t)((uint32_t)a * 2u); }
inline int32_t wabs(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }

/* ---- tile.rs ----------------------------------------------------------- */

/* tile.rs:88-101 coords_to_xy */
void coords_to_xy(double lat, double lon, uint8_t zoom, double* ox, double* oy) {
    const double lat_rad = lat * (PI / 180.0); /* f64::to_radians */
    const double lon_rad = lon * (PI / 180.0);
    const double x = lon_rad + PI;
    const double y = PI - std::log(std::tan((PI / 4.0) + (lat_rad / 2.0)));
    const double dimension_in_pixels = (double)(TILE_SIZE * (1u << zoom)); /* f64::from(u32) */
    *ox = (x / (2.0 * PI)) * dimension_in_pixels;
    *oy = (y / (2.0 * PI)) * dimension_in_pixels;
}

/* tile.rs:103-106 coords_to_xy_tile_relative */
void coords_to_xy_tile_relative(double lat, double lon, uint8_t zoom, uint32_t tx, uint32_t ty, double* ox,
                                double* oy) {
    double x, y;
    coords_to_xy(lat, lon, zoom, &x, &y);
    *ox = x - (double)(uint32_t)(tx * TILE_SIZE);
    *oy = y - (double)(uint32_t)(ty * TILE_SIZE);
}

/* ---- point.rs ---------------------------------------------------------- */
struct Point {
    int32_t x, y;
    bool operator==(const Point& o) const { return x == o.x && y == o.y; }
    bool operator!=(const Point& o) const { return !(*this == o); }
};

/* point.rs:11-19 Point::from_node */
Point point_from_node(double lat, double lon, uint8_t zoom, uint32_t tx, uint32_t ty, double scale) {
    double x, y;
    coords_to_xy_tile_relative(lat, lon, zoom, tx, ty, &x, &y);
    return Point{f64_as_i32(std::round(x * scale)), f64_as_i32(std::round(y * scale))};
}

/* point.rs:21-25 dist */
double point_dist(const Point& a, const Point& b) {
    const double dx = (double)wsub(a.x, b.x);
    const double dy = (double)wsub(a.y, b.y);
    return std::sqrt(dx * dx + dy * dy);
}

/* point.rs:27-35 push_away_from */
Point push_away_from(const Point& self, const Point& other, double by) {
    const double dist = point_dist(self, other);
    const double push_away_dist = by / dist;
    auto push = [&](int32_t our_c, int32_t other_c) {
        return wadd(our_c, f64_as_i32(std::round((double)wsub(our_c, other_c) * push_away_dist)));
    };
    return Point{push(self.x, other.x), push(self.y, other.y)};
}

/* ---- tile_pixels.rs ---------------------------------------------------- */
struct RgbaColor {
    double r, g, b, a;
};
/* tile_pixels.rs:226-228 */
inline double component_to_opacity(uint8_t c) { return (double)c / 255.0; }
/* tile_pixels.rs:12-19 from_color */
inline RgbaColor from_color(const uint8_t c[3], double opacity) {
    return RgbaColor{opacity * component_to_opacity(c[0]), opacity * component_to_opacity(c[1]),
                     opacity * component_to_opacity(c[2]), opacity};
}
struct NextPixel { /* Option<NextPixel>, tile_pixels.rs:41-44 */
    bool some;
    RgbaColor color;
    size_t generation;
};
struct BoundingBox {
    int32_t min_x, max_x, min_y, max_y;
};

}  // namespace

struct orc_pixels {
    BoundingBox bb, labels_bb;
    size_t scaled_tile_size, scaled_extended_tile_size;
    std::vector<RgbaColor> pixels;
    std::vector<NextPixel> next_pixels;
    size_t generation;
    std::vector<bool> label_generation_statuses;

    /* tile_pixels.rs:57-87 new */
    explicit orc_pixels(size_t scale) {
        scaled_tile_size = TILE_SIZE * scale;
        const int32_t s = (int32_t)scaled_tile_size;
        bb = BoundingBox{0, s - 1, 0, s - 1};
        labels_bb = BoundingBox{bb.min_x - s, bb.max_x + s, bb.min_y - s, bb.max_y + s};
        scaled_extended_tile_size = 3 * TILE_SIZE * scale;
        const size_t pixel_count = scaled_extended_tile_size * scaled_extended_tile_size;
        pixels.assign(pixel_count, RgbaColor{0.0, 0.0, 0.0, 1.0});
        next_pixels.assign(pixel_count, NextPixel{false, {0, 0, 0, 0}, 0});
        generation = 0;
    }
    /* tile_pixels.rs:89-105 reset */
    void reset(bool has_canvas, const uint8_t rgb[3]) {
        const RgbaColor init = has_canvas ? from_color(rgb, 1.0) : RgbaColor{0.0, 0.0, 0.0, 1.0};
        for (auto& p : pixels) p = init;
        for (auto& n : next_pixels) n.some = false;
        generation = 0;
        label_generation_statuses.clear();
    }
    size_t local_coords_to_idx(size_t x, size_t y) const { return y * scaled_extended_tile_size + x; }
    /* tile_pixels.rs:191-199 global_coords_to_idx (for_labels = false) */
    bool global_coords_to_idx(int32_t x, int32_t y, size_t* idx) const {
        if (x < bb.min_x || x > bb.max_x || y < bb.min_y || y > bb.max_y) return false;
        *idx = local_coords_to_idx((size_t)(x - labels_bb.min_x), (size_t)(y - labels_bb.min_y));
        return true;
    }
    /* tile_pixels.rs:205-223 blend_pixel (for_labels = false) */
    void blend_pixel(size_t idx) {
        NextPixel& np = next_pixels[idx];
        if (np.some) {
            RgbaColor& old = pixels[idx];
            const double a = np.color.a;
            RgbaColor nw;
            nw.r = np.color.r + (1.0 - a) * old.r;
            nw.g = np.color.g + (1.0 - a) * old.g;
            nw.b = np.color.b + (1.0 - a) * old.b;
            nw.a = np.color.a + (1.0 - a) * old.a;
            old = nw;
        }
        np.some = false;
    }
    /* tile_pixels.rs:107-129 set_pixel */
    void set_pixel(int32_t x, int32_t y, const RgbaColor& color) {
        size_t idx;
        if (!global_coords_to_idx(x, y, &idx)) return;
        bool from_same_generation = false;
        NextPixel& np = next_pixels[idx];
        if (np.some) {
            if (np.generation == generation) {
                if (color.a > np.color.a) np.color = color;
                from_same_generation = true;
            }
        }
        if (!from_same_generation) {
            blend_pixel(idx);
            next_pixels[idx] = NextPixel{true, color, generation};
        }
    }
    /* tile_pixels.rs:154-158 */
    void blend_unfinished_pixels() {
        for (size_t idx = 0; idx < next_pixels.size(); ++idx) blend_pixel(idx);
    }
    /* tile_pixels.rs:131-148 set_label_pixel (global_coords_to_idx with for_labels = true) */
    bool set_label_pixel(int32_t x, int32_t y, const RgbaColor& color) {
        if (x < labels_bb.min_x || x > labels_bb.max_x || y < labels_bb.min_y || y > labels_bb.max_y) return true;
        const size_t idx = local_coords_to_idx((size_t)(x - labels_bb.min_x), (size_t)(y - labels_bb.min_y));
        const size_t label_generation = label_generation_statuses.size();
        NextPixel& np = next_pixels[idx];
        if (np.some) {
            if (np.generation < label_generation && label_generation_statuses[np.generation]) return false;
        }
        np = NextPixel{true, color, label_generation};
        return true;
    }
    /* tile_pixels.rs:160-162 */
    void bump_label_generation(bool succeeded) { label_generation_statuses.push_back(succeeded); }
    /* tile_pixels.rs:154-158 with for_labels = true; blend_pixel :205-223 */
    void blend_unfinished_label_pixels() {
        for (size_t idx = 0; idx < next_pixels.size(); ++idx) {
            NextPixel& np = next_pixels[idx];
            if (np.some && label_generation_statuses[np.generation]) {
                RgbaColor& old = pixels[idx];
                const double a = np.color.a;
                RgbaColor nw;
                nw.r = np.color.r + (1.0 - a) * old.r;
                nw.g = np.color.g + (1.0 - a) * old.g;
                nw.b = np.color.b + (1.0 - a) * old.b;
                nw.a = np.color.a + (1.0 - a) * old.a;
                old = nw;
            }
            np.some = false;
        }
    }
    /* tile_pixels.rs:164-181 to_rgb_triples */
    void to_rgb(uint8_t* out, bool rgba) const {
        for (size_t y = scaled_tile_size; y < 2 * scaled_tile_size; ++y) {
            for (size_t x = scaled_tile_size; x < 2 * scaled_tile_size; ++x) {
                const RgbaColor& p = pixels[local_coords_to_idx(x, y)];
                auto postdivide = [&](double val) {
                    const double mul = (p.a == 0.0) ? 0.0 : val / p.a;
                    return f64_as_u8(255.0 * mul);
                };
                *out++ = postdivide(p.r);
                *out++ = postdivide(p.g);
                *out++ = postdivide(p.b);
                if (rgba) *out++ = 255;
            }
        }
    }
};

namespace {

/* ---- opacity_calculator.rs --------------------------------------------- */
struct DashSegment { /* :88-96 */
    double start_from, start_to, end_from, end_to, opacity_mul;
    bool has_original_endpoints;
    double orig_a, orig_b;
};
inline bool is_non_trivial_cap(int cap) { /* styler.rs:24-26 */
    return cap == OSMT_CAP_SQUARE || cap == OSMT_CAP_ROUND;
}

struct OpacityCalculator {
    double half_line_width;
    std::vector<DashSegment> dashes;
    double total_dash_len;
    double traveled_distance;

    /* :16-30 new ; :98-143 compute_segments */
    OpacityCalculator(double hlw, const double* dash, int n_dashes /* <0: None */, int cap) {
        half_line_width = hlw;
        double len_before = 0.0;
        if (n_dashes >= 0) {
            /* dash_indexes = (0..len).chain(0..1) — indexing dashes[0] panics in Rust when
             * the list is empty; callers never pass Some([]) (parser yields >= 1 number). */
            std::vector<int> idxs;
            for (int i = 0; i < n_dashes; ++i) idxs.push_back(i);
            idxs.push_back(0);
            for (int idx : idxs) {
                if (idx >= n_dashes) break; /* unreachable for valid input */
                const double d = dash[idx];
                double start = len_before;
                if (idx != 0 || dashes.empty()) len_before += d;
                if (idx % 2 != 0) continue;
                double end = start + d;
                DashSegment s{};
                s.has_original_endpoints = (cap == OSMT_CAP_ROUND);
                s.orig_a = start;
                s.orig_b = end;
                if (is_non_trivial_cap(cap)) {
                    start -= hlw;
                    end += hlw;
                }
                const double midpoint = (start + end) / 2.0;
                s.start_from = std::fmin(start - 0.5, midpoint - 1.0);
                s.start_to = std::fmin(start + 0.5, midpoint);
                s.end_from = std::fmax(end - 0.5, midpoint);
                s.end_to = std::fmax(end + 0.5, midpoint + 1.0);
                s.opacity_mul = std::fmin(end - start, 1.0);
                dashes.push_back(s);
            }
        }
        total_dash_len = len_before;
        traveled_distance = 0.0;
    }

    /* :145-157 */
    static bool get_opacity_by_segment(double dist, const DashSegment& s, double* op) {
        double base;
        if (dist < s.start_from || dist > s.end_to) {
            return false;
        } else if (dist <= s.start_to) {
            base = (dist - s.start_from) / (s.start_to - s.start_from);
        } else if (dist < s.end_from) {
            base = 1.0;
        } else {
            base = (s.end_to - dist) / (s.end_to - s.end_from);
        }
        *op = s.opacity_mul * base;
        return true;
    }
    /* :159-169 */
    static bool get_distance_in_cap(double dist, const DashSegment& s, double* out) {
        if (!s.has_original_endpoints) return false;
        if (dist < s.orig_a)
            *out = s.orig_a - dist;
        else if (dist <= s.orig_b)
            *out = 0.0;
        else
            *out = dist - s.orig_b;
        return true;
    }
    /* :171-185 */
    static double get_opacity_by_center_distance(double center_distance, double hlw) {
        const double feather_from = std::fmax(hlw - 0.5, 0.0);
        const double feather_to = std::fmax(hlw + 0.5, 1.0);
        const double feather_dist = feather_to - feather_from;
        const double opacity_mul = std::fmin(2.0 * hlw, 1.0);
        double v;
        if (center_distance < feather_from)
            v = 1.0;
        else if (center_distance < feather_to)
            v = (feather_to - center_distance) / feather_dist;
        else
            v = 0.0;
        return opacity_mul * v;
    }
    /* :49-80 */
    void get_opacity_by_start_distance(double start_distance, double* opacity, bool* has_cap, double* cap_d) const {
        if (dashes.empty()) {
            *opacity = 1.0;
            *has_cap = false;
            return;
        }
        double dist_rem = traveled_distance + start_distance;
        if (total_dash_len > 0.0) dist_rem = std::fmod(dist_rem, total_dash_len);
        double op = 0.0;
        bool has = false;
        double dic = 0.0;
        for (const DashSegment& d : dashes) {
            double o;
            if (get_opacity_by_segment(dist_rem, d, &o)) {
                op = std::fmax(op, o);
                double dist;
                if (get_distance_in_cap(dist_rem, d, &dist)) {
                    if (!has || dist < dic) {
                        has = true;
                        dic = dist;
                    }
                }
            }
        }
        *opacity = op;
        *has_cap = has;
        *cap_d = dic;
    }
    /* :32-43 calculate */
    void calculate(double center_distance, double start_distance, double* opacity, bool* is_in_line) const {
        double sd_op, cap_d = 0.0;
        bool has_cap;
        get_opacity_by_start_distance(start_distance, &sd_op, &has_cap, &cap_d);
        const double cap_dist = has_cap ? cap_d : 0.0; /* unwrap_or_default */
        const double hlw = std::sqrt(half_line_width * half_line_width - cap_dist * cap_dist);
        const double cd = get_opacity_by_center_distance(center_distance, hlw);
        *opacity = std::fmin(sd_op, cd);
        *is_in_line = cd > 0.0;
    }
};

/* ---- fill.rs ----------------------------------------------------------- */
struct Edge { /* :108-112 */
    int32_t x_min, x_max;
    bool is_poisoned;
};
/* IndexMap<i32, IndexMap<usize, Edge>> (:106): insertion-ordered in both levels.  Edges
 * arrive in increasing edge_idx, so an edge's record on a row, if present, is the row's
 * last one. */
struct EdgesByY {
    std::vector<int32_t> ys;
    std::vector<std::vector<std::pair<size_t, Edge>>> rows;
    std::unordered_map<int32_t, size_t> index;
    std::vector<std::pair<size_t, Edge>>& row(int32_t y) {
        auto it = index.find(y);
        if (it == index.end()) {
            index.emplace(y, rows.size());
            ys.push_back(y);
            rows.emplace_back();
            return rows.back();
        }
        return rows[it->second];
    }
};

/* fill.rs:51-104 draw_line */
void fill_draw_line(size_t edge_idx, const Point& p1, const Point& p2, EdgesByY& y_to_edges, int32_t min_y,
                    int32_t max_y) {
    const int32_t dx = wabs(wsub(p2.x, p1.x));
    const int32_t dy = -wabs(wsub(p2.y, p1.y));
    const int32_t sx = (p1.x < p2.x) ? 1 : -1;
    const int32_t sy = (p1.y < p2.y) ? 1 : -1;
    int32_t err = wadd(dx, dy);
    Point cur = p1;
    for (;;) {
        const bool is_start = cur == p1;
        const bool is_end = cur == p2;
        bool is_poisoned;
        if (is_start)
            is_poisoned = p1.y <= p2.y;
        else if (is_end)
            is_poisoned = p2.y <= p1.y;
        else
            is_poisoned = false;

        if (cur.y >= min_y && cur.y <= max_y) {
            auto& row = y_to_edges.row(cur.y);
            if (row.empty() || row.back().first != edge_idx) row.push_back({edge_idx, Edge{cur.x, cur.x, is_poisoned}});
            Edge& e = row.back().second;
            e.x_min = std::min(e.x_min, cur.x);
            e.x_max = std::max(e.x_max, cur.x);
            e.is_poisoned |= is_poisoned;
        }
        if (is_end) break;
        const int32_t e2 = wmul2(err);
        if (e2 >= dy) {
            err = wadd(err, dy);
            cur.x = wadd(cur.x, sx);
        }
        if (e2 <= dx) {
            err = wadd(err, dx);
            cur.y = wadd(cur.y, sy);
        }
    }
}

/* fill.rs:16-47 fill_contour */
void fill_contour(const Point* pairs /* 2 per edge */, size_t n_pairs, const uint8_t color[3], const orc_icon* icon,
                  double opacity, orc_pixels& pixels) {
    EdgesByY y_to_edges;
    for (size_t idx = 0; idx < n_pairs; ++idx)
        fill_draw_line(idx, pairs[2 * idx], pairs[2 * idx + 1], y_to_edges, pixels.bb.min_y, pixels.bb.max_y);

    for (size_t r = 0; r < y_to_edges.ys.size(); ++r) {
        const int32_t y = y_to_edges.ys[r];
        std::vector<const Edge*> good;
        for (auto& kv : y_to_edges.rows[r])
            if (!kv.second.is_poisoned) good.push_back(&kv.second);
        std::stable_sort(good.begin(), good.end(), [](const Edge* a, const Edge* b) { return a->x_min < b->x_min; });
        size_t idx = 0;
        while (idx + 1 < good.size()) {
            const Edge* e1 = good[idx];
            const Edge* e2 = good[idx + 1];
            const int32_t from_x = std::max(e1->x_min, pixels.bb.min_x);
            const int32_t to_x = std::min(e2->x_max, pixels.bb.max_x) + 1;
            for (int32_t x = from_x; x < to_x; ++x) {
                RgbaColor fill;
                if (!icon) {
                    fill = from_color(color, opacity);
                } else {
                    const size_t icon_x = (size_t)x % icon->width;
                    const size_t icon_y = (size_t)y % icon->height;
                    const double* px = icon->rgba + 4 * (icon_y * icon->width + icon_x);
                    fill = RgbaColor{px[0], px[1], px[2], px[3]};
                }
                pixels.set_pixel(x, y, fill);
            }
            idx += 2;
        }
    }
}

/* ---- line.rs ----------------------------------------------------------- */

/* line.rs:65-158 draw_line */
void stroke_draw_line(const Point& p1, const Point& p2, const uint8_t color[3], double initial_opacity,
                      const OpacityCalculator& oc, orc_pixels& pixels) {
    if (p1 == p2) return;
    auto get_inc = [](int32_t from, int32_t to) { return from <= to ? 1 : -1; };
    const int32_t dx = wabs(wsub(p2.x, p1.x)), dy = wabs(wsub(p2.y, p1.y));
    const bool swap = dx > dy;
    /* swap_x_y_if_needed(a, b, swap) = swap ? (b, a) : (a, b)   (:160-166) */
    int32_t mn = swap ? p1.y : p1.x, mx = swap ? p1.x : p1.y;
    const int32_t mn_last = swap ? p2.y : p2.x, mx_last = swap ? p2.x : p2.y;
    const int32_t mn_delta = swap ? dy : dx, mx_delta = swap ? dx : dy;
    const int32_t inc_x = get_inc(p1.x, p2.x), inc_y = get_inc(p1.y, p2.y);
    const int32_t mn_inc = swap ? inc_y : inc_x, mx_inc = swap ? inc_x : inc_y;

    int32_t error = 0, p_error = 0;
    auto update_error = [&](int32_t& e) {
        bool was_corrected;
        if (wadd(e, wmul2(mn_delta)) > mx_delta) {
            e = wsub(e, wmul2(mx_delta));
            was_corrected = true;
        } else {
            was_corrected = false;
        }
        e = wadd(e, wmul2(mn_delta));
        return was_corrected;
    };

    const int64_t numer_const = (int64_t)p2.x * (int64_t)p1.y - (int64_t)p2.y * (int64_t)p1.x;
    const int64_t sdx = (int64_t)p2.x - (int64_t)p1.x, sdy = (int64_t)p2.y - (int64_t)p1.y;
    const double dx_float = (double)dx, dy_float = (double)dy;
    const double center_dist_denom = std::sqrt(dy_float * dy_float + dx_float * dx_float);

    auto draw_perpendiculars = [&](int32_t mn_, int32_t mx_, int32_t p_error_) {
        auto draw_one_perpendicular = [&](int32_t mul) {
            int32_t p_mn = mx_;
            int32_t p_mx = mn_;
            int32_t err = mul * p_error_;
            for (;;) {
                const int32_t perp_x = swap ? p_mn : p_mx;
                const int32_t perp_y = swap ? p_mx : p_mn;
                const Point cur{perp_x, perp_y};
                const int64_t non_const = sdy * (int64_t)perp_x - sdx * (int64_t)perp_y;
                const int64_t raw = numer_const + non_const;
                const double center_dist = std::fabs((double)raw) / center_dist_denom;
                const double long_start_dist = point_dist(cur, p1);
                const double short_start_dist =
                    std::sqrt(std::fmax(long_start_dist * long_start_dist - center_dist * center_dist, 0.0));
                double op;
                bool in_line;
                oc.calculate(center_dist, short_start_dist, &op, &in_line);
                if (!in_line) break;
                pixels.set_pixel(cur.x, cur.y, from_color(color, initial_opacity * op));
                if (update_error(err)) p_mn = wsub(p_mn, mul * mx_inc);
                p_mx = wadd(p_mx, mul * mn_inc);
            }
        };
        draw_one_perpendicular(1);
        draw_one_perpendicular(-1);
    };

    for (;;) {
        draw_perpendiculars(mn, mx, p_error);
        if (mn == mn_last && mx == mx_last) break;
        if (update_error(error)) {
            mn = wadd(mn, mn_inc);
            if (update_error(p_error)) draw_perpendiculars(mn, mx, p_error);
        }
        mx = wadd(mx, mx_inc);
    }
}

/* line.rs:9-61 draw_lines */
void draw_lines(const Point* pairs, size_t n_pairs, double width, const uint8_t color[3], double opacity,
                const double* dashes, int n_dashes, int cap, bool use_caps_for_dashes, orc_pixels& pixels) {
    const double half_width = width / 2.0;
    const int cap_for_dashes = use_caps_for_dashes ? cap : OSMT_CAP_NONE;
    OpacityCalculator oc(half_width, dashes, n_dashes, cap_for_dashes);
    const double zero = 0.0;
    const OpacityCalculator oc_caps(half_width, &zero, 1, cap);
    const bool has_caps = is_non_trivial_cap(cap);
    bool first = true;
    for (size_t i = 0; i < n_pairs; ++i) {
        const Point& p1 = pairs[2 * i];
        const Point& p2 = pairs[2 * i + 1];
        stroke_draw_line(p1, p2, color, opacity, oc, pixels);
        oc.traveled_distance += point_dist(p1, p2);
        if (p1 != p2 && has_caps) {
            if (first) {
                const Point cap_end = push_away_from(p1, p2, half_width);
                stroke_draw_line(p1, cap_end, color, opacity, oc_caps, pixels);
            }
            if (i + 1 == n_pairs) { /* peek().is_none() */
                const Point cap_end = push_away_from(p2, p1, half_width);
                stroke_draw_line(p2, cap_end, color, opacity, oc_caps, pixels);
            }
        }
        first = false;
    }
}

/* ---- point_pairs.rs + drawer.rs ----------------------------------------- */

Point batch_point(const osmt_batch* b, const osmt_tile_job& job, uint32_t pt) {
    if (b->coord_kind == OSMT_COORD_POINT_I32) return Point{b->points[2 * (size_t)pt], b->points[2 * (size_t)pt + 1]};
    if (b->coord_kind == OSMT_COORD_NODE_REF) { /* Way::get_node(idx) -> Node (reader.rs:291-336) -> Point::from_node */
        const size_t n = b->node_refs[pt];
        return point_from_node(b->nodes[2 * n], b->nodes[2 * n + 1], job.zoom, job.x, job.y, (double)b->scale);
    }
    return point_from_node(b->latlon[2 * (size_t)pt], b->latlon[2 * (size_t)pt + 1], job.zoom, job.x, job.y,
                           (double)b->scale);
}

/* point_pairs.rs:11-41: edges of all rings, concatenated */
void op_point_pairs(const osmt_batch* b, const osmt_tile_job& job, const osmt_op& op, std::vector<Point>& pairs) {
    pairs.clear();
    for (uint32_t r = 0; r < op.n_rings; ++r) {
        const osmt_ring& ring = b->rings[op.ring_off + r];
        for (uint32_t i = 1; i < ring.n_pts; ++i) {
            pairs.push_back(batch_point(b, job, ring.first_pt + i - 1));
            pairs.push_back(batch_point(b, job, ring.first_pt + i));
        }
    }
}

/* ---- font/rasterizer.rs ------------------------------------------------------ */
struct Stripe { /* :6-10 */
    std::map<int32_t, double> a, s;
};
struct Rasterizer { /* :14-17 */
    std::map<int32_t, Stripe> stripes;
    uint8_t color[3];

    /* :27-88 draw_line */
    void draw_line(double x0, double y0, double x1, double y1) {
        const double delta = y1 - y0;
        if (delta == 0.0) return;
        const double sign = (y0 <= y1) ? 1.0 : -1.0;
        const double slope = (x1 - x0) / delta;
        const double slope_recip = 1.0 / slope; /* f64::recip */
        auto eval_x_at_y = [&](double y) { return x0 + (y - y0) * slope; };
        auto eval_y_at_x = [&](double x) { return y0 + (x - x0) * slope_recip; };
        const double y_min = std::fmin(y0, y1);
        const double y_max = std::fmax(y0, y1);
        const int32_t yf = f64_as_i32(std::floor(y_min)), yl = f64_as_i32(std::floor(y_max));
        for (int64_t yy = yf; yy <= yl; ++yy) {
            const int32_t y = (int32_t)yy;
            Stripe& cur = stripes[y]; /* entry(y).or_default() */
            const double y_bottom = std::fmax((double)y, y_min);
            const double y_top = std::fmin((double)wadd(y, 1), y_max);
            const double y_delta = y_top - y_bottom;
            const double x_at_bottom = eval_x_at_y(y_bottom);
            const double x_at_top = eval_x_at_y(y_top);
            bool flip_edge;
            double x_smallest, x_largest;
            if (x_at_bottom <= x_at_top) {
                flip_edge = false, x_smallest = x_at_bottom, x_largest = x_at_top;
            } else {
                flip_edge = true, x_smallest = x_at_top, x_largest = x_at_bottom;
            }
            const int32_t x_to = f64_as_i32(std::floor(x_largest));
            for (int64_t xx = f64_as_i32(std::floor(x_smallest)); xx <= x_to; ++xx) {
                const int32_t x = (int32_t)xx;
                const double x_left = std::fmax((double)x, x_smallest);
                const double x_next = (double)wadd(x, 1);
                const double x_right = std::fmin(x_next, x_largest);
                double pixel_area = (x_next - x_right) * y_delta;
                const double trapezoid_width = x_right - x_left;
                if (trapezoid_width > 0.0) {
                    const double y_at_left = eval_y_at_x(x_left);
                    const double y_at_right = eval_y_at_x(x_right);
                    const double trapezoid_height =
                        flip_edge ? (y_top - y_at_left) + (y_top - y_at_right) : (y_at_left - y_bottom) + (y_at_right - y_bottom);
                    pixel_area += trapezoid_width * trapezoid_height / 2.0;
                }
                auto it = cur.a.emplace(x, 0.0).first; /* entry(x).or_insert(0.0) */
                it->second += sign * pixel_area;
            }
            auto it = cur.s.emplace(wadd(x_to, 1), 0.0).first;
            it->second += sign * y_delta;
        }
    }

    /* :90-113 draw_quad */
    template <class F>
    static void flatten_quad(double x0, double y0, double x1, double y1, double x2, double y2, F&& line) {
        auto dist_between = [](double xa, double ya, double xb, double yb) { return std::hypot(std::fabs(xa - xb), std::fabs(ya - yb)); };
        const double d01 = dist_between(x0, y0, x1, y1);
        const double d12 = dist_between(x1, y1, x2, y2);
        const double d02 = dist_between(x0, y0, x2, y2);
        if ((d01 + d12) <= 1.0001 * d02) {
            line(x0, y0, x2, y2);
            return;
        }
        auto midpoint = [](double c1, double c2) { return (c1 + c2) / 2.0; };
        const double m01_x = midpoint(x0, x1), m01_y = midpoint(y0, y1);
        const double m12_x = midpoint(x1, x2), m12_y = midpoint(y1, y2);
        const double m012_x = midpoint(m01_x, m12_x), m012_y = midpoint(m01_y, m12_y);
        flatten_quad(x0, y0, m01_x, m01_y, m012_x, m012_y, line);
        flatten_quad(m012_x, m012_y, m12_x, m12_y, x2, y2, line);
    }

    /* :115-147 save_to_figure; `visit(x, y, total)` returns false to abort like set_label_pixel */
    template <class F>
    bool for_each_pixel(F&& visit) const {
        for (const auto& ys : stripes) {
            const Stripe& stripe = ys.second;
            auto a_it = stripe.a.begin();
            auto s_it = stripe.s.begin();
            double s_acc = 0.0;
            int32_t x_min = INT32_MAX, x_max = INT32_MIN;
            if (!stripe.a.empty()) {
                x_min = std::min(x_min, stripe.a.begin()->first);
                x_max = std::max(x_max, stripe.a.rbegin()->first);
            }
            if (!stripe.s.empty()) {
                x_min = std::min(x_min, stripe.s.begin()->first);
                x_max = std::max(x_max, stripe.s.rbegin()->first);
            }
            for (int64_t xx = x_min; xx <= x_max; ++xx) {
                const int32_t x = (int32_t)xx;
                double sv = 0.0, av = 0.0;
                if (s_it != stripe.s.end() && s_it->first == x) sv = (s_it++)->second;
                s_acc += sv;
                if (a_it != stripe.a.end() && a_it->first == x) av = (a_it++)->second;
                const double total = std::fmin(av + s_acc, 1.0);
                if (total > 0.0 && !visit(x, ys.first, total)) return false;
            }
        }
        return true;
    }
    bool save_to_figure(orc_pixels& pixels) const {
        return for_each_pixel([&](int32_t x, int32_t y, double total) { return pixels.set_label_pixel(x, y, from_color(color, total)); });
    }
};

/* labeler.rs:91-106 draw_icon */
bool draw_icon(const orc_icon& icon, double center_x, double center_y, orc_pixels& pixels) {
    auto get_start_coord = [](double coord, uint32_t dimension) { return f64_as_i32(coord - ((double)dimension / 2.0)); };
    const int32_t start_x = get_start_coord(center_x, icon.width);
    const int32_t start_y = get_start_coord(center_y, icon.height);
    for (uint32_t x = 0; x < icon.width; ++x)
        for (uint32_t y = 0; y < icon.height; ++y) {
            const double* px = icon.rgba + 4 * ((size_t)y * icon.width + x);
            if (!pixels.set_label_pixel(wadd(start_x, (int32_t)x), wadd(start_y, (int32_t)y), RgbaColor{px[0], px[1], px[2], px[3]}))
                return false;
        }
    return true;
}

/* labeler.rs:16-38 label_entity over the display-list form of one label */
bool label_entity(const osmt_label& lb, const double* segs, const orc_icon* icons, size_t n_icons, orc_pixels& px) {
    bool succeeded;
    bool icon_ok = true; /* label_with_icon: Some(..) unless draw_icon failed (:40-67) */
    if (lb.has_icon && lb.image_id < n_icons) icon_ok = draw_icon(icons[lb.image_id], lb.icon_center_x, lb.icon_center_y, px);
    if (icon_ok) {
        if (lb.has_text) { /* label_with_text -> TextPlacer::place -> save_to_figure (:69-89) */
            Rasterizer r;
            std::memcpy(r.color, lb.text_color, 3);
            for (uint32_t i = 0; i < lb.n_segs; ++i) {
                const double* q = segs + 4 * ((size_t)lb.seg_off + i);
                r.draw_line(q[0], q[1], q[2], q[3]);
            }
            succeeded = r.save_to_figure(px);
        } else {
            succeeded = true;
        }
    } else {
        succeeded = false;
    }
    px.bump_label_generation(succeeded);
    return succeeded;
}

/* drawer.rs:107-125: draw_labels + blend_unfinished_pixels(true) for one tile */
void label_job_into(const osmt_label_batch* lb, size_t job_idx, const orc_icon* icons, size_t n_icons, orc_pixels& px,
                    uint8_t* out_status) {
    if (!lb || lb->n_labels == 0) return;
    for (uint32_t i = lb->job_label_off[job_idx]; i < lb->job_label_off[job_idx + 1]; ++i) {
        const bool ok = label_entity(lb->labels[i], lb->segs, icons, n_icons, px);
        if (out_status) out_status[i] = ok ? 1 : 0;
    }
    px.blend_unfinished_label_pixels();
}

/* drawer.rs:60-131 draw_to_pixels (display-list form; labels excluded) */
void render_job_into(const osmt_batch* b, size_t job_idx, const orc_icon* icons, size_t n_icons, orc_pixels& px) {
    const osmt_tile_job& job = b->jobs[job_idx];
    px.reset(job.has_canvas != 0, job.canvas_rgb);
    std::vector<Point> pairs;
    for (uint32_t k = 0; k < job.n_ops; ++k) {
        const osmt_op& op = b->ops[job.op_off + k];
        op_point_pairs(b, job, op, pairs);
        const size_t n_pairs = pairs.size() / 2;
        switch (op.kind) {
            case OSMT_OP_FILL_COLOR:
                fill_contour(pairs.data(), n_pairs, op.color, nullptr, op.opacity, px);
                break;
            case OSMT_OP_FILL_IMAGE:
                if (op.image_id < n_icons) fill_contour(pairs.data(), n_pairs, op.color, &icons[op.image_id], op.opacity, px);
                break;
            case OSMT_OP_STROKE:
                draw_lines(pairs.data(), n_pairs, op.width, op.color, op.opacity, b->dashes + op.dashes_off,
                           op.has_dashes ? (int)op.n_dashes : -1, op.cap, op.use_caps_for_dashes != 0, px);
                break;
            default:
                break;
        }
        px.generation += 1; /* bump_generation, drawer.rs:218 */
    }
    px.blend_unfinished_pixels(); /* drawer.rs:104 */
}

}  // namespace

/* ======================================================================== */
extern "C" {

void orc_coords_to_xy(double lat, double lon, uint8_t zoom, double* x, double* y) { coords_to_xy(lat, lon, zoom, x, y); }
void orc_coords_to_xy_tile_relative(double lat, double lon, uint8_t zoom, uint32_t tx, uint32_t ty, double* x,
                                    double* y) {
    coords_to_xy_tile_relative(lat, lon, zoom, tx, ty, x, y);
}
void orc_project_points(const double* latlon, size_t n, uint8_t zoom, uint32_t tx, uint32_t ty, double scale,
                        int32_t* xy) {
    for (size_t i = 0; i < n; ++i) {
        const Point p = point_from_node(latlon[2 * i], latlon[2 * i + 1], zoom, tx, ty, scale);
        xy[2 * i] = p.x;
        xy[2 * i + 1] = p.y;
    }
}
/* tile.rs:30-38 */
void orc_coords_to_max_zoom_tile(double lat, double lon, uint32_t* x, uint32_t* y) {
    double fx, fy;
    coords_to_xy(lat, lon, MAX_ZOOM, &fx, &fy);
    *x = f64_as_u32(fx) / TILE_SIZE;
    *y = f64_as_u32(fy) / TILE_SIZE;
}
void orc_push_away_from(const int32_t s[2], const int32_t o[2], double by, int32_t out[2]) {
    const Point r = push_away_from(Point{s[0], s[1]}, Point{o[0], o[1]}, by);
    out[0] = r.x;
    out[1] = r.y;
}

orc_pixels* orc_pixels_new(uint32_t scale) { return new orc_pixels(scale); }
void orc_pixels_free(orc_pixels* p) { delete p; }
void orc_pixels_reset(orc_pixels* p, int has_canvas, uint8_t r, uint8_t g, uint8_t b) {
    const uint8_t c[3] = {r, g, b};
    p->reset(has_canvas != 0, c);
}
void orc_set_pixel(orc_pixels* p, int32_t x, int32_t y, const double c[4]) {
    p->set_pixel(x, y, RgbaColor{c[0], c[1], c[2], c[3]});
}
void orc_bump_generation(orc_pixels* p) { p->generation += 1; }
void orc_blend_unfinished_pixels(orc_pixels* p) { p->blend_unfinished_pixels(); }
uint32_t orc_dimension(const orc_pixels* p) { return (uint32_t)p->scaled_tile_size; }
void orc_to_rgb_triples(const orc_pixels* p, uint8_t* rgb) { p->to_rgb(rgb, false); }
void orc_read_pixels_f64(const orc_pixels* p, double* out) {
    const size_t s = p->scaled_tile_size;
    for (size_t y = 0; y < s; ++y)
        for (size_t x = 0; x < s; ++x) {
            const RgbaColor& c = p->pixels[p->local_coords_to_idx(x + s, y + s)];
            *out++ = c.r;
            *out++ = c.g;
            *out++ = c.b;
            *out++ = c.a;
        }
}
void orc_read_pending_alpha(const orc_pixels* p, uint64_t gen, double* out) {
    const size_t s = p->scaled_tile_size;
    for (size_t y = 0; y < s; ++y)
        for (size_t x = 0; x < s; ++x) {
            const NextPixel& n = p->next_pixels[p->local_coords_to_idx(x + s, y + s)];
            *out++ = (n.some && n.generation == gen) ? n.color.a : 0.0;
        }
}

void orc_fill_contour(orc_pixels* p, const int32_t* pairs, size_t n_pairs, const uint8_t color[3],
                      const orc_icon* icon, double opacity) {
    fill_contour(reinterpret_cast<const Point*>(pairs), n_pairs, color, icon, opacity, *p);
}
void orc_draw_lines(orc_pixels* p, const int32_t* pairs, size_t n_pairs, double width, const uint8_t color[3],
                    double opacity, const double* dashes, int n_dashes, int cap, int use_caps_for_dashes) {
    draw_lines(reinterpret_cast<const Point*>(pairs), n_pairs, width, color, opacity, dashes, n_dashes, cap,
               use_caps_for_dashes != 0, *p);
}
void orc_opacity_calculate(double half_width, const double* dashes, int n_dashes, int cap, double traveled,
                           double center_distance, double start_distance, double* opacity, int* is_in_line) {
    OpacityCalculator oc(half_width, dashes, n_dashes, cap);
    oc.traveled_distance = traveled;
    bool in;
    oc.calculate(center_distance, start_distance, opacity, &in);
    *is_in_line = in ? 1 : 0;
}
size_t orc_fill_edge_walk(const int32_t a[2], const int32_t b[2], int32_t* out_xy, size_t cap) {
    /* same walk as fill_draw_line, recording every visited pixel */
    const Point p1{a[0], a[1]}, p2{b[0], b[1]};
    const int32_t dx = wabs(wsub(p2.x, p1.x));
    const int32_t dy = -wabs(wsub(p2.y, p1.y));
    const int32_t sx = (p1.x < p2.x) ? 1 : -1;
    const int32_t sy = (p1.y < p2.y) ? 1 : -1;
    int32_t err = wadd(dx, dy);
    Point cur = p1;
    size_t n = 0;
    for (;;) {
        if (n < cap) {
            out_xy[2 * n] = cur.x;
            out_xy[2 * n + 1] = cur.y;
        }
        ++n;
        if (cur == p2) break;
        const int32_t e2 = wmul2(err);
        if (e2 >= dy) {
            err = wadd(err, dy);
            cur.x = wadd(cur.x, sx);
        }
        if (e2 <= dx) {
            err = wadd(err, dx);
            cur.y = wadd(cur.y, sy);
        }
    }
    return n;
}

int orc_render_job_labels(const osmt_batch* batch, const osmt_label_batch* labels, size_t job_idx, const orc_icon* icons,
                          size_t n_icons, uint8_t* out_rgba, double* out_f64, uint8_t* out_status) {
    if (!batch || job_idx >= batch->n_jobs) return -1;
    orc_pixels px(batch->scale);
    render_job_into(batch, job_idx, icons, n_icons, px);
    label_job_into(labels, job_idx, icons, n_icons, px, out_status);
    if (out_rgba) px.to_rgb(out_rgba, true);
    if (out_f64) orc_read_pixels_f64(&px, out_f64);
    return 0;
}
int orc_render_job(const osmt_batch* batch, size_t job_idx, const orc_icon* icons, size_t n_icons, uint8_t* out_rgba,
                   double* out_f64) {
    return orc_render_job_labels(batch, nullptr, job_idx, icons, n_icons, out_rgba, out_f64, nullptr);
}

size_t orc_rasterizer_pixels(const double* segs, size_t n_segs, int32_t* out_xy, double* out_total, size_t cap) {
    Rasterizer r;
    r.color[0] = r.color[1] = r.color[2] = 0;
    for (size_t i = 0; i < n_segs; ++i) r.draw_line(segs[4 * i], segs[4 * i + 1], segs[4 * i + 2], segs[4 * i + 3]);
    size_t n = 0;
    r.for_each_pixel([&](int32_t x, int32_t y, double total) {
        if (n < cap) {
            out_xy[2 * n] = x;
            out_xy[2 * n + 1] = y;
            out_total[n] = total;
        }
        ++n;
        return true;
    });
    return n;
}

size_t orc_flatten_quad(const double q[6], double* out_segs, size_t cap) {
    size_t n = 0;
    Rasterizer::flatten_quad(q[0], q[1], q[2], q[3], q[4], q[5], [&](double x0, double y0, double x1, double y1) {
        if (n < cap) {
            out_segs[4 * n] = x0;
            out_segs[4 * n + 1] = y0;
            out_segs[4 * n + 2] = x1;
            out_segs[4 * n + 3] = y1;
        }
        ++n;
    });
    return n;
}

int orc_render_batch(const osmt_batch* batch, size_t first, size_t count, const orc_icon* icons, size_t n_icons,
                     uint8_t* out_rgba, size_t out_tile_stride, int threads) {
    return orc_render_batch_labels(batch, nullptr, first, count, icons, n_icons, out_rgba, out_tile_stride, threads, nullptr);
}

int orc_render_batch_labels(const osmt_batch* batch, const osmt_label_batch* labels, size_t first, size_t count,
                            const orc_icon* icons, size_t n_icons, uint8_t* out_rgba, size_t out_tile_stride, int threads,
                            uint8_t* out_status) {
    if (!batch || first + count > batch->n_jobs) return -1;
    if (threads < 1) threads = 1;
    auto worker = [&](int tid) {
        orc_pixels px(batch->scale); /* one TilePixels per worker (http_server.rs:69-72) */
        for (size_t i = (size_t)tid; i < count; i += (size_t)threads) {
            render_job_into(batch, first + i, icons, n_icons, px);
            label_job_into(labels, first + i, icons, n_icons, px, out_status);
            px.to_rgb(out_rgba + i * out_tile_stride, true);
        }
    };
    if (threads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
        for (auto& t : pool) t.join();
    }
    return 0;
}

/* A persistent worker pool: `threads` TilePixels allocated ONCE, like the reference's server, which builds one
 * TilePixels per worker thread at start-up (http_server.rs:69-72) and reuses it for every request.  The cpu_baseline
 * leg of bench.py creates the pool (and the output array) before its timer starts. */
struct orc_pool {
    std::vector<orc_pixels*> px;
};

orc_pool* orc_pool_create(int threads, uint32_t scale) {
    if (threads < 1) threads = 1;
    orc_pool* p = new orc_pool();
    p->px.assign((size_t)threads, nullptr);
    /* each worker builds (and first-touches) its own canvas, as the server's worker threads do */
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([p, t, scale] { p->px[(size_t)t] = new orc_pixels(scale); });
    for (auto& t : th) t.join();
    return p;
}

void orc_pool_free(orc_pool* p) {
    if (!p) return;
    for (orc_pixels* q : p->px) delete q;
    delete p;
}

int orc_pool_threads(const orc_pool* p) { return p ? (int)p->px.size() : 0; }

int orc_pool_render(orc_pool* pool, const osmt_batch* batch, const osmt_label_batch* labels, size_t first, size_t count,
                    const orc_icon* icons, size_t n_icons, uint8_t* out_rgba, size_t out_tile_stride, uint8_t* out_status) {
    if (!pool || !batch || first + count > batch->n_jobs) return -1;
    const int threads = (int)pool->px.size();
    for (orc_pixels* q : pool->px)
        if (q->scaled_tile_size != (size_t)TILE_SIZE * batch->scale) return -2;
    auto worker = [&](int tid) {
        orc_pixels& px = *pool->px[(size_t)tid];
        for (size_t i = (size_t)tid; i < count; i += (size_t)threads) { /* tiles dealt round-robin (http_server.rs:105-108) */
            render_job_into(batch, first + i, icons, n_icons, px);
            label_job_into(labels, first + i, icons, n_icons, px, out_status);
            px.to_rgb(out_rgba + i * out_tile_stride, true);
        }
    };
    if (threads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    return 0;
}

void orc_job_points(const osmt_batch* batch, size_t job_idx, int32_t* xy) {
    const osmt_tile_job& job = batch->jobs[job_idx];
    for (uint32_t i = 0; i < job.n_pts; ++i) {
        const Point p = batch_point(batch, job, job.pt_off + i);
        xy[2 * (size_t)i] = p.x;
        xy[2 * (size_t)i + 1] = p.y;
    }
}

void orc_composite(const double* planes, const double canvas[4], uint32_t n, uint32_t L, uint32_t W, uint32_t H,
                   uint8_t* out_rgba, int threads) {
    const size_t npx = (size_t)W * H;
    auto worker = [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; ++t) {
            for (size_t p = 0; p < npx; ++p) {
                RgbaColor d{canvas[0], canvas[1], canvas[2], canvas[3]};
                for (uint32_t l = 0; l < L; ++l) {
                    const double* s = planes + (((size_t)t * L + l) * npx + p) * 4;
                    const double a = s[3];
                    RgbaColor nw; /* tile_pixels.rs:209-219 */
                    nw.r = s[0] + (1.0 - a) * d.r;
                    nw.g = s[1] + (1.0 - a) * d.g;
                    nw.b = s[2] + (1.0 - a) * d.b;
                    nw.a = s[3] + (1.0 - a) * d.a;
                    d = nw;
                }
                auto postdivide = [&](double val) { /* tile_pixels.rs:171-175 */
                    const double mul = (d.a == 0.0) ? 0.0 : val / d.a;
                    return f64_as_u8(255.0 * mul);
                };
                uint8_t* o = out_rgba + (t * npx + p) * 4;
                o[0] = postdivide(d.r);
                o[1] = postdivide(d.g);
                o[2] = postdivide(d.b);
                o[3] = 255;
            }
        }
    };
    if (threads <= 1 || n < 2) {
        worker(0, n);
    } else {
        std::vector<std::thread> pool;
        const size_t per = (n + threads - 1) / threads;
        for (int t = 0; t < threads; ++t) {
            const size_t a = std::min<size_t>(n, t * per), b = std::min<size_t>(n, (t + 1) * per);
            if (a < b) pool.emplace_back(worker, a, b);
        }
        for (auto& t : pool) t.join();
    }
}

void orc_icon_from_rgba8(const uint8_t* rgba8, size_t n_px, double* out) {
    for (size_t i = 0; i < n_px; ++i) {
        /* RgbaColor::from_components(r,g,b,a) = from_color(Color{r,g,b}, a/255) */
        const RgbaColor c = from_color(rgba8 + 4 * i, component_to_opacity(rgba8[4 * i + 3]));
        out[4 * i] = c.r;
        out[4 * i + 1] = c.g;
        out[4 * i + 2] = c.b;
        out[4 * i + 3] = c.a;
    }
}

} /* extern "C" */
