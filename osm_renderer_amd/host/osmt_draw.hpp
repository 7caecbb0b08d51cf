/*
 * osmt_draw.hpp — C++ host-side mirror of the reference's draw interface, above the C ABI.
 *
 * The reference is Rust and no Rust toolchain exists in this image, so the host side that
 * would live in src/draw/{tile_pixels,fill,line,drawer}.rs is written here in C++ with the
 * SAME names, argument meaning and call order:
 *
 *   reference (Rust)                                         here (C++)
 *   tile::Tile {zoom,x,y}                 tile.rs:9-13       osmt::Tile
 *   draw::point::Point {x,y}              point.rs:5-8       osmt::Point
 *   mapcss::color::Color {r,g,b}          color.rs:2-6       osmt::Color
 *   mapcss::styler::LineCap               styler.rs:11-16    osmt::LineCap (+ std::optional)
 *   draw::fill::Filler                    fill.rs:11-14      osmt::Filler
 *   TilePixels::new/reset/bump_generation/
 *     blend_unfinished_pixels/to_rgb_triples/dimension
 *                                         tile_pixels.rs:57,89,150,154,164,183
 *                                                            osmt::TilePixels (same methods)
 *   fill_contour(points, &filler, opacity, &mut pixels)      fill.rs:16     osmt::fill_contour
 *   draw_lines(points, width, &color, opacity, &dashes,
 *              &line_cap, use_caps_for_dashes, &mut pixels)  line.rs:9-18   osmt::draw_lines
 *   TileRenderedPixels {triples, dimension}                  drawer.rs:27-30 osmt::TileRenderedPixels
 *
 * Semantics: the draw calls are RECORDED into a display list (one op per call, in order —
 * the generation order of drawer.rs:218); nothing is rasterised on the CPU.  The list is
 * executed on the GPU when pixels are requested (to_rgb_triples) or when a TileBatch of
 * several recorded tiles is flushed with one osmt_render_batch call.  Because pixels can only
 * be observed through to_rgb_triples(), deferred execution is indistinguishable from the
 * reference's eager canvas (labels — out of scope — are the one reader that would need
 * osmt_render_scene_f64 instead).
 *
 * Errors: the reference returns anyhow::Result up to the server; here every failing ABI call
 * throws osmt::Error(code, osmt_last_error()).
 */
#ifndef OSMT_DRAW_HPP
#define OSMT_DRAW_HPP

#include <cstdint>
#include <cmath>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/osmtile.h"

namespace osmt {

constexpr uint32_t TILE_SIZE = OSMT_TILE_SIZE; /* tile.rs:6 */
constexpr uint8_t MAX_ZOOM = OSMT_MAX_ZOOM;    /* tile.rs:5 */

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != OSMT_OK) throw Error(rc, osmt_last_error());
}

struct Tile {
    uint8_t zoom;
    uint32_t x, y;
};
struct Point {
    int32_t x, y;
};
struct Color {
    uint8_t r, g, b;
};
enum class LineCap { Butt, Round, Square };
using PointPairs = std::vector<std::pair<Point, Point>>; /* PointPairIter (point_pairs.rs:5) */
using RgbTriples = std::vector<std::tuple<uint8_t, uint8_t, uint8_t>>; /* tile_pixels.rs:46 */
struct TileRenderedPixels {
    RgbTriples triples;
    size_t dimension;
};

/* Drawer + worker pool state: one per GPU. */
class Context {
  public:
    explicit Context(int device = 0) {
        osmt_config cfg{device, 0};
        check(osmt_create(&cfg, &ctx_));
    }
    ~Context() { osmt_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    osmt_ctx* raw() const { return ctx_; }
    /* IconCache entry (icon_cache.rs:21-45): straight-alpha RGBA8 pixels of a decoded PNG */
    uint32_t register_image(const uint8_t* rgba8, uint32_t w, uint32_t h) {
        uint32_t id = 0;
        check(osmt_register_image(ctx_, rgba8, w, h, &id));
        return id;
    }

  private:
    osmt_ctx* ctx_ = nullptr;
};

struct Filler { /* fill.rs:11-14 */
    enum Kind { ColorFill, ImageFill } kind;
    Color color;
    uint32_t image_id;
    static Filler from_color(const Color& c) { return Filler{ColorFill, c, 0}; }
    static Filler from_image(uint32_t id) { return Filler{ImageFill, Color{0, 0, 0}, id}; }
};

class TileBatch;

/* The per-worker canvas of the reference (http_server.rs:25-28,69-72), as a recorder. */
class TilePixels {
  public:
    TilePixels(Context& ctx, size_t scale) : ctx_(&ctx), scale_(scale) {}

    /* tile_pixels.rs:89-105 */
    void reset(const std::optional<Color>& canvas_color) {
        ops_.clear();
        rings_.clear();
        points_.clear();
        dashes_.clear();
        canvas_ = canvas_color;
        pending_op_ = false;
        labels_.clear();
        label_segs_.clear();
        pending_label_ = osmt_label{};
    }
    /* drawer.rs:218: closes the current area; an area that drew nothing still counts */
    void bump_generation() {
        if (!pending_op_) {
            osmt_op nop{};
            nop.kind = OSMT_OP_NONE;
            ops_.push_back(nop);
        }
        pending_op_ = false;
    }
    /* tile_pixels.rs:154-158: a no-op for a recorder (blending happens on the GPU, in order) */
    void blend_unfinished_pixels(bool /*for_labels*/) {}
    /* tile_pixels.rs:160-162: closes the current label (one Labeler::label_entity call, labeler.rs:16-38).
     * The verdict itself is computed on the GPU (set_label_pixel's collision rule needs every earlier label);
     * the argument is accepted for source compatibility and ignored. */
    void bump_label_generation(bool /*succeeded*/) {
        labels_.push_back(pending_label_);
        pending_label_ = osmt_label{};
    }
    size_t dimension() const { return TILE_SIZE * scale_; } /* tile_pixels.rs:183-185 */
    size_t scale() const { return scale_; }

    /* tile_pixels.rs:164-181: runs the recorded list for `tile` on the GPU */
    RgbTriples to_rgb_triples(const Tile& tile = Tile{0, 0, 0}) {
        osmt_tile_job job = make_job(tile, 0, 0);
        osmt_batch b = make_batch(&job, 1, ops_, rings_, points_, dashes_);
        const size_t dim = dimension();
        std::vector<uint8_t> rgb(dim * dim * 3); /* the reference's triples, packed (std::tuple's own layout is not) */
        const uint32_t off[2] = {0u, (uint32_t)labels_.size()};
        osmt_label_batch lb{labels_.data(), labels_.size(), off, label_segs_.data(), label_segs_.size() / 4};
        check(osmt_render_batch_rgb(ctx_->raw(), &b, labels_.empty() ? nullptr : &lb, rgb.data(), rgb.size()));
        RgbTriples out(dim * dim);
        for (size_t i = 0; i < dim * dim; ++i) out[i] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        return out;
    }

  private:
    friend class TileBatch;
    friend class Rasterizer;
    friend bool draw_icon(uint32_t, double, double, TilePixels&);
    friend void fill_contour(const PointPairs&, const Filler&, double, TilePixels&);
    friend void draw_lines(const PointPairs&, double, const Color&, double, const std::optional<std::vector<double>>&,
                           const std::optional<LineCap>&, bool, TilePixels&);

    /* PointPairIter -> rings: consecutive pairs that chain (p2 == next p1) form one ring, which
     * is how point_pairs.rs:11-41 produces them; a break starts the next ring of a multipolygon. */
    std::pair<uint32_t, uint32_t> add_rings(const PointPairs& pairs) {
        const uint32_t first_ring = (uint32_t)rings_.size();
        size_t i = 0;
        while (i < pairs.size()) {
            osmt_ring r{(uint32_t)(points_.size() / 2), 0};
            points_.push_back(pairs[i].first.x);
            points_.push_back(pairs[i].first.y);
            size_t j = i;
            for (;;) {
                points_.push_back(pairs[j].second.x);
                points_.push_back(pairs[j].second.y);
                if (j + 1 < pairs.size() && pairs[j + 1].first.x == pairs[j].second.x &&
                    pairs[j + 1].first.y == pairs[j].second.y)
                    ++j;
                else
                    break;
            }
            r.n_pts = (uint32_t)(j - i + 2);
            rings_.push_back(r);
            i = j + 1;
        }
        return {first_ring, (uint32_t)rings_.size() - first_ring};
    }
    void push(const osmt_op& op) {
        ops_.push_back(op);
        pending_op_ = true;
    }
    osmt_tile_job make_job(const Tile& t, uint32_t op_off, uint32_t pt_off) const {
        osmt_tile_job j{};
        j.x = t.x;
        j.y = t.y;
        j.zoom = t.zoom;
        j.has_canvas = canvas_ ? 1 : 0;
        if (canvas_) {
            j.canvas_rgb[0] = canvas_->r;
            j.canvas_rgb[1] = canvas_->g;
            j.canvas_rgb[2] = canvas_->b;
        }
        j.n_ops = (uint32_t)ops_.size();
        j.op_off = op_off;
        j.n_pts = (uint32_t)(points_.size() / 2);
        j.pt_off = pt_off;
        return j;
    }
    osmt_batch make_batch(const osmt_tile_job* jobs, size_t n_jobs, const std::vector<osmt_op>& ops,
                          const std::vector<osmt_ring>& rings, const std::vector<int32_t>& pts,
                          const std::vector<double>& dashes) const {
        osmt_batch b{};
        b.jobs = jobs;
        b.n_jobs = n_jobs;
        b.ops = ops.data();
        b.n_ops = ops.size();
        b.rings = rings.data();
        b.n_rings = rings.size();
        b.coord_kind = OSMT_COORD_POINT_I32;
        b.scale = (uint32_t)scale_;
        b.points = pts.data();
        b.n_pts = pts.size() / 2;
        b.dashes = dashes.data();
        b.n_dashes = dashes.size();
        return b;
    }

    Context* ctx_;
    size_t scale_;
    std::optional<Color> canvas_;
    std::vector<osmt_op> ops_;
    std::vector<osmt_ring> rings_;
    std::vector<int32_t> points_;
    std::vector<double> dashes_;
    bool pending_op_ = false;
    std::vector<osmt_label> labels_;
    std::vector<double> label_segs_; /* x0, y0, x1, y1 per Rasterizer::draw_line call */
    osmt_label pending_label_{};
};

/* Labeler::draw_icon (labeler.rs:91-106): the icon of the label being built, centred at
 * get_label_position.  Always true here: collisions are resolved on the GPU, in label order. */
inline bool draw_icon(uint32_t image_id, double center_x, double center_y, TilePixels& pixels) {
    pixels.pending_label_.has_icon = 1;
    pixels.pending_label_.image_id = image_id;
    pixels.pending_label_.icon_center_x = center_x;
    pixels.pending_label_.icon_center_y = center_y;
    return true;
}

/* font/rasterizer.rs: the glyph walk of TextPlacer::place calls draw_line / draw_quad exactly as in the
 * reference; the calls are recorded (curves flattened here, with the same libm hypot) and replayed on
 * the GPU, which owns the exact-area accumulation and save_to_figure's pixel loop. */
class Rasterizer {
  public:
    explicit Rasterizer(const Color& color) : color_(color) {}
    /* :27-88 */
    void draw_line(double x0, double y0, double x1, double y1) {
        segs_.push_back(x0);
        segs_.push_back(y0);
        segs_.push_back(x1);
        segs_.push_back(y1);
    }
    /* :90-113 */
    void draw_quad(double x0, double y0, double x1, double y1, double x2, double y2) {
        auto dist_between = [](double xa, double ya, double xb, double yb) { return std::hypot(std::fabs(xa - xb), std::fabs(ya - yb)); };
        const double d01 = dist_between(x0, y0, x1, y1);
        const double d12 = dist_between(x1, y1, x2, y2);
        const double d02 = dist_between(x0, y0, x2, y2);
        if ((d01 + d12) <= 1.0001 * d02) {
            draw_line(x0, y0, x2, y2);
            return;
        }
        auto midpoint = [](double c1, double c2) { return (c1 + c2) / 2.0; };
        const double m01_x = midpoint(x0, x1), m01_y = midpoint(y0, y1);
        const double m12_x = midpoint(x1, x2), m12_y = midpoint(y1, y2);
        const double m012_x = midpoint(m01_x, m12_x), m012_y = midpoint(m01_y, m12_y);
        draw_quad(x0, y0, m01_x, m01_y, m012_x, m012_y);
        draw_quad(m012_x, m012_y, m12_x, m12_y, x2, y2);
    }
    /* :115-147: hands the text of the label being built to the canvas.  Always true (see draw_icon). */
    bool save_to_figure(TilePixels& pixels) const {
        osmt_label& l = pixels.pending_label_;
        l.has_text = 1;
        l.text_color[0] = color_.r;
        l.text_color[1] = color_.g;
        l.text_color[2] = color_.b;
        l.seg_off = (uint32_t)(pixels.label_segs_.size() / 4);
        l.n_segs = (uint32_t)(segs_.size() / 4);
        pixels.label_segs_.insert(pixels.label_segs_.end(), segs_.begin(), segs_.end());
        return true;
    }

  private:
    Color color_;
    std::vector<double> segs_;
};

/* fill.rs:16 */
inline void fill_contour(const PointPairs& points, const Filler& filler, double opacity, TilePixels& pixels) {
    osmt_op op{};
    op.kind = filler.kind == Filler::ColorFill ? OSMT_OP_FILL_COLOR : OSMT_OP_FILL_IMAGE;
    op.color[0] = filler.color.r;
    op.color[1] = filler.color.g;
    op.color[2] = filler.color.b;
    op.opacity = opacity;
    op.image_id = filler.image_id;
    auto [off, n] = pixels.add_rings(points);
    op.ring_off = off;
    op.n_rings = n;
    pixels.push(op);
}

/* line.rs:9-18.  `width` and `dashes` arrive already multiplied by scale, as in drawer.rs:191,206,171. */
inline void draw_lines(const PointPairs& points, double width, const Color& color, double opacity,
                       const std::optional<std::vector<double>>& dashes, const std::optional<LineCap>& line_cap,
                       bool use_caps_for_dashes, TilePixels& pixels) {
    osmt_op op{};
    op.kind = OSMT_OP_STROKE;
    op.cap = !line_cap ? OSMT_CAP_NONE
             : *line_cap == LineCap::Butt  ? OSMT_CAP_BUTT
             : *line_cap == LineCap::Round ? OSMT_CAP_ROUND
                                           : OSMT_CAP_SQUARE;
    op.use_caps_for_dashes = use_caps_for_dashes ? 1 : 0;
    op.color[0] = color.r;
    op.color[1] = color.g;
    op.color[2] = color.b;
    op.opacity = opacity;
    op.width = width;
    if (dashes) {
        op.has_dashes = 1;
        op.n_dashes = (uint32_t)dashes->size();
        op.dashes_off = (uint32_t)pixels.dashes_.size();
        pixels.dashes_.insert(pixels.dashes_.end(), dashes->begin(), dashes->end());
    }
    auto [off, n] = pixels.add_rings(points);
    op.ring_off = off;
    op.n_rings = n;
    pixels.push(op);
}

/* Several recorded tiles -> ONE osmt_render_batch call (the GPU fast path the single-tile
 * server loop of http_server.rs:162-171 would grow into). */
class TileBatch {
  public:
    explicit TileBatch(Context& ctx, size_t scale) : ctx_(&ctx), scale_(scale) {}
    void add(const Tile& tile, const TilePixels& px) {
        const uint32_t op_off = (uint32_t)ops_.size(), pt_off = (uint32_t)(points_.size() / 2);
        const uint32_t ring_off = (uint32_t)rings_.size(), dash_off = (uint32_t)dashes_.size();
        jobs_.push_back(px.make_job(tile, op_off, pt_off));
        for (osmt_op op : px.ops_) {
            op.ring_off += ring_off;
            if (op.n_dashes) op.dashes_off += dash_off;
            ops_.push_back(op);
        }
        for (osmt_ring r : px.rings_) {
            r.first_pt += pt_off;
            rings_.push_back(r);
        }
        points_.insert(points_.end(), px.points_.begin(), px.points_.end());
        dashes_.insert(dashes_.end(), px.dashes_.begin(), px.dashes_.end());
        const uint32_t seg_off = (uint32_t)(label_segs_.size() / 4);
        for (osmt_label l : px.labels_) {
            if (l.n_segs) l.seg_off += seg_off;
            labels_.push_back(l);
        }
        label_segs_.insert(label_segs_.end(), px.label_segs_.begin(), px.label_segs_.end());
        label_off_.push_back((uint32_t)labels_.size());
    }
    std::vector<TileRenderedPixels> render() {
        const size_t dim = TILE_SIZE * scale_;
        std::vector<uint8_t> rgb(jobs_.size() * dim * dim * 3);
        osmt_batch b{};
        b.jobs = jobs_.data();
        b.n_jobs = jobs_.size();
        b.ops = ops_.data();
        b.n_ops = ops_.size();
        b.rings = rings_.data();
        b.n_rings = rings_.size();
        b.coord_kind = OSMT_COORD_POINT_I32;
        b.scale = (uint32_t)scale_;
        b.points = points_.data();
        b.n_pts = points_.size() / 2;
        b.dashes = dashes_.data();
        b.n_dashes = dashes_.size();
        osmt_label_batch lb{labels_.data(), labels_.size(), label_off_.data(), label_segs_.data(), label_segs_.size() / 4};
        check(osmt_render_batch_rgb(ctx_->raw(), &b, labels_.empty() ? nullptr : &lb, rgb.data(), dim * dim * 3));
        std::vector<TileRenderedPixels> out(jobs_.size());
        for (size_t t = 0; t < jobs_.size(); ++t) {
            out[t].dimension = dim;
            out[t].triples.resize(dim * dim);
            const uint8_t* p = rgb.data() + t * dim * dim * 3;
            for (size_t i = 0; i < dim * dim; ++i) out[t].triples[i] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
        }
        return out;
    }

    /* Drawer::draw_tile for every added tile (drawer.rs:40-58): finished RGB8 PNG files, written by the GPU */
    std::vector<std::vector<uint8_t>> render_png() {
        const size_t dim = TILE_SIZE * scale_;
        osmt_batch b{};
        b.jobs = jobs_.data();
        b.n_jobs = jobs_.size();
        b.ops = ops_.data();
        b.n_ops = ops_.size();
        b.rings = rings_.data();
        b.n_rings = rings_.size();
        b.coord_kind = OSMT_COORD_POINT_I32;
        b.scale = (uint32_t)scale_;
        b.points = points_.data();
        b.n_pts = points_.size() / 2;
        b.dashes = dashes_.data();
        b.n_dashes = dashes_.size();
        osmt_label_batch lb{labels_.data(), labels_.size(), label_off_.data(), label_segs_.data(), label_segs_.size() / 4};
        std::vector<uint8_t> blob(jobs_.size() * osmt_png_device_bound((uint32_t)dim, (uint32_t)dim));
        std::vector<uint64_t> off(jobs_.size() + 1);
        check(osmt_render_batch_png(ctx_->raw(), &b, labels_.empty() ? nullptr : &lb, blob.data(), blob.size(), off.data()));
        std::vector<std::vector<uint8_t>> out(jobs_.size());
        for (size_t t = 0; t < jobs_.size(); ++t) out[t].assign(blob.begin() + (long)off[t], blob.begin() + (long)off[t + 1]);
        return out;
    }

  private:
    Context* ctx_;
    size_t scale_;
    std::vector<osmt_tile_job> jobs_;
    std::vector<osmt_op> ops_;
    std::vector<osmt_ring> rings_;
    std::vector<int32_t> points_;
    std::vector<double> dashes_;
    std::vector<osmt_label> labels_;
    std::vector<double> label_segs_;
    std::vector<uint32_t> label_off_{0u};
};

}  // namespace osmt
#endif
