/*
 * osmt_styled.hpp — entities + styles -> osmt_batch, the C++ mirror of what sits between the
 * reference's MapCSS styler and its canvas calls:
 *
 *   reference (Rust)                                                      here (C++)
 *   mapcss::styler::Style                         styler.rs:48-71         osmt::Style (text_style excepted, see below)
 *   compare_styled_entities                       styler.rs:246-272       osmt::compare_styled_entities
 *   Styler::style_entities' final sort_by         styler.rs:163           osmt::sort_styled (stable, like sort_by)
 *   Styler::style_areas (merge, ties -> relation) styler.rs:168-203       osmt::style_areas
 *   Drawer::draw_to_pixels: reset, Fill pass with multipolygons,
 *     Casing pass, Stroke pass (ways only)        drawer.rs:60-99         osmt::SceneBuilder::add_tile
 *   Drawer::draw_areas / draw_one_area            drawer.rs:133-219       osmt::SceneBuilder::draw_one_area
 *   PointPairCollection for Way / Multipolygon    point_pairs.rs:11-41    rings of NODE INDICES (OSMT_COORD_NODE_REF)
 *   Drawer::draw_labels' iteration order          drawer.rs:221-262       osmt::label_order
 *
 * What stays with the integrator: the MapCSS parser and rule matching (Styler::style_area, property_map_to_style) —
 * they hand in the resulting Style records, one per (entity, MapCSS layer) exactly as style_entities pushes them —
 * the icon cache (a fill-image arrives as the id osmt_register_image returned, or not at all when the icon is
 * missing: drawer.rs:165-168 draws nothing then) and the text of labels (TextPlacer, font files).
 *
 * Nothing here touches the GPU: the result is a plain osmt_batch over the geodata file's node table, ready for
 * osmt_render_batch / osmt_scene_upload.  The draw order of the reference IS the op order of the batch.
 */
#ifndef OSMT_STYLED_HPP
#define OSMT_STYLED_HPP

#include <algorithm>
#include <cstdint>
#include <optional>
#include <vector>

#include "../../include/osmtile.h"
#include "osmt_draw.hpp"
#include "osmt_geodata.hpp"

namespace osmt {

enum class TextPosition { Center, Line }; /* styler.rs:18-22 */

/* mapcss::styler::Style (styler.rs:48-71).  text_style is not here: what a label draws is handed over as
 * Rasterizer::draw_line calls (osmt_label), its text never reaches this library. */
struct Style {
    std::optional<int64_t> layer;
    double z_index = 0.0;

    std::optional<Color> color;
    std::optional<Color> fill_color;
    bool is_foreground_fill = true;
    std::optional<Color> background_color;
    std::optional<double> opacity;
    std::optional<double> fill_opacity;

    std::optional<double> width;
    std::optional<std::vector<double>> dashes;
    std::optional<LineCap> line_cap;

    std::optional<Color> casing_color;
    std::optional<double> casing_width;
    std::optional<std::vector<double>> casing_dashes;
    std::optional<LineCap> casing_line_cap;

    std::optional<uint32_t> fill_image; /* id from osmt_register_image; empty: no fill-image, or its icon is not in the cache */
};

/* one element of what Styler::style_entities returns: an entity of the tile (LOCAL id in the geodata file) and one
 * of its styles (one per MapCSS layer) */
struct StyledEntity {
    uint32_t id;
    const Style* style;
};

/* styler.rs:246-272; < 0: a first, 0: equal, > 0: b first.  z_index values are never NaN (partial_cmp().unwrap()). */
inline int compare_styled_entities(uint64_t a_global_id, const Style& a, uint64_t b_global_id, const Style& b, bool for_labels) {
    const int64_t a_layer = a.layer.value_or(0), b_layer = b.layer.value_or(0);
    if (a_layer != b_layer) return a_layer < b_layer ? -1 : 1;
    if (!for_labels && a.is_foreground_fill != b.is_foreground_fill) return a.is_foreground_fill ? 1 : -1; /* false < true */
    if (a.z_index != b.z_index) return a.z_index < b.z_index ? -1 : 1;
    if (a_global_id != b_global_id) return a_global_id < b_global_id ? -1 : 1;
    return 0;
}

/* the sort_by at the end of Styler::style_entities (styler.rs:163); Rust's sort_by is stable */
template <class GlobalId>
void sort_styled(std::vector<StyledEntity>& v, GlobalId global_id_of, bool for_labels) {
    std::stable_sort(v.begin(), v.end(), [&](const StyledEntity& a, const StyledEntity& b) {
        return compare_styled_entities(global_id_of(a.id), *a.style, global_id_of(b.id), *b.style, for_labels) < 0;
    });
}

struct StyledArea { /* styler.rs:86-92 */
    bool is_multipolygon;
    uint32_t id;
    const Style* style;
};

/* Styler::style_areas (styler.rs:168-203): both lists sorted, then merged; a multipolygon goes first unless it
 * compares Greater.  `ways` / `multipolygons` arrive in the order style_entities builds them (entity order of the
 * tile, layers in rule order) and are sorted here. */
inline std::vector<StyledArea> style_areas(const GeodataReader& r, std::vector<StyledEntity> ways, std::vector<StyledEntity> multipolygons,
                                           bool for_labels) {
    sort_styled(ways, [&](uint32_t i) { return r.way_global_id(i); }, for_labels);
    sort_styled(multipolygons, [&](uint32_t i) { return r.multipolygon_global_id(i); }, for_labels);
    std::vector<StyledArea> out;
    out.reserve(ways.size() + multipolygons.size());
    size_t wi = 0, mi = 0;
    while (wi < ways.size() || mi < multipolygons.size()) {
        bool is_rel_better;
        if (mi >= multipolygons.size())
            is_rel_better = false;
        else if (wi >= ways.size())
            is_rel_better = true;
        else
            is_rel_better = compare_styled_entities(r.multipolygon_global_id(multipolygons[mi].id), *multipolygons[mi].style,
                                                    r.way_global_id(ways[wi].id), *ways[wi].style, for_labels) <= 0;
        if (is_rel_better) {
            out.push_back(StyledArea{true, multipolygons[mi].id, multipolygons[mi].style});
            ++mi;
        } else {
            out.push_back(StyledArea{false, ways[wi].id, ways[wi].style});
            ++wi;
        }
    }
    return out;
}

/* Drawer::draw_labels' order (drawer.rs:221-262): the areas as style_areas(.., for_labels = true) returns them
 * (ways label along the line, relations at the centre), then the nodes as style_entities(nodes, .., true) does. */
struct LabelTarget {
    enum Kind { Way, Multipolygon, Node } kind;
    uint32_t id;
    const Style* style;
    TextPosition position;
};
inline std::vector<LabelTarget> label_order(const GeodataReader& r, std::vector<StyledEntity> ways, std::vector<StyledEntity> multipolygons,
                                            std::vector<StyledEntity> nodes) {
    std::vector<LabelTarget> out;
    for (const StyledArea& a : style_areas(r, std::move(ways), std::move(multipolygons), true))
        out.push_back(LabelTarget{a.is_multipolygon ? LabelTarget::Multipolygon : LabelTarget::Way, a.id, a.style,
                                  a.is_multipolygon ? TextPosition::Center : TextPosition::Line});
    sort_styled(nodes, [&](uint32_t i) { return r.node_global_id(i); }, true);
    for (const StyledEntity& n : nodes) out.push_back(LabelTarget{LabelTarget::Node, n.id, n.style, TextPosition::Center});
    return out;
}

/* Tiles of ONE geodata file -> one osmt_batch (OSMT_COORD_NODE_REF over the file's node table). */
class SceneBuilder {
  public:
    SceneBuilder(const GeodataReader& reader, uint32_t scale) : r_(&reader), scale_(scale), nodes_(reader.node_table()) {}

    /* Drawer::draw_to_pixels up to "Blend after areas" (drawer.rs:60-103) for one tile: reset(canvas), then the
     * Fill pass over ways and multipolygons, the Casing pass and the Stroke pass over ways. */
    void add_tile(const Tile& tile, const std::vector<StyledEntity>& ways, const std::vector<StyledEntity>& multipolygons,
                  const std::optional<Color>& canvas_fill_color, bool use_caps_for_dashes) {
        const std::vector<StyledArea> areas = style_areas(*r_, ways, multipolygons, false);
        osmt_tile_job job{};
        job.x = tile.x;
        job.y = tile.y;
        job.zoom = tile.zoom;
        job.has_canvas = canvas_fill_color ? 1 : 0;
        if (canvas_fill_color) {
            job.canvas_rgb[0] = canvas_fill_color->r;
            job.canvas_rgb[1] = canvas_fill_color->g;
            job.canvas_rgb[2] = canvas_fill_color->b;
        }
        job.op_off = (uint32_t)ops_.size();
        job.pt_off = (uint32_t)refs_.size();
        draw_areas(areas, DrawType::Fill, true, use_caps_for_dashes);
        draw_areas(areas, DrawType::Casing, false, use_caps_for_dashes);
        draw_areas(areas, DrawType::Stroke, false, use_caps_for_dashes);
        job.n_ops = (uint32_t)ops_.size() - job.op_off;
        job.n_pts = (uint32_t)refs_.size() - job.pt_off;
        jobs_.push_back(job);
    }

    /* valid until the next add_tile / the builder's end */
    osmt_batch batch() const {
        osmt_batch b{};
        b.jobs = jobs_.data();
        b.n_jobs = jobs_.size();
        b.ops = ops_.data();
        b.n_ops = ops_.size();
        b.rings = rings_.data();
        b.n_rings = rings_.size();
        b.coord_kind = OSMT_COORD_NODE_REF;
        b.scale = scale_;
        b.points = nullptr;
        b.n_pts = refs_.size();
        b.dashes = dashes_.data();
        b.n_dashes = dashes_.size();
        b.nodes = nodes_.data();
        b.n_nodes = nodes_.size() / 2;
        b.node_refs = refs_.data();
        return b;
    }
    const std::vector<osmt_op>& ops() const { return ops_; }
    const std::vector<osmt_ring>& rings() const { return rings_; }
    const std::vector<uint32_t>& node_refs() const { return refs_; }
    const std::vector<double>& dashes() const { return dashes_; }
    const std::vector<osmt_tile_job>& jobs() const { return jobs_; }

  private:
    enum class DrawType { Fill, Stroke, Casing }; /* drawer.rs:20-25 */

    /* drawer.rs:133-154 */
    void draw_areas(const std::vector<StyledArea>& areas, DrawType draw_type, bool use_multipolygons, bool use_caps_for_dashes) {
        for (const StyledArea& a : areas)
            if (!a.is_multipolygon || use_multipolygons) draw_one_area(a, draw_type, use_caps_for_dashes);
    }

    /* to_point_pairs (point_pairs.rs:11-41) as rings of node indices: a way is one ring, a multipolygon one ring per
     * polygon; a ring of fewer than two nodes has no pair and is left out */
    std::pair<uint32_t, uint32_t> add_rings(const StyledArea& a) {
        const uint32_t first = (uint32_t)rings_.size();
        auto ring = [&](std::pair<const uint32_t*, size_t> ids) {
            if (ids.second < 2) return;
            rings_.push_back(osmt_ring{(uint32_t)refs_.size(), (uint32_t)ids.second});
            refs_.insert(refs_.end(), ids.first, ids.first + ids.second);
        };
        if (a.is_multipolygon) {
            const auto polys = r_->multipolygon_polygon_ids(a.id);
            for (size_t k = 0; k < polys.second; ++k) ring(r_->polygon_node_ids(polys.first[k]));
        } else {
            ring(r_->way_node_ids(a.id));
        }
        return {first, (uint32_t)rings_.size() - first};
    }

    void stroke(const StyledArea& a, double width, const Color& color, double opacity, const std::optional<std::vector<double>>& dashes,
                const std::optional<LineCap>& line_cap, bool use_caps_for_dashes) {
        osmt_op op{};
        op.kind = OSMT_OP_STROKE;
        op.cap = !line_cap ? OSMT_CAP_NONE
                 : *line_cap == LineCap::Butt  ? OSMT_CAP_BUTT
                 : *line_cap == LineCap::Round ? OSMT_CAP_ROUND
                                               : OSMT_CAP_SQUARE;
        op.use_caps_for_dashes = use_caps_for_dashes ? 1 : 0;
        op.color[0] = color.r, op.color[1] = color.g, op.color[2] = color.b;
        op.opacity = opacity;
        op.width = width;
        if (dashes) { /* scale_dashes (drawer.rs:170-171) */
            op.has_dashes = 1;
            op.n_dashes = (uint32_t)dashes->size();
            op.dashes_off = (uint32_t)dashes_.size();
            for (double d : *dashes) dashes_.push_back(d * (double)scale_);
        }
        const auto rr = add_rings(a);
        op.ring_off = rr.first;
        op.n_rings = rr.second;
        if (op.n_rings) ops_.push_back(op);
    }

    /* drawer.rs:156-219; an area that draws nothing only bumps the generation, which no pixel can tell */
    void draw_one_area(const StyledArea& a, DrawType draw_type, bool use_caps_for_dashes) {
        const Style& s = *a.style;
        const double scale = (double)scale_;
        switch (draw_type) {
            case DrawType::Fill: {
                const double opacity = s.fill_opacity.value_or(1.0);
                if (!s.fill_color && !s.fill_image) return;
                osmt_op op{};
                if (s.fill_color) {
                    op.kind = OSMT_OP_FILL_COLOR;
                    op.color[0] = s.fill_color->r, op.color[1] = s.fill_color->g, op.color[2] = s.fill_color->b;
                } else {
                    op.kind = OSMT_OP_FILL_IMAGE;
                    op.image_id = *s.fill_image;
                }
                op.opacity = opacity;
                const auto rr = add_rings(a);
                op.ring_off = rr.first;
                op.n_rings = rr.second;
                if (op.n_rings) ops_.push_back(op);
                return;
            }
            case DrawType::Casing:
                if (s.casing_color && s.casing_width)
                    stroke(a, *s.casing_width * scale, *s.casing_color, 1.0, s.casing_dashes, s.casing_line_cap, use_caps_for_dashes);
                return;
            case DrawType::Stroke:
                if (s.color) stroke(a, scale * s.width.value_or(1.0), *s.color, s.opacity.value_or(1.0), s.dashes, s.line_cap, use_caps_for_dashes);
                return;
        }
    }

    const GeodataReader* r_;
    uint32_t scale_;
    std::vector<double> nodes_;
    std::vector<osmt_tile_job> jobs_;
    std::vector<osmt_op> ops_;
    std::vector<osmt_ring> rings_;
    std::vector<uint32_t> refs_;
    std::vector<double> dashes_;
};

}  // namespace osmt

#endif
