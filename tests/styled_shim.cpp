// C entry points over osm_renderer_amd/host/osmt_styled.hpp for tests/test_styled_builder.py (ctypes).  Host only:
// the header's GPU-facing neighbours (osmt_draw.hpp's Context / TilePixels) are inline and never instantiated here.
#include <cstring>

#include "../osm_renderer_amd/host/osmt_styled.hpp"

using namespace osmt;

#pragma pack(push, 1)
struct ShimStyle {
    int64_t layer;
    double z_index, opacity, fill_opacity, width, casing_width;
    uint32_t fill_image, dashes_off, n_dashes, casing_dashes_off, n_casing_dashes;
    uint8_t has_layer, is_foreground_fill;
    uint8_t has_color, color[3];
    uint8_t has_fill_color, fill_color[3];
    uint8_t has_opacity, has_fill_opacity, has_width, has_dashes, line_cap; /* cap: 0 none, 1 butt, 2 round, 3 square */
    uint8_t has_casing_color, casing_color[3];
    uint8_t has_casing_width, has_casing_dashes, casing_line_cap, has_fill_image;
};
#pragma pack(pop)

static std::optional<LineCap> cap_of(uint8_t c) {
    if (c == 1) return LineCap::Butt;
    if (c == 2) return LineCap::Round;
    if (c == 3) return LineCap::Square;
    return std::nullopt;
}

static std::vector<Style> styles_of(const ShimStyle* s, size_t n, const double* pool) {
    std::vector<Style> out(n);
    for (size_t i = 0; i < n; ++i) {
        Style& o = out[i];
        if (s[i].has_layer) o.layer = s[i].layer;
        o.z_index = s[i].z_index;
        if (s[i].has_color) o.color = Color{s[i].color[0], s[i].color[1], s[i].color[2]};
        if (s[i].has_fill_color) o.fill_color = Color{s[i].fill_color[0], s[i].fill_color[1], s[i].fill_color[2]};
        o.is_foreground_fill = s[i].is_foreground_fill != 0;
        if (s[i].has_opacity) o.opacity = s[i].opacity;
        if (s[i].has_fill_opacity) o.fill_opacity = s[i].fill_opacity;
        if (s[i].has_width) o.width = s[i].width;
        if (s[i].has_dashes) o.dashes = std::vector<double>(pool + s[i].dashes_off, pool + s[i].dashes_off + s[i].n_dashes);
        o.line_cap = cap_of(s[i].line_cap);
        if (s[i].has_casing_color) o.casing_color = Color{s[i].casing_color[0], s[i].casing_color[1], s[i].casing_color[2]};
        if (s[i].has_casing_width) o.casing_width = s[i].casing_width;
        if (s[i].has_casing_dashes)
            o.casing_dashes = std::vector<double>(pool + s[i].casing_dashes_off, pool + s[i].casing_dashes_off + s[i].n_casing_dashes);
        o.casing_line_cap = cap_of(s[i].casing_line_cap);
        if (s[i].has_fill_image) o.fill_image = s[i].fill_image;
    }
    return out;
}

static std::vector<StyledEntity> entities_of(const uint32_t* ids, const uint32_t* style_idx, size_t n, const std::vector<Style>& st) {
    std::vector<StyledEntity> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = StyledEntity{ids[i], &st[style_idx[i]]};
    return v;
}

extern "C" {
void* sb_new(void* reader, uint32_t scale) { return new SceneBuilder(*(const GeodataReader*)reader, scale); }
void sb_free(void* sb) { delete (SceneBuilder*)sb; }
void sb_add_tile(void* sb, uint8_t zoom, uint32_t x, uint32_t y, const ShimStyle* styles, size_t n_styles, const double* dash_pool,
                 const uint32_t* way_ids, const uint32_t* way_style, size_t n_ways, const uint32_t* mp_ids, const uint32_t* mp_style,
                 size_t n_mps, int has_canvas, const uint8_t* canvas, int use_caps_for_dashes) {
    const std::vector<Style> st = styles_of(styles, n_styles, dash_pool);
    std::optional<Color> cv;
    if (has_canvas) cv = Color{canvas[0], canvas[1], canvas[2]};
    ((SceneBuilder*)sb)->add_tile(Tile{zoom, x, y}, entities_of(way_ids, way_style, n_ways, st), entities_of(mp_ids, mp_style, n_mps, st), cv,
                                  use_caps_for_dashes != 0);
}
void sb_counts(void* sb, uint64_t out[5]) {
    auto* b = (SceneBuilder*)sb;
    out[0] = b->jobs().size(), out[1] = b->ops().size(), out[2] = b->rings().size(), out[3] = b->node_refs().size(), out[4] = b->dashes().size();
}
void sb_copy(void* sb, osmt_tile_job* jobs, osmt_op* ops, osmt_ring* rings, uint32_t* refs, double* dashes) {
    auto* b = (SceneBuilder*)sb;
    memcpy(jobs, b->jobs().data(), b->jobs().size() * sizeof(osmt_tile_job));
    memcpy(ops, b->ops().data(), b->ops().size() * sizeof(osmt_op));
    memcpy(rings, b->rings().data(), b->rings().size() * sizeof(osmt_ring));
    memcpy(refs, b->node_refs().data(), b->node_refs().size() * sizeof(uint32_t));
    memcpy(dashes, b->dashes().data(), b->dashes().size() * sizeof(double));
}
/* the batch view must agree with the vectors (what an integrator passes to osmt_render_batch) */
int sb_batch_consistent(void* sb) {
    auto* b = (SceneBuilder*)sb;
    const osmt_batch v = b->batch();
    return v.coord_kind == OSMT_COORD_NODE_REF && v.n_jobs == b->jobs().size() && v.n_ops == b->ops().size() && v.n_rings == b->rings().size() &&
           v.n_pts == b->node_refs().size() && v.node_refs == b->node_refs().data() && v.nodes != nullptr && v.n_dashes == b->dashes().size();
}
/* out rows: kind (0 way, 1 multipolygon, 2 node), id, style index, position (0 centre, 1 line) */
size_t sb_label_order(void* reader, const ShimStyle* styles, size_t n_styles, const uint32_t* way_ids, const uint32_t* way_style, size_t n_ways,
                      const uint32_t* mp_ids, const uint32_t* mp_style, size_t n_mps, const uint32_t* node_ids, const uint32_t* node_style,
                      size_t n_nodes, const double* dash_pool, uint32_t* out) {
    const std::vector<Style> st = styles_of(styles, n_styles, dash_pool);
    const auto order = label_order(*(const GeodataReader*)reader, entities_of(way_ids, way_style, n_ways, st), entities_of(mp_ids, mp_style, n_mps, st),
                                   entities_of(node_ids, node_style, n_nodes, st));
    for (size_t i = 0; i < order.size(); ++i) {
        out[4 * i] = (uint32_t)order[i].kind;
        out[4 * i + 1] = order[i].id;
        out[4 * i + 2] = (uint32_t)(order[i].style - st.data());
        out[4 * i + 3] = order[i].position == TextPosition::Line ? 1u : 0u;
    }
    return order.size();
}
}
