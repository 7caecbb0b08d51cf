/*
 * osm_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * A scalar, line-by-line faithful C++ restatement of the reference's Rust hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (libosmtile.so) never links or calls it.
 *
 * PARITY PIN STATUS: see the header of osm_oracle.cpp.
 */
#ifndef OSM_ORACLE_H
#define OSM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/osmtile.h" /* display-list structs only (data layout) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_pixels orc_pixels; /* draw::tile_pixels::TilePixels */
typedef struct orc_icon {             /* draw::icon::Icon: premultiplied f64 RGBA */
    const double* rgba;
    uint32_t width, height;
} orc_icon;

/* tile.rs:88-101 / :103-106 / point.rs:11-19 */
void orc_coords_to_xy(double lat, double lon, uint8_t zoom, double* x, double* y);
void orc_coords_to_xy_tile_relative(double lat, double lon, uint8_t zoom, uint32_t tx, uint32_t ty, double* x,
                                    double* y);
void orc_project_points(const double* latlon, size_t n, uint8_t zoom, uint32_t tx, uint32_t ty, double scale,
                        int32_t* xy);
/* tile.rs:30-38 */
void orc_coords_to_max_zoom_tile(double lat, double lon, uint32_t* x, uint32_t* y);
/* point.rs:27-35 */
void orc_push_away_from(const int32_t self_xy[2], const int32_t other_xy[2], double by, int32_t out_xy[2]);

/* tile_pixels.rs */
orc_pixels* orc_pixels_new(uint32_t scale);
void orc_pixels_free(orc_pixels* p);
void orc_pixels_reset(orc_pixels* p, int has_canvas, uint8_t r, uint8_t g, uint8_t b);
void orc_set_pixel(orc_pixels* p, int32_t x, int32_t y, const double rgba[4]);
void orc_bump_generation(orc_pixels* p);
void orc_blend_unfinished_pixels(orc_pixels* p);
uint32_t orc_dimension(const orc_pixels* p);
void orc_to_rgb_triples(const orc_pixels* p, uint8_t* rgb /* [dim*dim*3] */);
/* centre-tile canvas, premultiplied f64 RGBA [dim*dim*4] (for bit-exact checks) */
void orc_read_pixels_f64(const orc_pixels* p, double* rgba);
/* pending (un-blended) alpha of the centre tile for generation `gen`; 0 where none */
void orc_read_pending_alpha(const orc_pixels* p, uint64_t gen, double* alpha);

/* fill.rs:16-47; pairs = [n_pairs][4] = (p1.x, p1.y, p2.x, p2.y) */
void orc_fill_contour(orc_pixels* p, const int32_t* pairs, size_t n_pairs, const uint8_t color[3],
                      const orc_icon* icon /* NULL = Filler::Color */, double opacity);
/* line.rs:9-61; n_dashes < 0 means dashes == None; cap = osmt_line_cap */
void orc_draw_lines(orc_pixels* p, const int32_t* pairs, size_t n_pairs, double width, const uint8_t color[3],
                    double opacity, const double* dashes, int n_dashes, int cap, int use_caps_for_dashes);
/* opacity_calculator.rs:32-43 with a fresh calculator (traveled set explicitly) */
void orc_opacity_calculate(double half_width, const double* dashes, int n_dashes, int cap, double traveled,
                           double center_distance, double start_distance, double* opacity, int* is_in_line);
/* fill.rs:51-104 walk of one edge: writes visited (x,y) pairs, returns count (cap-limited) */
size_t orc_fill_edge_walk(const int32_t p1[2], const int32_t p2[2], int32_t* out_xy, size_t cap);

/* drawer.rs:60-131 for one display-list tile (labels excluded).
 * out_rgba: [dim*dim*4], A = 255; out_f64 (optional): [dim*dim*4]. */
int orc_render_job(const osmt_batch* batch, size_t job_idx, const orc_icon* icons, size_t n_icons, uint8_t* out_rgba,
                   double* out_f64);
/* Renders jobs [first, first+count) with `threads` worker threads, one canvas
 * per thread, tiles dealt round-robin (http_server.rs:50-83,105-108). */
int orc_render_batch(const osmt_batch* batch, size_t first, size_t count, const orc_icon* icons, size_t n_icons,
                     uint8_t* out_rgba, size_t out_tile_stride, int threads);
/* The same followed by the label pass (drawer.rs:107-125): Labeler::label_entity per osmt_label in order,
 * then blend_unfinished_pixels(true).  out_status (optional): [labels->n_labels], written for the tile's labels. */
int orc_render_job_labels(const osmt_batch* batch, const osmt_label_batch* labels, size_t job_idx, const orc_icon* icons,
                          size_t n_icons, uint8_t* out_rgba, double* out_f64, uint8_t* out_status);
int orc_render_batch_labels(const osmt_batch* batch, const osmt_label_batch* labels, size_t first, size_t count,
                            const orc_icon* icons, size_t n_icons, uint8_t* out_rgba, size_t out_tile_stride, int threads,
                            uint8_t* out_status);
/* Persistent worker pool: one TilePixels per worker allocated once (http_server.rs:69-72), tiles dealt round-robin
 * (:105-108); orc_pool_render allocates nothing.  Returns 0, -1 (bad range) or -2 (pool scale != batch scale). */
typedef struct orc_pool orc_pool;
orc_pool* orc_pool_create(int threads, uint32_t scale);
void orc_pool_free(orc_pool* pool);
int orc_pool_threads(const orc_pool* pool);
int orc_pool_render(orc_pool* pool, const osmt_batch* batch, const osmt_label_batch* labels, size_t first, size_t count,
                    const orc_icon* icons, size_t n_icons, uint8_t* out_rgba, size_t out_tile_stride, uint8_t* out_status);
/* font/rasterizer.rs:27-88 + :115-147 on a fresh Rasterizer: the (x, y, total) triples save_to_figure would pass
 * to set_label_pixel, in its order.  Returns the count (may exceed cap; only cap are written). */
size_t orc_rasterizer_pixels(const double* segs, size_t n_segs, int32_t* out_xy, double* out_total, size_t cap);
/* font/rasterizer.rs:90-113 draw_quad: the draw_line calls it makes, [n][4]. */
size_t orc_flatten_quad(const double q[6], double* out_segs, size_t cap);
/* Converts the batch's points of job `job_idx` like Point::from_node: xy = [n_pts][2] */
void orc_job_points(const osmt_batch* batch, size_t job_idx, int32_t* xy);

/* tile_pixels.rs:205-223 + :164-181 over L planes (see osmt_composite) */
void orc_composite(const double* planes, const double canvas_rgba[4], uint32_t n, uint32_t L, uint32_t W, uint32_t H,
                   uint8_t* out_rgba, int threads);
/* icon.rs:32-52 + tile_pixels.rs:21-23: straight RGBA8 -> premultiplied f64 */
void orc_icon_from_rgba8(const uint8_t* rgba8, size_t n_px, double* out_rgba);

#ifdef __cplusplus
}
#endif
#endif
