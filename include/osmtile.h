/*
 * osmtile.h — C ABI of the MI355X tile rasterizer (libosmtile.so).
 *
 * Drop-in boundary for ONE hot path of dfyz/osm-renderer: the per-tile
 * project -> fill -> stroke -> blend -> RGB(A) pipeline.  The reference has no
 * FFI of its own (pure Rust crate); every entry point below cites the
 * reference interface (path:line under /root/reference) it stands in for.
 * INTEGRATION.md shows the Rust `extern "C"` binding a maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; all structs are POD with fixed layout;
 *   - the caller owns every input/output buffer for the duration of a call
 *     (borrowed, never retained — the Rust `&` / `&mut` of the reference);
 *   - every function returns an osmt_status (0 = OK, negative = error) unless
 *     stated; the message of the last error on the calling thread is
 *     available from osmt_last_error(); no exception crosses this boundary;
 *   - geometry outside the tile is not an error: pixels outside the drawable
 *     box are silently dropped (src/draw/tile_pixels.rs:107-111,191-195);
 *   - pixel coordinates are tile-relative, already multiplied by `scale`,
 *     y down; W = H = 256*scale (src/tile.rs:6, src/draw/tile_pixels.rs:57-66).
 */
#ifndef OSMTILE_H
#define OSMTILE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSMT_TILE_SIZE 256u /* src/tile.rs:6  TILE_SIZE */
#define OSMT_MAX_ZOOM 18u   /* src/tile.rs:5  MAX_ZOOM  */
#define OSMT_MAX_SCALE 4u
#define OSMT_MAX_DASHES 16u /* dash-pattern entries per op (stylesheets use <= 6) */

typedef enum osmt_status {
    OSMT_OK = 0,
    OSMT_INVALID_ARG = -1,
    OSMT_OOM = -2,
    OSMT_HIP_ERROR = -3,
    OSMT_UNSUPPORTED = -4,
    OSMT_NO_DEVICE = -5,
    OSMT_RCCL_ERROR = -6 /* the RCCL library could not be loaded, or a collective failed */
} osmt_status;

/* One draw_one_area() call of the reference == one op == one "generation"
 * (src/draw/drawer.rs:156-219, bump_generation at :218).  Casing is a STROKE
 * op with opacity 1.0 emitted at the casing-pass position (drawer.rs:186-201). */
typedef enum osmt_op_kind {
    OSMT_OP_NONE = 0,       /* draws nothing (generation bump only)            */
    OSMT_OP_FILL_COLOR = 1, /* fill_contour(.., Filler::Color, opacity)  fill.rs:16 */
    OSMT_OP_FILL_IMAGE = 2, /* fill_contour(.., Filler::Image, _)        fill.rs:36-40 */
    OSMT_OP_STROKE = 3      /* draw_lines(..)                            line.rs:9-61 */
} osmt_op_kind;

/* Option<LineCap>  (src/mapcss/styler.rs:11-16) */
typedef enum osmt_line_cap {
    OSMT_CAP_NONE = 0,
    OSMT_CAP_BUTT = 1,
    OSMT_CAP_ROUND = 2,
    OSMT_CAP_SQUARE = 3
} osmt_line_cap;

typedef enum osmt_coord_kind {
    OSMT_COORD_LATLON_F64 = 0, /* (lat, lon) degrees; projected on the GPU (tile.rs:88-106, point.rs:11-19) */
    OSMT_COORD_POINT_I32 = 1,  /* already-projected draw::point::Point {x, y} (point.rs:5-8) */
    OSMT_COORD_NODE_REF = 2    /* indices into a shared node table of (lat, lon) — the layout of the reference's
                                * geodata file, where ways hold node REFERENCES and neighbouring tiles share nodes
                                * (reader.rs:291-336, saver.rs:54-109); SURVEY.md 8(f) N2: removes the 16 B/point stream */
} osmt_coord_kind;

/* 64-byte op header. */
typedef struct osmt_op {
    uint8_t kind;                /* osmt_op_kind                                              */
    uint8_t cap;                 /* osmt_line_cap (STROKE)                       line.rs:15   */
    uint8_t use_caps_for_dashes; /* Styler::use_caps_for_dashes (STROKE)         line.rs:16   */
    uint8_t has_dashes;          /* 1 = Some(dashes), 0 = None (STROKE)          line.rs:14   */
    uint8_t color[3];            /* mapcss::color::Color {r,g,b}                 color.rs:2-6 */
    uint8_t _pad0;
    double opacity;              /* fill_opacity / opacity (1.0 when the style has none; drawer.rs:169) */
    double width;                /* STROKE: line width, already * scale          drawer.rs:191,206 */
    uint32_t n_dashes;           /* entries in the dash pattern (<= OSMT_MAX_DASHES)          */
    uint32_t dashes_off;         /* first entry in osmt_batch.dashes (already * scale; drawer.rs:171-172) */
    uint32_t n_rings;            /* Way: 1; Multipolygon: polygon_count()        point_pairs.rs:36-40 */
    uint32_t ring_off;           /* first entry in osmt_batch.rings                           */
    uint32_t image_id;           /* FILL_IMAGE: id from osmt_register_image                   */
    uint32_t _reserved[5];
} osmt_op;

/* One ring = consecutive nodes of a Way / Polygon; edges are (P[i-1], P[i]),
 * i = 1..n_pts-1 (point_pairs.rs:11-22).  A multipolygon's rings share ONE
 * running edge index (fill.rs:19). */
typedef struct osmt_ring {
    uint32_t first_pt; /* index into osmt_batch.latlon / .points */
    uint32_t n_pts;
} osmt_ring;

/* One tile == one Drawer::draw_to_pixels() call (drawer.rs:60-131), minus labels. */
typedef struct osmt_tile_job {
    uint32_t x, y;         /* tile::Tile {x, y}        tile.rs:9-13 */
    uint8_t zoom;          /* tile::Tile {zoom}                      */
    uint8_t has_canvas;    /* 0: canvas = opaque black (tile_pixels.rs:231-236) */
    uint8_t canvas_rgb[3]; /* Styler::canvas_fill_color (tile_pixels.rs:89-93)  */
    uint8_t _pad[3];
    uint32_t n_ops;  /* ops of this tile, in draw (= generation) order  */
    uint32_t op_off; /* first op in osmt_batch.ops                      */
    uint32_t n_pts;  /* the tile's points are one contiguous pool range */
    uint32_t pt_off;
} osmt_tile_job;

/* A batch of tiles sharing flat pools (display list).  All indices absolute. */
typedef struct osmt_batch {
    const osmt_tile_job* jobs;
    size_t n_jobs;
    const osmt_op* ops;
    size_t n_ops;
    const osmt_ring* rings;
    size_t n_rings;
    uint32_t coord_kind;  /* osmt_coord_kind: which of the two pools below is used */
    uint32_t scale;       /* integer scale: 1 or 2 (.. OSMT_MAX_SCALE); http_server.rs:250-258 */
    const double* latlon; /* [n_pts][2] = (lat, lon) degrees   (coords.rs:1-14) */
    const int32_t* points; /* [n_pts][2] = (x, y)               (point.rs:5-8)   */
    size_t n_pts;
    const double* dashes; /* dash pool, already * scale */
    size_t n_dashes;
    /* OSMT_COORD_NODE_REF only: point i of the pools above is nodes[node_refs[i]] */
    const double* nodes;       /* [n_nodes][2] = (lat, lon) degrees, uploaded once per scene */
    size_t n_nodes;
    const uint32_t* node_refs; /* [n_pts] */
} osmt_batch;

/* ---- label pass ---------------------------------------- */
/* One Labeler::label_entity call (labeler.rs:16-38): an optional icon blit
 * (labeler.rs:91-106) followed by an optional text (TextPlacer::place,
 * font/text_placer.rs:24-160).  The text arrives as the Rasterizer::draw_line
 * calls the reference's glyph walk makes (font/rasterizer.rs:27-88; draw_quad,
 * :90-113, flattens curves into draw_line calls on the host, so libm's hypot
 * stays where the reference calls it), in call order: the GPU replays them
 * into the exact-area accumulators and runs save_to_figure (:115-147),
 * set_label_pixel / bump_label_generation (tile_pixels.rs:131-162) and the
 * final blend_unfinished_pixels(true) (:154-158, 205-223).  40 bytes. */
typedef struct osmt_label {
    uint8_t has_icon;      /* style.icon_image found in the IconCache AND get_label_position is Some (labeler.rs:48-66) */
    uint8_t has_text;      /* place() reached save_to_figure (text_placer.rs:159); 0 = place() returned true early / no text_style */
    uint8_t text_color[3]; /* Rasterizer::color (text_placer.rs:50-54) */
    uint8_t _pad[3];
    uint32_t image_id;     /* has_icon: id from osmt_register_image */
    uint32_t seg_off;      /* first draw_line call in osmt_label_batch.segs */
    uint32_t n_segs;
    uint32_t _reserved;
    double icon_center_x, icon_center_y; /* get_label_position (labeler.rs:57-60), already scaled */
} osmt_label;

/* Labels of a whole batch; tile i owns labels [job_label_off[i], job_label_off[i+1]) in
 * draw order (drawer.rs:221-262: areas first, then nodes). */
typedef struct osmt_label_batch {
    const osmt_label* labels;
    size_t n_labels;
    const uint32_t* job_label_off; /* [n_jobs + 1] */
    const double* segs;            /* [n_segs][4] = (x0, y0, x1, y1) exactly as passed to Rasterizer::draw_line */
    size_t n_segs;
} osmt_label_batch;

typedef struct osmt_config {
    int32_t device; /* HIP device ordinal */
    uint32_t flags; /* reserved, 0 */
} osmt_config;

typedef struct osmt_ctx osmt_ctx;     /* one per GPU; analogue of Drawer + worker pool state */
typedef struct osmt_scene osmt_scene; /* a batch resident in HBM + its workspace              */

/* ---- lifecycle --------------------------------------------------------- */
/* Drawer::new (drawer.rs:33-38) + per-worker TilePixels::new (tile_pixels.rs:57-87). */
int osmt_create(const osmt_config* cfg, osmt_ctx** out_ctx);
/* Scenes hold a reference on their context: destroying a context that still has scenes only drops the handle; its
 * device buffers, streams and icon registry go with the last osmt_scene_free (no use-after-free in either order). */
void osmt_destroy(osmt_ctx* ctx);
/* anyhow::Error text (http_server.rs:127-132 prints it); thread-local, never NULL. */
const char* osmt_last_error(void);
/* Library / ABI version: (major << 16) | minor. */
uint32_t osmt_version(void);

/* ---- icons for Filler::Image ------------------------------------------- */
/* Icon::load result (icon.rs:14-58): straight-alpha RGBA8 pixels, row-major;
 * stored premultiplied exactly as RgbaColor::from_components (tile_pixels.rs:21-23). */
int osmt_register_image(osmt_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height, uint32_t* out_image_id);

/* ---- display-list validation (host only, no device needed) ----------------- */
/* The checks every upload runs first; what the reference's type system and borrow checker guarantee for
 * Drawer::draw_to_pixels' arguments (drawer.rs:60-67) has to be verified at a C boundary.  OSMT_OK, or
 * OSMT_INVALID_ARG / OSMT_UNSUPPORTED with the reason in osmt_last_error():
 *   - indices in range; the jobs' op ranges PARTITION the op pool (every op belongs to exactly one job — the per-op
 *     pre-pass runs over the whole pool) and their point ranges do not overlap (a point is projected against the
 *     tile of the job that owns it);
 *   - zoom <= OSMT_MAX_ZOOM, scale in 1..OSMT_MAX_SCALE, opacity in [0, 2^52], finite widths, known caps, dash lists
 *     non-empty (Some([]) panics in the reference, opacity_calculator.rs:109) and <= OSMT_MAX_DASHES;
 *   - coordinates the integer walks can hold: |x|, |y| <= 2^28 for OSMT_COORD_POINT_I32; finite (lat, lon) inside the
 *     Web-Mercator square (|lat| <= 85.06, |lon| <= 180) otherwise — beyond it Point::from_node saturates
 *     (point.rs:11-19) and the reference itself draws garbage. */
int osmt_validate_batch(const osmt_batch* batch);

/* ---- whole path, host buffers (Drawer::draw_to_pixels, drawer.rs:60-131) -- */
/* out_rgba: n_jobs tiles of (256*scale)^2 RGBA8 pixels (A = 255), tile i at
 * out_rgba + i*out_tile_stride_bytes, rows tightly packed
 * (TileRenderedPixels, drawer.rs:27-30; to_rgb_triples, tile_pixels.rs:164-181). */
int osmt_render_batch(osmt_ctx* ctx, const osmt_batch* batch, uint8_t* out_rgba, size_t out_tile_stride_bytes);

/* The same with the reference's own output format: packed RGB8, 3 bytes per pixel, tile i at out_rgb +
 * i*out_tile_stride_bytes (>= W*H*3) — byte for byte the memory of TileRenderedPixels.triples: Vec<(u8, u8, u8)>
 * (drawer.rs:27-30, tile_pixels.rs:46,164-181), so a Rust caller takes the buffer as it is.  The alpha byte is dropped on
 * the device: a quarter less PCIe traffic than osmt_render_batch.  `labels` may be NULL. */
int osmt_render_batch_rgb(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgb,
                          size_t out_tile_stride_bytes);

/* The same followed by the label pass (drawer.rs:107-125) when `labels` is not NULL. */
int osmt_render_batch_labels(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgba,
                             size_t out_tile_stride_bytes);

/* Pinned host memory for out_rgba: with it (or any hipHostMalloc'ed / hipHostRegister'ed buffer) and a batch of
 * >= 256 tiles, osmt_render_batch[_labels] overlaps the kernels of one 128-tile chunk with the device-to-host
 * copy of the previous one on a second stream (the per-GPU pipeline of SURVEY.md 8(e)); pageable buffers
 * take one blocking copy at the end.  Analogue of the reference's per-worker output Vec (drawer.rs:27-30). */
int osmt_host_alloc(osmt_ctx* ctx, size_t bytes, void** out_ptr);
void osmt_host_free(osmt_ctx* ctx, void* ptr);

/* ---- whole path, HBM-resident (the fast path) --------------------------- */
int osmt_scene_upload(osmt_ctx* ctx, const osmt_batch* batch, osmt_scene** out_scene);
void osmt_scene_free(osmt_scene* scene);
/* d_out_rgba: DEVICE pointer, same layout as osmt_render_batch's out_rgba.
 * stream: hipStream_t (NULL = default stream).  Asynchronous w.r.t. the host. */
int osmt_render_scene(osmt_ctx* ctx, osmt_scene* scene, void* d_out_rgba, size_t out_tile_stride_bytes, void* stream);
/* Same, but returns the un-quantised canvas: d_out_f64 = [n_jobs][H][W][4]
 * premultiplied f64 RGBA == TilePixels::pixels of the centre tile after
 * blend_unfinished_pixels(false) (tile_pixels.rs:154-158) — what the label pass
 * of the reference would continue from. */
int osmt_render_scene_f64(osmt_ctx* ctx, osmt_scene* scene, void* d_out_f64, void* stream);
/* Individual stages of osmt_render_scene (for profiling/tests): 1 = project
 * (Point::from_node), 2 = per-op extents + traveled distances, 4 = raster. */
int osmt_render_scene_stages(osmt_ctx* ctx, osmt_scene* scene, uint32_t stage_mask, void* d_out_rgba,
                             size_t out_tile_stride_bytes, void* stream);
/* Attaches the label pass of every tile of the scene (Drawer::draw_labels,
 * drawer.rs:107-125,221-262); osmt_render_scene then returns the pixels after
 * blend_unfinished_pixels(true).  NULL / n_labels == 0 detaches.  Coordinates must
 * be finite and |v| <= 2^20.  osmt_render_scene_f64 keeps returning the canvas
 * BEFORE labels. */
int osmt_scene_set_labels(osmt_ctx* ctx, osmt_scene* scene, const osmt_label_batch* labels);
/* Label statuses of the last osmt_render_scene (label_generation_statuses,
 * tile_pixels.rs:160-162): ok[i] = 1 if label i succeeded.  Synchronises the stream.  Before the first render after
 * osmt_scene_set_labels every status is 0 (nothing has been placed yet). */
int osmt_scene_read_label_status(osmt_ctx* ctx, osmt_scene* scene, uint8_t* ok);
/* Waits for the launches that read the scene (not for the device) and reports what they could not report themselves:
 * osmt_render_scene is asynchronous, so a kernel-side internal error — a pre-pass arena that does not fit, which the
 * sizing at upload rules out — would otherwise show as blank tiles.  OSMT_OK, or OSMT_HIP_ERROR with the detail in
 * osmt_last_error().  The host-buffer calls (osmt_render_batch*) make the same check before they return.  (No analogue in
 * the reference: its canvas is written synchronously by the calling thread, src/draw/drawer.rs:60-131.) */
int osmt_scene_check(osmt_ctx* ctx, osmt_scene* scene);
/* Copies the projected integer points of the scene back: xy = [n_pts][2]. */
int osmt_scene_read_points(osmt_ctx* ctx, osmt_scene* scene, int32_t* xy);

/* ---- projection only (tile.rs:88-106 + point.rs:11-19) ------------------ */
/* xy[i] = round(coords_to_xy_tile_relative(latlon[i], tile) * scale) as i32 */
int osmt_project(osmt_ctx* ctx, const double* latlon, size_t n, uint8_t zoom, uint32_t tile_x, uint32_t tile_y,
                 double scale, int32_t* xy);

/* ---- layer compositing only (tile_pixels.rs:205-223 + :164-181) ---------- */
/* planes: [n][L][H][W][4] premultiplied f64 RGBA (NextPixel.color of L
 * successive generations); canvas_rgba: premultiplied f64[4]; result per pixel:
 * dst = canvas; for l in 0..L: dst = src_l + (1 - src_l.a) * dst; then
 * un-premultiply + truncate to u8 -> out_rgba [n][H][W][4], A = 255.
 * W*H must be a multiple of 64 (true for every (256*scale)^2 tile). */
int osmt_composite(osmt_ctx* ctx, const double* planes, const double canvas_rgba[4], uint32_t n, uint32_t L,
                   uint32_t W, uint32_t H, uint8_t* out_rgba);
/* DEVICE pointers; asynchronous on `stream`. */
int osmt_composite_device(osmt_ctx* ctx, const void* d_planes, const double canvas_rgba[4], uint32_t n, uint32_t L,
                          uint32_t W, uint32_t H, void* d_out_rgba, void* stream);

/* ---- PNG encoding of a rendered tile (host side) ------- */
/* rgb_triples_to_png (src/draw/png_writer.rs:4-21), the tail of Drawer::draw_tile
 * (drawer.rs:40-58): RGB8 PNG of an RGBA8 framebuffer tile (A dropped).  The reference's
 * tests compare DECODED pixels only, so filter / compression choices are free.
 * level: zlib 0..9 (other values: zlib default).  out_capacity >= osmt_png_bound(). */
size_t osmt_png_bound(uint32_t width, uint32_t height);
int osmt_encode_png(const uint8_t* rgba, uint32_t width, uint32_t height, size_t row_stride_bytes, int level,
                    uint8_t* out_png, size_t out_capacity, size_t* out_len);

/* ---- PNG files produced on the GPU ---------------------------------------- */
/* The same file format as above (RGB8, one IDAT) written by a HIP kernel, one workgroup per tile: Paeth-filtered
 * rows, one deflate block with distance-1 run matches under a prefix code fitted to map tiles (the same "dynamic
 * Huffman" header in every file), Adler-32 and chunk CRCs computed on the device.  Decoded pixels equal the
 * framebuffer; a map tile shrinks 5-8x, so a server moves 30-50 KB per tile over PCIe instead of 256 KB and spends
 * no host time in zlib.
 * d_rgba: n framebuffers (RGBA8, rows tightly packed) tile_stride bytes apart; d_png: n slots png_stride >=
 * osmt_png_device_bound(W, H) bytes apart (multiples of 4); d_len[i] = size of file i.  W multiple of 4, <= 1024. */
size_t osmt_png_device_bound(uint32_t width, uint32_t height);
int osmt_encode_png_device(osmt_ctx* ctx, const void* d_rgba, size_t tile_stride_bytes, uint32_t n, uint32_t width, uint32_t height,
                           void* d_png, size_t png_stride_bytes, uint32_t* d_len, void* stream);
/* Drawer::draw_tile for a batch (drawer.rs:40-58): display lists (+ labels, may be NULL) in, PNG files out.
 * File i = out_png[out_off[i] .. out_off[i + 1]); out_off has n_jobs + 1 entries (filled even when out_capacity
 * is too small, so the call can be repeated with out_off[n_jobs] bytes). */
int osmt_render_batch_png(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_png, size_t out_capacity,
                          uint64_t* out_off);
/* The same call in two halves, for a caller thread that renders batch after batch (the reference answers request after
 * request, src/http_server.rs:183-201): _begin validates, uploads and queues every kernel of the batch and returns
 * without waiting; _end waits for the files, compacts them and copies them into out_png (same out_png / out_off
 * contract as the one-piece call, same errors), and frees the job whatever it returns.  With batch k + 1 begun before
 * batch k is ended, the validation and upload of k + 1 run on the host while the GPU renders and encodes k, and the
 * read-back of k runs under the kernels of k + 1: one thread reaches what several threads of one-piece calls reach.
 * The arrays of `batch` / `labels` must stay valid until _end returns (uploads are stream-ordered).  Every job holds its
 * device buffers until it is ended: ~0.9 GB per 1024 tiles of 256 x 256 (PNG slots at 12 bits per filtered byte, the
 * compacted blob, two chunks of framebuffers; kept in the context's buffer cache afterwards). */
typedef struct osmt_png_job osmt_png_job;
int osmt_render_batch_png_begin(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels /* may be NULL */, osmt_png_job** out_job);
int osmt_render_batch_png_end(osmt_png_job* job, uint8_t* out_png, size_t out_capacity, uint64_t* out_off);

/* ---- the per-request entry: one worker handle per server thread (SURVEY.md 8(b) "Threading") ----------------
 * The reference's server gives every request — ONE tile — to one of available_parallelism() worker threads, each
 * with a TilePixels of its own (src/http_server.rs:50-83,105-108,134-181: handle_connection -> draw_tile_png).
 * An osmt_worker is that per-thread handle.  osmt_worker_render is osmt_render_batch_rgb for the calling thread
 * (same arguments, same pixels, same errors, blocks until out_rgb holds the tiles), but requests of different
 * workers of one context that are in flight at the same moment are gathered into ONE launch sequence on the device:
 * a request that finds the device free starts at once, alone; requests that arrive while it renders wait and go out
 * together with the next one (at most 64 tiles per group, OSMT_WORKER_INFLIGHT = 2 groups on the device at a time).
 * Sixteen threads calling the batch entry with one tile each queue up behind each other instead (round 3: 17 k
 * tiles/s at p99 9.6 ms).  Any thread may use any worker; a worker keeps its context alive like a scene does.
 * Batches of more than 64 tiles are rendered directly. */
typedef struct osmt_worker osmt_worker;
int osmt_worker_create(osmt_ctx* ctx, osmt_worker** out_worker);
void osmt_worker_destroy(osmt_worker* worker);
int osmt_worker_render(osmt_worker* worker, const osmt_batch* batch, const osmt_label_batch* labels /* may be NULL */, uint8_t* out_rgb,
                       size_t out_tile_stride_bytes);

/* ---- one node, several GPUs ------------------------------------------------ */
/* The reference deals tiles round-robin to its worker threads, each with its own TilePixels
 * (src/http_server.rs:50-83,105-108); tiles never exchange data (neighbours' geometry is duplicated into every
 * tile's entity list, reader.rs:60-100).  Here a worker is a GPU: tile i of a batch belongs to shard i mod world. */

/* The display list of one shard: jobs rank, rank + world, ... of `batch` with their ops, rings, points and dashes
 * re-packed into pools of their own (a valid osmt_batch: the shard's op ranges partition its op pool).  The node table
 * of OSMT_COORD_NODE_REF is shared, not copied: `batch->nodes` must outlive the shard.  Host only. */
typedef struct osmt_batch_shard osmt_batch_shard;
int osmt_batch_shard_create(const osmt_batch* batch, uint32_t rank, uint32_t world, osmt_batch_shard** out_shard);
const osmt_batch* osmt_batch_shard_get(const osmt_batch_shard* shard);
void osmt_batch_shard_free(osmt_batch_shard* shard);

/* One call, n GPUs of one node: one host thread per context builds its shard and runs osmt_render_batch on it
 * (upload, kernels and the chunked read-back pipeline of each GPU overlap with the others'); tile i lands at
 * out_rgba + i * out_tile_stride_bytes exactly as with one GPU — every GPU writes its own interleaved slices of the
 * one buffer (pinned memory from osmt_host_alloc of ANY of the contexts keeps the copies asynchronous).
 * *out_tile_count (optional) = the all-reduced number of rendered tiles: over RCCL when osmt_comm_init_local has
 * joined the contexts, summed on the host otherwise; it equals batch->n_jobs on success. */
int osmt_render_batch_multi(osmt_ctx* const* ctxs, uint32_t n_ctx, const osmt_batch* batch, uint8_t* out_rgba,
                            size_t out_tile_stride_bytes, uint64_t* out_tile_count);
/* The same with what the reference's workers always do after the areas — Drawer::draw_labels per tile
 * (drawer.rs:107-125) — and with the reference's own output format: `labels` (may be NULL) is sliced per shard by
 * job_label_off (tile i's labels travel with tile i, segments re-packed); flags & OSMT_MULTI_RGB8 writes packed RGB8
 * (out_tile_stride_bytes >= W*H*3, the layout of osmt_render_batch_rgb) instead of RGBA8.  Image ids of the labels /
 * FILL_IMAGE ops are per context: register the same icons in the same order on every context. */
#define OSMT_MULTI_RGB8 1u
int osmt_render_batch_multi_ex(osmt_ctx* const* ctxs, uint32_t n_ctx, const osmt_batch* batch, const osmt_label_batch* labels,
                               uint32_t flags, uint8_t* out, size_t out_tile_stride_bytes, uint64_t* out_tile_count);

/* RCCL communicators for the tile-count reduction (the path's only collective: 8 bytes, latency-bound).  The library
 * is loaded at the first of these calls (dlopen: an already loaded RCCL — e.g. PyTorch's — is reused).
 *   one process, n GPUs:  osmt_comm_init_local(ctxs, n)                                   (ncclCommInitAll)
 *   one process per GPU:  rank 0 calls osmt_comm_unique_id and sends the 128 bytes to the other ranks by its own
 *                         means; every rank then calls osmt_comm_init_rank(ctx, id, rank, nranks)  (ncclCommInitRank) */
#define OSMT_COMM_ID_BYTES 128
int osmt_comm_unique_id(uint8_t id[OSMT_COMM_ID_BYTES]);
int osmt_comm_init_rank(osmt_ctx* ctx, const uint8_t id[OSMT_COMM_ID_BYTES], uint32_t rank, uint32_t nranks);
int osmt_comm_init_local(osmt_ctx* const* ctxs, uint32_t n_ctx);
/* ncclAllReduce(sum) of one uint64 over the communicator of `ctx`; collective: every rank calls it.  Blocks until the
 * result is on the host. */
int osmt_allreduce_tile_count(osmt_ctx* ctx, uint64_t local, uint64_t* out_global);
/* The same reduction behind the kernels of a render: enqueued on `stream` (the stream the batch was rendered on), no
 * host synchronisation, so the next batch's kernels queue up behind it.  `..._result` copies the most recent sum back
 * and waits for `stream` only.  Collective: every rank enqueues the same number of reductions. */
int osmt_allreduce_tile_count_enqueue(osmt_ctx* ctx, uint64_t local, void* stream);
int osmt_allreduce_tile_count_result(osmt_ctx* ctx, void* stream, uint64_t* out_global);
/* the same for the contexts of ONE process (a grouped call over all of them) */
int osmt_allreduce_tile_count_local(osmt_ctx* const* ctxs, uint32_t n_ctx, const uint64_t* locals, uint64_t* out_global);

/* ---- diagnostics ------------------------------------------------------------------------------ */
/* What "HBM speed" is on this device: a 16-byte-per-lane grid-stride stream over `bytes`, `iters` launches timed with
 * HIP events after one warm-up.  *out_copy_gb_per_s = copy (bytes read + bytes written); *out_read_gb_per_s (optional)
 * = the same stream read only — the ceiling of a read-dominated pass such as the layer composite.  bench.py quotes
 * roofline fractions against these next to the 8 TB/s datasheet figure. */
int osmt_hbm_copy_probe(osmt_ctx* ctx, size_t bytes, uint32_t iters, double* out_copy_gb_per_s, double* out_read_gb_per_s);
/* 1 when the process runs with OSMT_POISON_ALLOC=1: every device buffer and every pinned staging buffer the library hands
 * out — fresh or recycled from its caches — is filled with 0xA5 first.  The reference resets every pixel and pending entry
 * per tile (src/draw/tile_pixels.rs:89-105); the library recycles buffers un-zeroed, so a kernel may only read what this
 * render wrote.  The GPU tests and the fuzz run under it (tests/conftest.py); production leaves it off (one memset per
 * allocation). */
int osmt_debug_poison_enabled(void);

#ifdef __cplusplus
}
#endif
#endif /* OSMTILE_H */
