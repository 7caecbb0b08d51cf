/*
 * osmt_api.cpp — host side of libosmtile.so: contexts, HBM-resident scenes, the C ABI
 * of include/osmtile.h.  No CPU rendering path exists here on purpose: every entry point
 * either runs the HIP kernels or fails with an error code.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <dlfcn.h>

#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <vector>

#include "../../include/osmtile.h"
#include "osmt_geom.h"
#include "osmt_internal.h"
#include "osmt_png_table.h" /* PNG_LMAX, PNG_BLOCK_HDR_BITS: the slot bound */

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

/* No C++ exception may cross the C boundary (a std::bad_alloc from a host-side table, say): every entry point
 * runs its body through this. */
template <class F>
int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(OSMT_OOM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(OSMT_HIP_ERROR, "internal error: %s", e.what());
    } catch (...) {
        return fail(OSMT_HIP_ERROR, "internal error");
    }
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(_e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "%s failed: %s", #expr, \
                        hipGetErrorString(_e));                                                  \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

/* error reporting for the other translation units of the library */
int osmt_fail_public(int code, const char* msg) { return fail(code, "%s", msg); }

struct osmt_ctx {
    int device = 0;
    std::mutex mu; /* guards the image registry */
    /* Device buffers of finished calls, kept for the next one: a per-request server loop uploads, renders and frees
     * a scene per call, and hipMalloc / hipFree of a few hundred MB cost more than the kernels. */
    struct cached_buf {
        void* p;
        size_t bytes;
        bool used;
        uint64_t tick; /* when it was last handed out: the idle buffers given back to the driver are the least recently used */
    };
    std::mutex cache_mu;
    std::vector<cached_buf> cache;
    uint64_t cache_tick = 0;
    /* idle non-blocking streams: every host-buffer call runs on its own stream, so calls from the reference's N
     * worker threads (http_server.rs:50-83) overlap on the GPU instead of queueing behind the NULL stream */
    std::vector<hipStream_t> idle_streams;
    hipStream_t poison_stream = nullptr; /* OSMT_POISON_ALLOC: the fills run on a stream of the highest priority (dev_poison) */
    std::vector<cached_buf> host_cache; /* pinned staging buffers (one packed H2D copy per small call) */
    std::vector<osmt_image_desc> images;
    std::vector<double> image_pool_host; /* premultiplied f64 RGBA */
    /* Device copy of the registry.  A registration followed by a render makes a NEW pair of buffers; the old pair
     * moves to `image_graveyard` and lives until the context goes, so a kernel launched by another worker thread
     * with the previous snapshot never reads freed memory (registrations are rare: once per icon at start-up). */
    osmt_image_desc* d_images = nullptr;
    double4* d_image_pool = nullptr;
    uint32_t d_n_images = 0;
    std::vector<void*> image_graveyard;
    bool images_dirty = false;
    /* one reference for the handle returned by osmt_create + one per live scene: osmt_destroy on a context that still
     * has scenes only drops the handle's reference, the last osmt_scene_free tears the context down */
    std::atomic<int> refs{1};
    /* RCCL communicator of the tile-count reduction (osmt_comm_init_*): ncclComm_t, rank and size */
    void* comm = nullptr;
    uint32_t comm_rank = 0, comm_size = 0;
    unsigned long long* d_count = nullptr; /* device words of the reductions: [0], [1] blocking call, [2], [3] enqueued one */
    /* One pinned, device-visible word per live scene: a pre-pass kernel whose arena reservation does not fit stores a code
     * there (osmt_prepass_args.err) and the host reads it behind the synchronisation it does anyway — an internal error
     * surfaces as OSMT_HIP_ERROR instead of blank tiles, and costs no copy and no extra wait. */
    uint32_t* err_page = nullptr;
    std::vector<uint16_t> err_free;
    /* What the pre-pass arenas of recent big uploads needed per fill op / per virtual stroke segment, by scale: the next
     * big upload of a host-buffer call takes its arenas from these densities (+ 25 %) instead of asking the device with a
     * counting run of the pre-pass and a round trip; see scene_size_arenas (guarded by cache_mu). */
    struct arena_density {
        double groups_per_fill = 0.0, recs_per_vseg = 0.0;
        uint32_t uploads = 0;
    } density[OSMT_MAX_SCALE + 1];
    /* The gathering point of the per-request entry (osmt_worker_render): requests of concurrent worker threads wait here
     * and are rendered together, see the "coalescing worker" section. */
    std::mutex co_mu;
    std::condition_variable co_cv;
    std::deque<struct coalesce_req*> co_queue;
    int co_in_flight = 0;
};

struct osmt_scene {
    osmt_ctx* ctx = nullptr;
    uint32_t n_jobs = 0, n_ops = 0, n_rings = 0, n_pts = 0, n_dashes = 0, n_strokes = 0;
    uint32_t scale = 1, coord_kind = 0, n_blk = 0;
    char* d_base = nullptr; /* one allocation, carved below */
    char* d_front = nullptr; /* big uploads: the arrays the host provides, in an allocation of their own (copied by a helper thread while the tables are built) */
    size_t bytes = 0;
    osmt_tile_job* d_jobs = nullptr;
    osmt_op* d_ops = nullptr;
    osmt_ring* d_rings = nullptr;
    double* d_latlon = nullptr;     /* per point, or the node table (OSMT_COORD_NODE_REF) */
    uint32_t* d_node_refs = nullptr;
    int32_t* d_pts = nullptr;
    double* d_dashes = nullptr;
    uint32_t* d_pt_job = nullptr;
    uint32_t* d_op_aux = nullptr;
    osmt_opinfo* d_info = nullptr;
    osmt_stroke_aux* d_aux = nullptr;
    osmt_dash_seg* d_dseg = nullptr;
    uint32_t* d_submask = nullptr;
    uint32_t* d_op_blk = nullptr;
    uint32_t* d_op_vseg = nullptr; /* op -> its first virtual segment (stroke ops with segments) */
    osmt_blk_bbox* d_blk = nullptr;
    uint32_t* d_op_job = nullptr;
    osmt_vseg* d_vseg = nullptr; /* per virtual segment: end points, traveled, length, slot offset, op (k_opinfo -> k_prebin) */
    unsigned long long* d_cursors = nullptr; /* 4 words; the per-sub-tile list counts follow (one memset) */
    uint32_t* d_cnt = nullptr;
    uint2* d_hdr = nullptr;
    osmt_ent* d_ent = nullptr;
    unsigned long long ent_cap = 0;
    uint32_t n_fill_ops = 0;
    uint32_t n_vsegs = 0;
    /* the two arenas of the pre-pass (fill coverage words, stroke records + keys): one allocation, sized at upload */
    char* d_arena = nullptr;
    uint32_t* d_fmask = nullptr;
    osmt_srec* d_srec = nullptr;
    uint2* d_skey = nullptr;
    unsigned long long fmask_cap = 0, srec_cap = 0; /* 64-byte groups / records */
    /* host-side tables whose upload may still be in flight on the call's stream */
    std::vector<uint32_t> h_pt_job, h_op_aux, h_op_blk, h_op_vseg, h_op_job, h_lab_wide;
    std::vector<osmt_label_band> h_lab_bands;
    std::vector<osmt_labelinfo> h_lab_info;
    hipStream_t own_stream = nullptr; /* internal per-call scene: everything about it happens on this stream */
    /* public scenes: one event per stream the scene was rendered on, recorded behind the last launch that reads it.
     * osmt_scene_free / set_labels / read_* wait for THESE events only — never for the device — so worker threads
     * with their own scenes and streams do not stall each other (http_server.rs:50-83: independent workers). */
    std::mutex use_mu;
    std::vector<std::pair<hipStream_t, hipEvent_t>> last_use;
    void* h_stage = nullptr;          /* pinned staging of a packed upload, returned to the pool when the scene goes */
    uint32_t* h_err = nullptr;        /* the scene's word of osmt_ctx::err_page (NULL: none left, errors stay unreported) */
    uint32_t max_job_ops = 0;         /* most ops of any of the scene's tiles: at most OSMT_FOLD_MAX_OPS = no list kernel */
    bool arena_guess = false;         /* the arenas were sized from osmt_ctx::density, not by the device: an overflow is a miss, not a bug */
    /* label pass (osmt_scene_set_labels): its own allocation */
    uint32_t n_labels = 0, n_label_segs = 0;
    char* d_lab_base = nullptr;
    osmt_labelinfo* d_lab = nullptr;
    uint32_t* d_job_label_off = nullptr;
    double* d_lab_segs = nullptr;
    double* d_lab_a = nullptr;
    double* d_lab_s_wide = nullptr;
    uint32_t* d_lab_wide = nullptr;
    osmt_label_band* d_lab_bands = nullptr;
    unsigned long long* d_lab_bits = nullptr;
    uint32_t n_lab_bands = 0;
    uint32_t n_lab_wide = 0;
    uint32_t* d_lab_bitmap = nullptr;
    uint8_t* d_lab_ok = nullptr;
    uint32_t* d_lab_err = nullptr;
    osmt_tile_label* d_tile_labels = nullptr;
    uint32_t* d_tile_label_cnt = nullptr;
};

namespace {

constexpr double OSMT_MAX_ABS_LAT = 85.06; /* Web-Mercator limit 85.0511..., with a little slack */
/* Idle buffers beyond this are returned to the driver, least recently used first.  hipFree waits for the device: a process
 * that has rendered a few big batches (a 10 000-tile call leaves 5 GB idle) and then runs pipelined jobs must not free —
 * and malloc again — a buffer on every call (round 4: 8 GB, oldest ALLOCATION first: the PNG begin / end leg of bench.py ran
 * at 3.6 ms per batch behind the other legs and at 2.1 ms on its own).  288 GB of HBM: 32 GB, OSMT_CACHE_GB overrides. */
size_t cache_keep_bytes() {
    static const size_t v = [] {
        const char* e = getenv("OSMT_CACHE_GB");
        const long gb = e ? atol(e) : 32;
        return (size_t)std::min<long>(std::max<long>(gb, 0), 256) << 30;
    }();
    return v;
}

/* OSMT_POISON_ALLOC=1 (tests/conftest.py, tools/fuzz_parity.py): every device buffer handed out — fresh from the driver or
 * recycled from the cache — is filled with 0xA5 first, and every pinned staging buffer too.  The reference resets every
 * pixel and every pending entry per tile (tile_pixels.rs:89-105); this library recycles its buffers un-zeroed, so whatever a
 * kernel reads it must have been written by this render.  A fresh process sees zero pages from hipMalloc and hides a missing
 * write (round 4's empty-tile list headers); under poison it reads 0xA5A5A5A5 and fails at once. */
bool poison_alloc() {
    static const bool v = [] {
        const char* e = getenv("OSMT_POISON_ALLOC");
        return e && *e && *e != '0';
    }();
    return v;
}

hipError_t stream_acquire(osmt_ctx* ctx, hipStream_t* out);
void stream_release(osmt_ctx* ctx, hipStream_t st);

hipError_t dev_poison(osmt_ctx* ctx, void* p, size_t bytes) {
    /* The fill is a kernel: on a pooled stream it can share a hardware queue with a caller's long queue of renders and wait
     * for all of them (tests/test_gpu_fullsize_and_errors.py::test_worker_threads_..., once in four runs).  A stream of the
     * highest priority has a queue of its own and its workgroups are dispatched first. */
    hipStream_t st = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        if (!ctx->poison_stream) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi); /* hi = the numerically lowest = greatest priority */
            if (hipStreamCreateWithPriority(&ctx->poison_stream, hipStreamNonBlocking, hi) != hipSuccess) {
                (void)hipGetLastError();
                ctx->poison_stream = nullptr;
            }
        }
        st = ctx->poison_stream;
    }
    const bool pooled = st == nullptr;
    hipError_t e = pooled ? stream_acquire(ctx, &st) : hipSuccess;
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p, 0xA5, bytes, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (pooled) stream_release(ctx, st);
    return e;
}

hipError_t dev_alloc_raw(osmt_ctx* ctx, void** out, size_t bytes, size_t* got);

hipError_t dev_alloc(osmt_ctx* ctx, void** out, size_t bytes) {
    size_t got = 0;
    hipError_t e = dev_alloc_raw(ctx, out, bytes, &got);
    if (e == hipSuccess && poison_alloc()) e = dev_poison(ctx, *out, got);
    return e;
}

hipError_t dev_alloc_raw(osmt_ctx* ctx, void** out, size_t bytes, size_t* got) {
    bytes = align_up(bytes ? bytes : 1, (size_t)2 << 20);
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        osmt_ctx::cached_buf* best = nullptr;
        for (auto& c : ctx->cache)
            if (!c.used && c.bytes >= bytes && c.bytes <= 2 * bytes && (!best || c.bytes < best->bytes)) best = &c;
        if (best) {
            best->used = true;
            best->tick = ++ctx->cache_tick;
            *out = best->p;
            *got = best->bytes;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) { /* give the idle buffers back and retry once */
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        for (size_t i = 0; i < ctx->cache.size();) {
            if (!ctx->cache[i].used) {
                (void)hipFree(ctx->cache[i].p);
                ctx->cache.erase(ctx->cache.begin() + (long)i);
            } else {
                ++i;
            }
        }
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    ctx->cache.push_back({*out, bytes, true, ++ctx->cache_tick});
    *got = bytes;
    return hipSuccess;
}

hipError_t stream_acquire(osmt_ctx* ctx, hipStream_t* out) {
    {
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        if (!ctx->idle_streams.empty()) {
            *out = ctx->idle_streams.back();
            ctx->idle_streams.pop_back();
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void stream_release(osmt_ctx* ctx, hipStream_t st) {
    if (!st) return;
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    ctx->idle_streams.push_back(st);
}

constexpr size_t STAGE_MAX_BYTES = (size_t)4 << 20; /* calls with more input than this copy array by array */
constexpr uint32_t ERR_SLOTS = 1024;

uint32_t* err_slot_acquire(osmt_ctx* ctx) {
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    if (!ctx->err_page) {
        void* p = nullptr;
        if (hipHostMalloc(&p, ERR_SLOTS * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        memset(p, 0, ERR_SLOTS * sizeof(uint32_t));
        ctx->err_page = (uint32_t*)p;
        ctx->err_free.reserve(ERR_SLOTS);
        for (uint32_t i = ERR_SLOTS; i-- > 0;) ctx->err_free.push_back((uint16_t)i);
    }
    if (ctx->err_free.empty()) return nullptr;
    uint32_t* w = ctx->err_page + ctx->err_free.back();
    ctx->err_free.pop_back();
    *w = 0u;
    return w;
}

void err_slot_release(osmt_ctx* ctx, uint32_t* w) {
    if (!w) return;
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    ctx->err_free.push_back((uint16_t)(w - ctx->err_page));
}

void* stage_acquire(osmt_ctx* ctx, size_t bytes) {
    bytes = align_up(bytes ? bytes : 1, (size_t)64 << 10);
    {
        /* best fit, and never a buffer more than four times the request: first fit handed the megabyte-sized output
         * staging of a gathered group to the next 20 KB upload, and the next group then had to hipHostMalloc (hundreds of
         * microseconds) a new one — every time */
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        osmt_ctx::cached_buf* best = nullptr;
        for (auto& c : ctx->host_cache)
            if (!c.used && c.bytes >= bytes && c.bytes <= 4 * bytes && (!best || c.bytes < best->bytes)) best = &c;
        if (best) {
            best->used = true;
            if (poison_alloc()) memset(best->p, 0xA5, best->bytes);
            return best->p;
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr; /* the caller falls back to per-array copies */
    }
    if (poison_alloc()) memset(p, 0xA5, bytes);
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    ctx->host_cache.push_back({p, bytes, true, 0});
    return p;
}

void stage_release(osmt_ctx* ctx, void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    for (auto& c : ctx->host_cache)
        if (c.p == p) c.used = false;
}

void dev_free(osmt_ctx* ctx, void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    size_t idle = 0;
    for (auto& c : ctx->cache) {
        if (c.p == p) c.used = false;
        if (!c.used) idle += c.bytes;
    }
    while (idle > cache_keep_bytes()) {
        size_t lru = ctx->cache.size();
        for (size_t i = 0; i < ctx->cache.size(); ++i)
            if (!ctx->cache[i].used && (lru == ctx->cache.size() || ctx->cache[i].tick < ctx->cache[lru].tick)) lru = i;
        if (lru == ctx->cache.size()) break;
        idle -= ctx->cache[lru].bytes;
        (void)hipFree(ctx->cache[lru].p);
        ctx->cache.erase(ctx->cache.begin() + (long)lru);
    }
}

struct image_snapshot {
    const osmt_image_desc* desc = nullptr;
    const double4* pool = nullptr;
    uint32_t n = 0;
};

/* Brings the device copy of the icon registry up to date and returns a consistent (descriptors, pool, count) triple
 * that stays valid for the life of the context. */
int sync_images(osmt_ctx* ctx, image_snapshot* snap) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->images_dirty) {
        osmt_image_desc* nd = nullptr;
        double4* np = nullptr;
        if (!ctx->images.empty()) {
            HIP_TRY(hipMalloc((void**)&nd, ctx->images.size() * sizeof(osmt_image_desc)));
            hipError_t e = hipMalloc((void**)&np, ctx->image_pool_host.size() * sizeof(double));
            if (e == hipSuccess) e = hipMemcpy(nd, ctx->images.data(), ctx->images.size() * sizeof(osmt_image_desc), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(np, ctx->image_pool_host.data(), ctx->image_pool_host.size() * sizeof(double), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(nd);
                if (np) (void)hipFree(np);
                return fail(e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "icon registry upload failed: %s", hipGetErrorString(e));
            }
        }
        if (ctx->d_images) ctx->image_graveyard.push_back(ctx->d_images);
        if (ctx->d_image_pool) ctx->image_graveyard.push_back(ctx->d_image_pool);
        ctx->d_images = nd;
        ctx->d_image_pool = np;
        ctx->d_n_images = (uint32_t)ctx->images.size();
        ctx->images_dirty = false;
    }
    if (snap) {
        snap->desc = ctx->d_images;
        snap->pool = ctx->d_image_pool;
        snap->n = ctx->d_n_images;
    }
    return OSMT_OK;
}

void comm_destroy(osmt_ctx* ctx);

void ctx_teardown(osmt_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    comm_destroy(ctx);
    if (ctx->d_images) (void)hipFree(ctx->d_images);
    if (ctx->d_image_pool) (void)hipFree(ctx->d_image_pool);
    for (void* p : ctx->image_graveyard) (void)hipFree(p);
    for (auto& c : ctx->cache) (void)hipFree(c.p);
    for (hipStream_t st : ctx->idle_streams) (void)hipStreamDestroy(st);
    if (ctx->poison_stream) (void)hipStreamDestroy(ctx->poison_stream);
    for (auto& c : ctx->host_cache) (void)hipHostFree(c.p);
    if (ctx->err_page) (void)hipHostFree(ctx->err_page);
    delete ctx;
}

void ctx_release(osmt_ctx* ctx) {
    if (ctx->refs.fetch_sub(1) == 1) ctx_teardown(ctx);
}

/* The wait at the end of a small request: the work behind it is ~100 us of kernels, and a thread that sleeps in the driver
 * until the completion interrupt wakes it adds a noticeable share of that on top.  Poll the stream for a bounded time
 * first (a request of the per-tile server loop is latency, not throughput), then fall back to the blocking wait.
 * OSMT_SPIN_SYNC=0 turns the polling off. */
hipError_t stream_sync_small(hipStream_t st) {
    static const bool spin = [] {
        const char* v = getenv("OSMT_SPIN_SYNC");
        return !(v && v[0] == '0');
    }();
    /* (Several groups in flight = several threads polling, and hipStreamQuery takes the runtime's lock every time: with six
     * groups in flight a two-tile group takes 520 us instead of 200.  Waiting on a word of pinned memory that the stream
     * writes behind the copy — hipStreamWriteValue32 — removes that contention (6 in flight: 20 k -> 42 k tiles/s) but the
     * write itself is slow: 2 in flight 33 k against 45 k.  Two groups in flight and the query it is.) */
    if (spin) {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
        for (;;) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            if (std::chrono::steady_clock::now() >= t_end) break;
        }
        (void)hipGetLastError(); /* hipErrorNotReady is sticky in hipGetLastError */
    }
    return hipStreamSynchronize(st);
}

/* Device -> host copy of finished data on a private pooled stream: a NULL-stream hipMemcpy would order itself against
 * every blocking stream of the process. */
hipError_t copy_back(osmt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return hipSuccess;
    hipStream_t st = nullptr;
    hipError_t e = stream_acquire(ctx, &st);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    stream_release(ctx, st);
    return e;
}

/* Host-buffer entry points: an internal error of the label coverage kernels (window overflow) must not return OSMT_OK
 * with wrong pixels.  Called after the call's work was enqueued on `st`; synchronises it. */
/* set by prepass_error_check when the overflow it reports belongs to a scene with guessed arenas: the caller renders again
 * with exact sizing instead of failing */
thread_local bool g_arena_guess_missed = false;

/* the pre-pass kernels' word: valid once the stream(s) the scene was rendered on have been synchronised */
int prepass_error_check(osmt_scene* sc) {
    if (!sc->h_err) return OSMT_OK;
    const uint32_t code = *(volatile uint32_t*)sc->h_err;
    if (!code) return OSMT_OK;
    *(volatile uint32_t*)sc->h_err = 0u;
    if (sc->arena_guess) {
        g_arena_guess_missed = true;
        std::lock_guard<std::mutex> lk(sc->ctx->cache_mu);
        sc->ctx->density[sc->scale].uploads = 0; /* the next big upload measures again */
    }
    return fail(OSMT_HIP_ERROR, "pre-pass arena overflow (internal error %u: %s arena)", code,
                code == OSMT_PREPASS_ERR_FILL_ARENA ? "fill" : code == OSMT_PREPASS_ERR_STROKE_ARENA ? "stroke" : "list");
}

int label_error_check(osmt_scene* sc, hipStream_t st) {
    {
        const int rc = prepass_error_check(sc); /* every caller has synchronised the scene's stream(s) by now */
        if (rc != OSMT_OK) return rc;
    }
    if (!sc->n_labels) return OSMT_OK;
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, sc->d_lab_err, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (err) return fail(OSMT_HIP_ERROR, "label coverage window overflow (internal error %u)", err);
    return OSMT_OK;
}

/* The scene object itself (device buffers are returned by the caller first). */
void scene_delete(osmt_scene* s) {
    for (auto& u : s->last_use) (void)hipEventDestroy(u.second);
    osmt_ctx* ctx = s->ctx;
    if (ctx) err_slot_release(ctx, s->h_err);
    delete s;
    if (ctx) ctx_release(ctx);
}

/* Records "the scene was last read here" behind the launches just issued on `st` (public scenes only). */
hipError_t scene_mark_use(osmt_scene* sc, hipStream_t st) {
    if (sc->own_stream) return hipSuccess; /* per-call scenes live and die on their own stream */
    std::lock_guard<std::mutex> lk(sc->use_mu);
    for (auto& u : sc->last_use)
        if (u.first == st) return hipEventRecord(u.second, st);
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    sc->last_use.emplace_back(st, ev);
    return hipEventRecord(ev, st);
}

/* Blocks the calling thread until every launch that reads or writes the scene has finished — and nothing else. */
hipError_t scene_wait_idle(osmt_scene* sc) {
    if (sc->own_stream) return hipStreamSynchronize(sc->own_stream);
    std::lock_guard<std::mutex> lk(sc->use_mu);
    hipError_t first = hipSuccess;
    for (auto& u : sc->last_use) {
        const hipError_t e = hipEventSynchronize(u.second);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    return first;
}

/* the coordinate scan of validate_batch: independent of everything else, O(n_pts) — big uploads run it on a helper thread
 * beside the host-table building and the copies (validate_coords_range is what the thread calls on its slice) */
int validate_coords_range(const osmt_batch* b, size_t lo, size_t hi) {
    if (b->coord_kind == OSMT_COORD_POINT_I32) {
        for (size_t i = 2 * lo; i < 2 * hi; ++i)
            if (b->points[i] > OSMT_COORD_LIMIT || b->points[i] < -OSMT_COORD_LIMIT)
                return fail(OSMT_UNSUPPORTED, "point %zu: |coordinate| > 2^28", i / 2);
    } else {
        /* The closed forms and the Bresenham state of the kernels need |pixel coordinate| <= 2^28 (osmt_geom.h); with
         * zoom <= 18 and scale <= 4 that holds for every (lat, lon) of the Web-Mercator square.  Outside it (or for a
         * NaN / infinity) Point::from_node saturates (point.rs:11-19) and the integer walks would overflow. */
        const double* ll = b->coord_kind == OSMT_COORD_NODE_REF ? b->nodes : b->latlon;
        for (size_t i = lo; i < hi; ++i) {
            const double lat = ll[2 * i], lon = ll[2 * i + 1];
            if (!(std::fabs(lat) <= OSMT_MAX_ABS_LAT) || !(std::fabs(lon) <= 180.0))
                return fail(OSMT_UNSUPPORTED, "point %zu: (lat, lon) = (%g, %g) outside the Web-Mercator square (|lat| <= %g, |lon| <= 180)", i, lat,
                            lon, OSMT_MAX_ABS_LAT);
        }
    }
    return OSMT_OK;
}
size_t coord_count(const osmt_batch* b) { return b->coord_kind == OSMT_COORD_NODE_REF ? b->n_nodes : b->n_pts; }

/* ---- what Rust's types guarantee has to be checked at a C boundary, in three pieces so that callers can validate in
 * parallel (osmt_render_batch_multi: one thread per GPU; osmt_worker_render: every requester its own request) ----------
 * validate_batch_global: what needs the whole batch — the header, and that the jobs' op ranges PARTITION the op pool
 * (every op of the pool is pre-processed on the device, k_opinfo runs over [0, n_ops): every op must be reachable through
 * exactly one job) and their point ranges do not overlap (a point is projected against the tile of the job that owns it)
 * — in O(n_jobs log n_jobs): two sorts of the ranges.  validate_job: everything about ONE job — its ops, rings, dashes,
 * node references and (with_coords) the coordinates of its points; jobs are independent, any number of threads may run
 * it on disjoint sets of jobs.  validate_nodes: a slice of the shared node table of OSMT_COORD_NODE_REF.  Points no job
 * owns are not looked at: no kernel reads them either (k_project skips them). */
int validate_batch_global(const osmt_batch* b) {
    if (!b) return fail(OSMT_INVALID_ARG, "batch is NULL");
    if (b->scale < 1 || b->scale > OSMT_MAX_SCALE) return fail(OSMT_INVALID_ARG, "scale %u not in 1..%u", b->scale, OSMT_MAX_SCALE);
    if (b->coord_kind != OSMT_COORD_LATLON_F64 && b->coord_kind != OSMT_COORD_POINT_I32 && b->coord_kind != OSMT_COORD_NODE_REF)
        return fail(OSMT_INVALID_ARG, "unknown coord_kind %u", b->coord_kind);
    if (b->n_jobs >= 0x7FFFFFFFull / 64 || b->n_ops >= 0xFFFFFFFFull || b->n_rings >= 0xFFFFFFFFull ||
        b->n_pts >= 0xFFFFFFFFull || b->n_dashes >= 0xFFFFFFFFull)
        return fail(OSMT_INVALID_ARG, "batch too large for 32-bit indices");
    if ((b->n_jobs && !b->jobs) || (b->n_ops && !b->ops) || (b->n_rings && !b->rings) || (b->n_dashes && !b->dashes))
        return fail(OSMT_INVALID_ARG, "NULL pool with non-zero count");
    if (b->n_pts) {
        if (b->coord_kind == OSMT_COORD_LATLON_F64 && !b->latlon) return fail(OSMT_INVALID_ARG, "latlon pool is NULL");
        if (b->coord_kind == OSMT_COORD_POINT_I32 && !b->points) return fail(OSMT_INVALID_ARG, "points pool is NULL");
        if (b->coord_kind == OSMT_COORD_NODE_REF) {
            if (!b->node_refs || !b->nodes) return fail(OSMT_INVALID_ARG, "node table / node_refs is NULL");
            if (b->n_nodes >= 0xFFFFFFFFull) return fail(OSMT_INVALID_ARG, "node table too large for 32-bit indices");
        }
    }
    struct range {
        uint64_t lo, hi;
        uint32_t job;
    };
    std::vector<range> ops, pts;
    ops.reserve(b->n_jobs);
    pts.reserve(b->n_jobs);
    for (size_t j = 0; j < b->n_jobs; ++j) {
        const osmt_tile_job& job = b->jobs[j];
        if (job.zoom > OSMT_MAX_ZOOM) return fail(OSMT_INVALID_ARG, "job %zu: zoom %u > MAX_ZOOM (src/tile.rs:5)", j, job.zoom);
        if ((size_t)job.op_off + job.n_ops > b->n_ops) return fail(OSMT_INVALID_ARG, "job %zu: op range out of bounds", j);
        if ((size_t)job.pt_off + job.n_pts > b->n_pts) return fail(OSMT_INVALID_ARG, "job %zu: point range out of bounds", j);
        if (job.n_ops) ops.push_back({job.op_off, (uint64_t)job.op_off + job.n_ops, (uint32_t)j});
        if (job.n_pts) pts.push_back({job.pt_off, (uint64_t)job.pt_off + job.n_pts, (uint32_t)j});
    }
    auto by_lo = [](const range& x, const range& y) { return x.lo < y.lo; };
    std::sort(ops.begin(), ops.end(), by_lo);
    uint64_t next = 0;
    for (const range& r : ops) {
        if (r.lo < next)
            return fail(OSMT_INVALID_ARG, "job %u: op %llu also belongs to another job (op ranges must not overlap)", r.job, (unsigned long long)r.lo);
        if (r.lo > next)
            return fail(OSMT_INVALID_ARG, "op %llu is not covered by any job (the jobs' op ranges must partition the op pool)", (unsigned long long)next);
        next = r.hi;
    }
    if (next != b->n_ops)
        return fail(OSMT_INVALID_ARG, "op %llu is not covered by any job (the jobs' op ranges must partition the op pool)", (unsigned long long)next);
    std::sort(pts.begin(), pts.end(), by_lo);
    for (size_t i = 1; i < pts.size(); ++i)
        if (pts[i].lo < pts[i - 1].hi) return fail(OSMT_INVALID_ARG, "the point ranges of two jobs overlap at point %llu", (unsigned long long)pts[i].lo);
    return OSMT_OK;
}

int validate_job(const osmt_batch* b, size_t j, bool with_coords = true) {
    const osmt_tile_job& job = b->jobs[j];
    for (uint32_t k = 0; k < job.n_ops; ++k) {
        const osmt_op& op = b->ops[job.op_off + k];
        if (op.kind > OSMT_OP_STROKE) return fail(OSMT_INVALID_ARG, "job %zu op %u: unknown kind %u", j, k, op.kind);
        if (op.kind == OSMT_OP_NONE) continue;
        if ((size_t)op.ring_off + op.n_rings > b->n_rings) return fail(OSMT_INVALID_ARG, "job %zu op %u: ring range out of bounds", j, k);
        for (uint32_t r = 0; r < op.n_rings; ++r) {
            const osmt_ring& ring = b->rings[op.ring_off + r];
            if (ring.first_pt < job.pt_off || (size_t)ring.first_pt + ring.n_pts > (size_t)job.pt_off + job.n_pts)
                return fail(OSMT_INVALID_ARG, "job %zu op %u ring %u: points outside the job's pool range", j, k, r);
        }
        if (!(op.opacity >= 0.0) || !(op.opacity <= 4503599627370496.0))
            return fail(OSMT_INVALID_ARG, "job %zu op %u: opacity must be in [0, 2^52]", j, k);
        if (op.kind == OSMT_OP_STROKE) {
            if (!std::isfinite(op.width)) return fail(OSMT_INVALID_ARG, "job %zu op %u: width not finite", j, k);
            if (std::fabs(op.width) > 65536.0) return fail(OSMT_UNSUPPORTED, "job %zu op %u: |width| > 65536 px", j, k);
            if (op.cap > OSMT_CAP_SQUARE) return fail(OSMT_INVALID_ARG, "job %zu op %u: unknown cap", j, k);
            if (op.has_dashes) {
                if (op.n_dashes == 0) return fail(OSMT_INVALID_ARG, "job %zu op %u: empty dash list", j, k);
                if (op.n_dashes > OSMT_MAX_DASHES) return fail(OSMT_UNSUPPORTED, "job %zu op %u: more than %u dashes", j, k, OSMT_MAX_DASHES);
                if ((size_t)op.dashes_off + op.n_dashes > b->n_dashes) return fail(OSMT_INVALID_ARG, "job %zu op %u: dash range out of bounds", j, k);
            }
        }
    }
    if (b->coord_kind == OSMT_COORD_NODE_REF) {
        for (size_t i = job.pt_off; i < (size_t)job.pt_off + job.n_pts; ++i)
            if (b->node_refs[i] >= b->n_nodes) return fail(OSMT_INVALID_ARG, "point %zu: node reference %u out of range", i, b->node_refs[i]);
        return OSMT_OK; /* the node table itself: validate_nodes */
    }
    return with_coords ? validate_coords_range(b, job.pt_off, (size_t)job.pt_off + job.n_pts) : OSMT_OK;
}

int validate_nodes(const osmt_batch* b, size_t lo, size_t hi) {
    return b->coord_kind == OSMT_COORD_NODE_REF ? validate_coords_range(b, lo, hi) : OSMT_OK;
}

/* the whole batch on the calling thread; with_coords = false: the caller scans the coordinates itself (big uploads do it
 * on helper threads, validate_coords_range over the whole pool) */
int validate_batch(const osmt_batch* b, bool with_coords = true) {
    int rc = validate_batch_global(b);
    for (size_t j = 0; rc == OSMT_OK && j < b->n_jobs; ++j) rc = validate_job(b, j, with_coords);
    if (rc == OSMT_OK && with_coords) rc = validate_nodes(b, 0, b->n_nodes);
    return rc;
}

/* Arguments of stage 2 (k_opinfo -> k_fill_rows -> k_stroke_bin); sizing: only the arena cursors are produced. */
osmt_prepass_args prepass_args(const osmt_scene* sc, bool sizing) {
    osmt_prepass_args a;
    memset(&a, 0, sizeof a);
    a.jobs = sc->d_jobs;
    a.n_jobs = sc->n_jobs;
    a.ops = sc->d_ops;
    a.n_ops = sc->n_ops;
    a.rings = sc->d_rings;
    a.pts = reinterpret_cast<const int2*>(sc->d_pts);
    a.dashes = sc->d_dashes;
    a.op_aux = sc->d_op_aux;
    a.op_job = sc->d_op_job;
    a.op_blk = sc->d_op_blk;
    a.op_vseg = sc->d_op_vseg;
    a.vseg = sc->d_vseg;
    a.n_vsegs = sc->n_vsegs;
    a.max_job_ops = sc->max_job_ops;
    a.fold_max_ops = sc->n_jobs <= OSMT_FOLD_MAX_JOBS ? OSMT_FOLD_MAX_OPS : 0u;
    a.scale = sc->scale;
    a.sub_rows = OSMT_TILE_SIZE * sc->scale / OSMT_SUB_H;
    a.info = sc->d_info;
    a.aux = sc->d_aux;
    a.dseg = sc->d_dseg;
    a.blk = sc->d_blk;
    a.submask = sc->d_submask;
    a.cursors = sc->d_cursors;
    a.cnt = sc->d_cnt;
    a.hdr = sc->d_hdr;
    a.ent = sc->d_ent;
    a.ent_cap = sizing ? 0ull : sc->ent_cap;
    a.fmask = sc->d_fmask;
    a.srec = sc->d_srec;
    a.skey = sc->d_skey;
    a.fmask_cap = sizing ? 0ull : sc->fmask_cap;
    a.srec_cap = sizing ? 0ull : sc->srec_cap;
    a.err = sizing ? nullptr : sc->h_err; /* hipHostMallocMapped memory: one address on both sides (unified addressing) */
    return a;
}

/* stages: 1 = project, 2 = per-op pre-pass, 4 = raster (with the label kernels first when the scene has labels),
 * 8 = label kernels only (cover + resolve), 16 = with 4: the label kernels already ran for this scene.
 * [first_job, first_job + n_range) = tiles the raster stage renders into d_out (n_range 0: all). */
int render_impl(osmt_ctx* ctx, osmt_scene* sc, uint32_t stages, void* d_out, size_t stride, bool f64, void* stream,
                uint32_t first_job = 0, uint32_t n_range = 0) {
    if (!ctx || !sc || sc->ctx != ctx) return fail(OSMT_INVALID_ARG, "bad ctx/scene");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const uint32_t W = OSMT_TILE_SIZE * sc->scale;
    if ((stages & 4u) && !d_out) return fail(OSMT_INVALID_ARG, "output pointer is NULL");
    const bool rgb8 = (stages & 32u) != 0u; /* stage bit 32 (internal): k_raster writes packed RGB8 */
    if ((stages & 4u) && !f64 && !rgb8 && stride < (size_t)W * W * 4) return fail(OSMT_INVALID_ARG, "out_tile_stride_bytes < W*H*4");
    if ((stages & 4u) && rgb8 && (f64 || stride < (size_t)W * W * 3 || (stride & 3u) || ((uintptr_t)d_out & 3u)))
        return fail(OSMT_INVALID_ARG, "RGB8 output: tile stride below W*H*3 or not a multiple of 4");
    bool zeroed = false;
    if ((stages & 1u) && sc->coord_kind != OSMT_COORD_POINT_I32) {
        /* when the pre-pass follows in this call, the projection kernel clears its cursors and list counts on the way */
        const osmt_prepass_args pa = prepass_args(sc, false);
        zeroed = (stages & 2u) != 0u && sc->n_pts != 0u;
        HIP_TRY(osmt_launch_project(sc->d_jobs, sc->d_pt_job, sc->d_latlon, sc->coord_kind == OSMT_COORD_NODE_REF ? sc->d_node_refs : nullptr,
                                    sc->n_pts, (double)sc->scale, sc->d_pts, st, zeroed ? reinterpret_cast<uint32_t*>(pa.cursors) : nullptr,
                                    zeroed ? osmt_prepass_zero_words(pa) : 0));
    }
    if (stages & 2u) HIP_TRY(osmt_launch_prepass(prepass_args(sc, false), st, zeroed));
    const bool want_labels = sc->n_labels && !f64;
    if (want_labels && ((stages & 8u) || ((stages & 4u) && !(stages & 16u)))) {
        /* the label pass does not read the area canvas: coverage + collisions first, then
         * k_raster blends the survivors right before to_rgb_triples */
        int rc = sync_images(ctx, nullptr);
        if (rc != OSMT_OK) return rc;
        osmt_label_launch ll;
        memset(&ll, 0, sizeof ll);
        ll.info = sc->d_lab;
        ll.n_labels = sc->n_labels;
        ll.n_jobs = sc->n_jobs;
        ll.scale = sc->scale;
        ll.n_wide = sc->n_lab_wide;
        ll.bands = sc->d_lab_bands;
        ll.n_bands = sc->n_lab_bands;
        ll.job_label_off = sc->d_job_label_off;
        ll.segs = sc->d_lab_segs;
        ll.wide = sc->d_lab_wide;
        ll.plane_a = sc->d_lab_a;
        ll.cell_bits = sc->d_lab_bits;
        ll.plane_s_wide = sc->d_lab_s_wide;
        ll.bitmap = sc->d_lab_bitmap;
        ll.ok = sc->d_lab_ok;
        ll.err = sc->d_lab_err;
        ll.tile_labels = sc->d_tile_labels;
        ll.tile_label_cnt = sc->d_tile_label_cnt;
        HIP_TRY(osmt_launch_labels(ll, st));
    }
    if (stages & 4u) {
        image_snapshot img;
        int rc = sync_images(ctx, &img);
        if (rc != OSMT_OK) return rc;
        if (first_job > sc->n_jobs || n_range > sc->n_jobs - first_job) return fail(OSMT_INVALID_ARG, "tile range out of bounds");
        const uint32_t n_render = n_range ? n_range : sc->n_jobs - first_job;
        osmt_raster_args a;
        memset(&a, 0, sizeof a);
        a.jobs = sc->d_jobs + first_job;
        a.n_jobs = n_render;
        a.scale = sc->scale;
        a.aux = sc->d_aux;
        a.dseg = sc->d_dseg;
        {
            const uint32_t Wt = OSMT_TILE_SIZE * sc->scale;
            a.hdr = sc->d_hdr + (size_t)first_job * (Wt / OSMT_SUB_W) * (Wt / OSMT_SUB_H);
        }
        a.ent = sc->d_ent;
        a.fmask = sc->d_fmask;
        a.srec = sc->d_srec;
        a.skey = sc->d_skey;
        a.info = sc->d_info;
        a.submask = sc->d_submask;
        a.fold_max_ops = sc->n_jobs <= OSMT_FOLD_MAX_JOBS ? OSMT_FOLD_MAX_OPS : 0u; /* the same rule as the pre-pass (prepass_args) */
        a.images = img.desc;
        a.image_pool = img.pool;
        a.n_images = img.n;
        a.out = d_out;
        a.out_tile_stride = stride;
        a.out_rgb8 = rgb8 ? 1u : 0u;
        if (want_labels) {
            a.labels.info = sc->d_lab;
            a.labels.n_labels = sc->n_labels;
            a.labels.job_label_off = sc->d_job_label_off + first_job; /* values stay absolute label indices */
            a.labels.tile_labels = sc->d_tile_labels;
            a.labels.tile_label_cnt = sc->d_tile_label_cnt + first_job;
            a.labels.plane = sc->d_lab_a;
        }
        HIP_TRY(osmt_launch_raster(a, f64, st));
    }
    HIP_TRY(scene_mark_use(sc, st));
    return OSMT_OK;
}

}  // namespace

extern "C" {

uint32_t osmt_version(void) { return (1u << 16) | 0u; }

const char* osmt_last_error(void) { return g_last_error.c_str(); }

static int osmt_create_body(const osmt_config* cfg, osmt_ctx** out_ctx) {
    if (!out_ctx) return fail(OSMT_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(OSMT_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    const int dev = cfg ? cfg->device : 0;
    if (dev < 0 || dev >= n) return fail(OSMT_INVALID_ARG, "device %d out of range (0..%d)", dev, n - 1);
    HIP_TRY(hipSetDevice(dev));
    osmt_ctx* c = new (std::nothrow) osmt_ctx();
    if (!c) return fail(OSMT_OOM, "out of host memory");
    c->device = dev;
    *out_ctx = c;
    return OSMT_OK;
}

int osmt_create(const osmt_config* cfg, osmt_ctx** out_ctx) {
    return guarded([&] { return osmt_create_body(cfg, out_ctx); });
}

void osmt_destroy(osmt_ctx* ctx) {
    if (!ctx) return;
    ctx_release(ctx); /* scenes still alive keep the context (and the device buffers they use) until they are freed */
}

static int osmt_register_image_body(osmt_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height, uint32_t* out_id) {
    if (!ctx || !rgba8 || !out_id) return fail(OSMT_INVALID_ARG, "NULL argument");
    if (width == 0 || height == 0) return fail(OSMT_INVALID_ARG, "empty image");
    std::lock_guard<std::mutex> lk(ctx->mu);
    osmt_image_desc d;
    d.offset = ctx->image_pool_host.size() / 4;
    d.width = width;
    d.height = height;
    const size_t n = (size_t)width * height;
    ctx->image_pool_host.reserve(ctx->image_pool_host.size() + 4 * n);
    for (size_t i = 0; i < n; ++i) {
        /* RgbaColor::from_components (tile_pixels.rs:21-23): from_color(Color{r,g,b}, a/255) */
        const double a = (double)rgba8[4 * i + 3] / 255.0;
        ctx->image_pool_host.push_back(a * ((double)rgba8[4 * i + 0] / 255.0));
        ctx->image_pool_host.push_back(a * ((double)rgba8[4 * i + 1] / 255.0));
        ctx->image_pool_host.push_back(a * ((double)rgba8[4 * i + 2] / 255.0));
        ctx->image_pool_host.push_back(a);
    }
    *out_id = (uint32_t)ctx->images.size();
    ctx->images.push_back(d);
    ctx->images_dirty = true;
    return OSMT_OK;
}

int osmt_register_image(osmt_ctx* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height, uint32_t* out_id) {
    return guarded([&] { return osmt_register_image_body(ctx, rgba8, width, height, out_id); });
}

/* The pre-pass writes into two arenas whose sizes depend on the PROJECTED geometry (how many sub-tiles every op
 * reaches).  Small scenes take the worst case — every fill covers the whole tile, every stroke segment reaches every
 * sub-tile — without asking the device (a single-tile request must not pay a round trip); larger ones run the
 * projection and the sizing half of k_opinfo once, read the two totals back and allocate exactly.  Either way the
 * reservation logic of k_opinfo is the same code at render time, so the arenas cannot overflow. */
static int scene_size_arenas(osmt_ctx* ctx, osmt_scene* s, size_t n_fills, bool allow_guess) {
    const size_t W = (size_t)OSMT_TILE_SIZE * s->scale;
    const size_t nsub = (W / OSMT_SUB_W) * (W / OSMT_SUB_H);
    unsigned long long groups = (unsigned long long)n_fills * nsub, recs = (unsigned long long)s->n_vsegs * nsub;
    const unsigned long long worst_bytes = groups * 64ull + recs * (sizeof(osmt_srec) + 8ull) + ((unsigned long long)n_fills + s->n_strokes) * nsub * sizeof(osmt_ent);
    /* (taking the worst case up to 2 GB instead — 1.8 GB for 1024 config-2 tiles — was tried to save this sizing run,
     * 0.14 ms of a 0.84 ms upload: no gain on one thread, and four worker threads' arenas then outgrow the buffer cache) */
    bool guessed = false;
    /* a guess needs the scene's error word: without one (more than ERR_SLOTS live scenes, or no pinned page) an overflow
     * could not be seen and the call would return OSMT_OK with geometry missing — such a scene is sized exactly */
    if (worst_bytes > ((unsigned long long)32 << 20) && allow_guess && s->h_err != nullptr) {
        /* A host-buffer call that follows others of its kind: the arenas from the densities the recent exact runs measured,
         * a quarter on top.  The kernels reserve with the same code as ever; if the guess is too small for this batch they
         * draw nothing for the ops that do not fit and say so in the scene's error word, and the call renders again with
         * exact sizing (g_arena_guess_missed).  Every 64th upload measures again, so the densities follow the workload. */
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        osmt_ctx::arena_density& d = ctx->density[s->scale];
        if (d.uploads != 0 && (d.uploads & 63u) != 0 && d.groups_per_fill > 0.0 && d.recs_per_vseg > 0.0) {
            const unsigned long long g = (unsigned long long)(d.groups_per_fill * 1.25 * (double)n_fills) + 4096ull;
            const unsigned long long r = (unsigned long long)(d.recs_per_vseg * 1.25 * (double)s->n_vsegs) + 4096ull;
            groups = std::min(groups, g);
            recs = std::min(recs, r);
            guessed = true;
            ++d.uploads;
        }
    }
    s->arena_guess = guessed;
    if (worst_bytes > ((unsigned long long)32 << 20) && !guessed) {
        hipStream_t st = s->own_stream;
        if (s->coord_kind != OSMT_COORD_POINT_I32)
            HIP_TRY(osmt_launch_project(s->d_jobs, s->d_pt_job, s->d_latlon, s->coord_kind == OSMT_COORD_NODE_REF ? s->d_node_refs : nullptr,
                                        s->n_pts, (double)s->scale, s->d_pts, st));
        HIP_TRY(osmt_launch_prepass(prepass_args(s, true), st));
        unsigned long long totals[2] = {0ull, 0ull};
        HIP_TRY(hipMemcpyAsync(totals, s->d_cursors, sizeof totals, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        groups = totals[0];
        recs = totals[1];
        std::lock_guard<std::mutex> lk(ctx->cache_mu);
        osmt_ctx::arena_density& d = ctx->density[s->scale];
        if (n_fills) d.groups_per_fill = std::max(d.groups_per_fill * 0.98, (double)groups / (double)n_fills);
        if (s->n_vsegs) d.recs_per_vseg = std::max(d.recs_per_vseg * 0.98, (double)recs / (double)s->n_vsegs);
        if (d.groups_per_fill <= 0.0) d.groups_per_fill = 1e-9; /* a workload without fills (or strokes) still counts as measured */
        if (d.recs_per_vseg <= 0.0) d.recs_per_vseg = 1e-9;
        ++d.uploads;
    }
    if (groups >= 0xFFFFFFFFull || recs >= 0xFFFFFFFFull)
        return fail(OSMT_UNSUPPORTED, "scene needs %llu fill groups / %llu stroke records (> 2^32): split the batch", groups, recs);
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    /* list entries: one per (op, sub-tile the op draws into) — at most one per fill group / stroke slot, and never more
     * than every op in every sub-tile */
    unsigned long long ents = std::min<unsigned long long>(groups + recs, ((unsigned long long)n_fills + s->n_strokes) * nsub);
    if (ents >= 0xFFFFFFFFull) return fail(OSMT_UNSUPPORTED, "scene needs %llu list entries (> 2^32): split the batch", ents);
    const size_t o_f = carve((size_t)(groups + 1) * 64);
    const size_t o_r = carve((size_t)(recs + 1) * sizeof(osmt_srec));
    const size_t o_k = carve((size_t)(recs + 1) * 8);
    const size_t o_e = carve((size_t)(ents + 1) * sizeof(osmt_ent));
    hipError_t e = dev_alloc(ctx, (void**)&s->d_arena, off + 256);
    if (e != hipSuccess) {
        s->d_arena = nullptr;
        return fail(e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "hipMalloc(%zu) for the pre-pass arenas failed: %s", off, hipGetErrorString(e));
    }
    s->d_fmask = (uint32_t*)(s->d_arena + o_f);
    s->d_srec = (osmt_srec*)(s->d_arena + o_r);
    s->d_skey = (uint2*)(s->d_arena + o_k);
    s->d_ent = (osmt_ent*)(s->d_arena + o_e);
    s->ent_cap = ents + 1;
    s->fmask_cap = groups + 1; /* never 0: 0 means "sizing pass" to the kernels */
    s->srec_cap = recs + 1;
    return OSMT_OK;
}

/* st == nullptr: blocking copies (the public osmt_scene_upload); otherwise stream-ordered on `st`, the caller
 * synchronises the stream before the batch's host arrays go away */
/* trusted: the batch was built by the library itself from a batch it has already validated (the shards of osmt_render_batch_multi) */
static int scene_upload_impl(osmt_ctx* ctx, const osmt_batch* b, osmt_scene** out_scene, hipStream_t st, bool trusted = false,
                             bool allow_guess = false) {
    if (!ctx || !out_scene) return fail(OSMT_INVALID_ARG, "NULL argument");
    *out_scene = nullptr;
    /* Big uploads: the O(n_pts) coordinate scan runs on helper threads while this one builds the index tables and feeds
     * the copies (they do not depend on it); the verdict is collected before the first kernel can be launched. */
    const bool big = b && coord_count(b) >= ((size_t)1 << 16);
    /* Where the host's own arrays go inside the scene's memory: known from the batch's sizes alone */
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    const bool ll = b && b->coord_kind == OSMT_COORD_LATLON_F64;
    const bool nr = b && b->coord_kind == OSMT_COORD_NODE_REF;
    size_t o_jobs = 0, o_ops = 0, o_rings = 0, o_latlon = 0, o_refs = 0, o_pts = 0, o_dashes = 0, o_ptjob = 0, o_opaux = 0, o_opblk = 0, o_opvseg = 0,
           o_opjob = 0, front_bytes = 0;
    if (b) {
        o_jobs = carve(b->n_jobs * sizeof(osmt_tile_job));
        o_ops = carve(b->n_ops * sizeof(osmt_op));
        o_rings = carve(b->n_rings * sizeof(osmt_ring));
        o_latlon = carve(ll ? b->n_pts * 16 : nr ? b->n_nodes * 16 : 0);
        o_refs = carve(nr ? b->n_pts * 4 : 0);
        o_pts = carve(b->n_pts * 8);
        o_dashes = carve((b->n_dashes + 1) * 8);
        o_ptjob = carve(b->n_pts * 4);
        o_opaux = carve(b->n_ops * 4);
        o_opblk = carve(b->n_ops * 4);
        o_opvseg = carve(b->n_ops * 4);
        o_opjob = carve(b->n_ops * 4);
        front_bytes = off; /* everything the host provides sits in [0, front_bytes) */
    }
    /* SPLIT upload (round 5; big stream-ordered uploads): the caller's arrays — 27 MB for 1024 config-2 tiles, 0.44 ms of
     * pageable copies — go to an allocation of their own and are copied by a helper thread WHILE this thread validates the jobs
     * and builds the index tables (0.3 + 0.19 ms): the host half of a one-piece call was validation -> tables -> copies in a
     * row in front of the GPU half.  Nothing on the device looks at the data before the verdicts are in (the first launch comes
     * after the join below). */
    const bool split = big && st != nullptr && front_bytes > STAGE_MAX_BYTES;
    int rc = OSMT_OK;
    if (!trusted) rc = split ? validate_batch_global(b) : validate_batch(b, !big);
    if (rc != OSMT_OK) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    struct coord_check {
        std::vector<std::thread> th;
        std::vector<int> rc;
        std::vector<std::string> msg;
        int join() { /* first failing slice wins: the message names the lowest point index */
            for (auto& t : th)
                if (t.joinable()) t.join();
            for (size_t i = 0; i < rc.size(); ++i)
                if (rc[i] != OSMT_OK) return fail(rc[i], "%s", msg[i].c_str());
            return OSMT_OK;
        }
        void wait() {
            for (auto& t : th)
                if (t.joinable()) t.join();
        }
        ~coord_check() { wait(); }
    } cc;
    const unsigned n_coord_thr = (big && !trusted) ? (coord_count(b) >= ((size_t)1 << 19) ? 2u : 1u) : 0u;
    cc.rc.assign(n_coord_thr + (split ? 2u : 0u), OSMT_OK); /* sized before any helper runs: they write their own slots */
    cc.msg.resize(n_coord_thr + (split ? 2u : 0u));
    if (big && !trusted) {
        const size_t n = coord_count(b);
        const unsigned n_thr = n_coord_thr;
        for (unsigned t = 0; t < n_thr; ++t) {
            const size_t lo = n * t / n_thr, hi = n * (t + 1) / n_thr;
            try {
                cc.th.emplace_back([&cc, b, t, lo, hi] {
                    cc.rc[t] = guarded([&] { return validate_coords_range(b, lo, hi); });
                    if (cc.rc[t] != OSMT_OK) cc.msg[t] = osmt_last_error();
                });
            } catch (...) { /* no thread to be had: scan here */
                cc.rc[t] = validate_coords_range(b, lo, hi);
                if (cc.rc[t] != OSMT_OK) cc.msg[t] = osmt_last_error();
            }
        }
    }

    char* d_front = nullptr;
    /* Owns d_front until the scene's free path does: whatever way this function is left before that — a refused batch,
     * a failed allocation, an exception out of the table building (bad_alloc, caught by guarded()) — the helpers are joined, THEIR
     * COPIES ON THE STREAM ARE WAITED FOR (joining a helper only means its hipMemcpyAsync calls were issued: up to tens of MB may
     * still be on their way out of the caller's arrays and into a buffer the cache would hand to the next upload at once) and the
     * buffer goes back to the cache.  Declared behind `cc`: destroyed before it. */
    struct front_guard {
        osmt_ctx* ctx;
        coord_check* cc;
        hipStream_t st;
        char* p = nullptr;
        ~front_guard() {
            if (!p) return;
            cc->wait();
            (void)hipStreamSynchronize(st);
            dev_free(ctx, p);
        }
    } fg{ctx, &cc, st ? st : nullptr};
    if (split) {
        hipError_t fe = dev_alloc(ctx, (void**)&d_front, front_bytes + 256);
        if (fe != hipSuccess) {
            cc.wait();
            return fail(fe == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "hipMalloc(%zu) failed: %s", front_bytes, hipGetErrorString(fe));
        }
        fg.p = d_front;
        /* two helpers: the op side (jobs, ops, rings, dashes) and the coordinate side — 16 + 11 MB for 1024 config-2 tiles */
        auto copy_user_arrays = [&cc, ctx, b, st, d_front, ll, nr, o_jobs, o_ops, o_rings, o_latlon, o_refs, o_pts, o_dashes](unsigned slot, bool coords) {
            cc.rc[slot] = guarded([&] {
                HIP_TRY(hipSetDevice(ctx->device));
                auto put = [&](size_t o, const void* src, size_t bytes) -> hipError_t {
                    return bytes ? hipMemcpyAsync(d_front + o, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
                };
                if (!coords) {
                    HIP_TRY(put(o_jobs, b->jobs, b->n_jobs * sizeof(osmt_tile_job)));
                    HIP_TRY(put(o_ops, b->ops, b->n_ops * sizeof(osmt_op)));
                    HIP_TRY(put(o_rings, b->rings, b->n_rings * sizeof(osmt_ring)));
                    HIP_TRY(put(o_dashes, b->dashes, b->n_dashes * 8));
                } else {
                    if (ll) HIP_TRY(put(o_latlon, b->latlon, b->n_pts * 16));
                    if (nr) HIP_TRY(put(o_latlon, b->nodes, b->n_nodes * 16));
                    if (nr) HIP_TRY(put(o_refs, b->node_refs, b->n_pts * 4));
                    if (!ll && !nr) HIP_TRY(put(o_pts, b->points, b->n_pts * 8));
                }
                return (int)OSMT_OK;
            });
            if (cc.rc[slot] != OSMT_OK) cc.msg[slot] = osmt_last_error();
        };
        for (unsigned h = 0; h < 2u; ++h) {
            try {
                cc.th.emplace_back(copy_user_arrays, n_coord_thr + h, h == 1u);
            } catch (...) { /* no thread to be had: copy here */
                copy_user_arrays(n_coord_thr + h, h == 1u);
            }
        }
        if (!trusted) { /* the per-job half of validate_batch, beside the copies (coordinates: the helpers above) */
            for (size_t j = 0; rc == OSMT_OK && j < b->n_jobs; ++j) rc = validate_job(b, j, false);
            if (rc != OSMT_OK) {
                const std::string msg = osmt_last_error();
                return fail(rc, "%s", msg.c_str()); /* (front_guard: join, wait for the copies, free) */
            }
        }
    }

    /* OSMT_TRACE_UPLOAD=1 (diagnostic): where the host time of this call goes, one line on stderr */
    static const bool trace_upload = getenv("OSMT_TRACE_UPLOAD") != nullptr;
    using tclock = std::chrono::steady_clock;
    tclock::time_point tp[6];
    auto mark = [&](int i) {
        if (trace_upload) tp[i] = tclock::now();
    };
    auto report = [&] {
        if (!trace_upload) return;
        auto us = [&](int a, int b2) { return std::chrono::duration<double, std::micro>(tp[b2] - tp[a]).count(); };
        fprintf(stderr, "osmt upload: tables %.0f us, alloc %.0f us, copies %.0f us, coordinate check join %.0f us, arenas %.0f us\n",
                us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5));
    };
    mark(0);
    osmt_scene* s = new (std::nothrow) osmt_scene();
    if (!s) return fail(OSMT_OOM, "out of host memory");
    s->own_stream = st;
    /* host-side index tables: point -> job (for projection), op -> stroke slot */
    std::vector<uint32_t>& pt_job = s->h_pt_job;
    std::vector<uint32_t>& op_aux = s->h_op_aux;
    std::vector<uint32_t>& op_blk = s->h_op_blk;
    std::vector<uint32_t>& op_vseg = s->h_op_vseg;
    const bool host_pt_job = !big; /* big uploads fill point -> job on the device (k_ptjob): no 4 B/point host loop and copy */
    if (host_pt_job) pt_job.assign(b->n_pts, 0xFFFFFFFFu);
    op_aux.assign(b->n_ops, 0u);
    op_blk.assign(b->n_ops, 0xFFFFFFFFu);
    op_vseg.assign(b->n_ops, 0u);
    /* op -> job; stroke slot -> op; first virtual segment (edges + the two cap stubs of Round/Square caps,
     * line.rs:33-57) of every stroke slot: k_stroke_bin runs one thread per virtual segment of the scene */
    std::vector<uint32_t>& op_job = s->h_op_job;
    op_job.assign(b->n_ops, 0u);
    uint32_t n_strokes = 0;
    size_t n_blk = 0; /* 64-edge blocks of the ops with more than 64 edges */
    size_t n_vsegs = 0, n_fills = 0;
    uint32_t max_job_ops = 0;
    for (size_t j = 0; j < b->n_jobs; ++j) {
        const osmt_tile_job& job = b->jobs[j];
        max_job_ops = std::max(max_job_ops, job.n_ops);
        if (host_pt_job)
            for (uint32_t i = 0; i < job.n_pts; ++i) pt_job[job.pt_off + i] = (uint32_t)j;
        for (uint32_t k = 0; k < job.n_ops; ++k) {
            const osmt_op& op = b->ops[job.op_off + k];
            op_job[job.op_off + k] = (uint32_t)j;
            if (op.kind == OSMT_OP_NONE) continue;
            size_t ne = 0;
            for (uint32_t r = 0; r < op.n_rings; ++r) {
                const uint32_t np = b->rings[op.ring_off + r].n_pts;
                if (np >= 2) ne += np - 1;
            }
            if (ne > 64 && n_blk + (ne + 63) / 64 < 0xFFFFFFFFull) {
                op_blk[job.op_off + k] = (uint32_t)n_blk;
                n_blk += (ne + 63) / 64;
            }
            if (op.kind == OSMT_OP_STROKE) {
                op_aux[job.op_off + k] = n_strokes++;
                const size_t nv = ne + ((op.cap == OSMT_CAP_ROUND || op.cap == OSMT_CAP_SQUARE) ? 2u : 0u);
                op_vseg[job.op_off + k] = (uint32_t)n_vsegs; /* k_opinfo writes the per-segment tables from here on */
                n_vsegs += nv;
            } else {
                ++n_fills;
            }
        }
    }
    if (n_vsegs >= 0xFFFFFFFFull) {
        delete s;
        return fail(OSMT_INVALID_ARG, "batch too large for 32-bit indices (stroke segments)");
    }

    s->ctx = ctx;
    ctx->refs.fetch_add(1);
    s->h_err = err_slot_acquire(ctx);
    s->n_jobs = (uint32_t)b->n_jobs;
    s->max_job_ops = max_job_ops;
    s->n_ops = (uint32_t)b->n_ops;
    s->n_rings = (uint32_t)b->n_rings;
    s->n_pts = (uint32_t)b->n_pts;
    s->n_dashes = (uint32_t)b->n_dashes;
    s->n_strokes = n_strokes;
    s->n_blk = (uint32_t)n_blk;
    s->n_vsegs = (uint32_t)n_vsegs;
    s->scale = b->scale;
    s->coord_kind = b->coord_kind;

    const size_t o_info = carve(b->n_ops * sizeof(osmt_opinfo));
    /* per virtual segment (not per point: two stroke ops may share a ring, e.g. a casing and its stroke) */
    const size_t o_aux = carve((size_t)(n_strokes + 1) * sizeof(osmt_stroke_aux));
    const size_t o_dseg = carve((size_t)(n_strokes + 1) * OSMT_MAX_DASH_SEGS * sizeof(osmt_dash_seg)); /* touched by dashed ops only */
    const size_t sub_rows = (size_t)OSMT_TILE_SIZE * b->scale / OSMT_SUB_H;
    const size_t o_submask = carve(b->n_ops * sub_rows * 4);
    const size_t o_blk = carve((n_blk + 1) * sizeof(osmt_blk_bbox));
    const size_t o_vseg = carve((n_vsegs + 1) * sizeof(osmt_vseg));
    const size_t n_sub = ((size_t)OSMT_TILE_SIZE * b->scale / OSMT_SUB_W) * sub_rows;
    const size_t o_cursors = carve(32 + b->n_jobs * n_sub * 4); /* cursors + list counts: zeroed together every frame */
    const size_t o_hdr = carve(b->n_jobs * n_sub * sizeof(uint2));
    s->bytes = off - (split ? front_bytes : 0) + 256;
    mark(1);
    hipError_t e = dev_alloc(ctx, (void**)&s->d_base, s->bytes);
    mark(2);
    if (e != hipSuccess) {
        scene_delete(s);
        return fail(e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "hipMalloc(%zu) failed: %s", off,
                    hipGetErrorString(e));
    }
    /* front offsets live in d_front when the upload is split, everything behind them in d_base */
    char* const fbase = split ? d_front : s->d_base;
    char* const rbase = split ? s->d_base - front_bytes : s->d_base;
    s->d_jobs = (osmt_tile_job*)(fbase + o_jobs);
    s->d_ops = (osmt_op*)(fbase + o_ops);
    s->d_rings = (osmt_ring*)(fbase + o_rings);
    s->d_latlon = (double*)(fbase + o_latlon);
    s->d_node_refs = (uint32_t*)(fbase + o_refs);
    s->d_pts = (int32_t*)(fbase + o_pts);
    s->d_dashes = (double*)(fbase + o_dashes);
    s->d_pt_job = (uint32_t*)(fbase + o_ptjob);
    s->d_op_aux = (uint32_t*)(fbase + o_opaux);
    s->d_info = (osmt_opinfo*)(rbase + o_info);
    s->d_aux = (osmt_stroke_aux*)(rbase + o_aux);
    s->d_dseg = (osmt_dash_seg*)(rbase + o_dseg);
    s->d_submask = (uint32_t*)(rbase + o_submask);
    s->d_op_blk = (uint32_t*)(fbase + o_opblk);
    s->d_op_vseg = (uint32_t*)(fbase + o_opvseg);
    s->d_blk = (osmt_blk_bbox*)(rbase + o_blk);
    s->d_op_job = (uint32_t*)(fbase + o_opjob);
    s->d_vseg = (osmt_vseg*)(rbase + o_vseg);
    s->d_cursors = (unsigned long long*)(rbase + o_cursors);
    s->d_cnt = (uint32_t*)(rbase + o_cursors + 32);
    s->d_hdr = (uint2*)(rbase + o_hdr);

    auto up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        if (!bytes) return hipSuccess;
        return st ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) : hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    };
    hipError_t err = hipSuccess;
    char* stage = (st && front_bytes <= STAGE_MAX_BYTES) ? (char*)stage_acquire(ctx, front_bytes) : nullptr;
    if (stage) {
        /* small call: pack the arrays into pinned staging and move them with ONE copy (a dozen pageable copies cost
         * more host time than the kernels of a single tile) */
        auto put = [&](size_t o, const void* src, size_t bytes) {
            if (bytes) memcpy(stage + o, src, bytes);
        };
        put(o_jobs, b->jobs, b->n_jobs * sizeof(osmt_tile_job));
        put(o_ops, b->ops, b->n_ops * sizeof(osmt_op));
        put(o_rings, b->rings, b->n_rings * sizeof(osmt_ring));
        if (ll) put(o_latlon, b->latlon, b->n_pts * 16);
        if (nr) put(o_latlon, b->nodes, b->n_nodes * 16);
        if (nr) put(o_refs, b->node_refs, b->n_pts * 4);
        if (!ll && !nr) put(o_pts, b->points, b->n_pts * 8);
        put(o_dashes, b->dashes, b->n_dashes * 8);
        if (host_pt_job) put(o_ptjob, pt_job.data(), b->n_pts * 4);
        put(o_opaux, op_aux.data(), b->n_ops * 4);
        put(o_opblk, op_blk.data(), b->n_ops * 4);
        put(o_opvseg, op_vseg.data(), b->n_ops * 4);
        put(o_opjob, op_job.data(), b->n_ops * 4);
        s->h_stage = stage;
        err = hipMemcpyAsync(s->d_base, stage, front_bytes, hipMemcpyHostToDevice, st);
        if (err == hipSuccess && !host_pt_job) err = osmt_launch_ptjob(s->d_jobs, s->n_jobs, s->d_pt_job, s->n_pts, st);
        if (err != hipSuccess) {
            stage_release(ctx, stage);
            dev_free(ctx, s->d_base);
            scene_delete(s);
            return fail(OSMT_HIP_ERROR, "upload failed: %s", hipGetErrorString(err));
        }
        mark(3);
        rc = cc.join(); /* no kernel has seen the coordinates yet */
        mark(4);
        if (rc == OSMT_OK) rc = scene_size_arenas(ctx, s, n_fills, allow_guess);
        mark(5);
        report();
        if (rc != OSMT_OK) {
            osmt_scene_free(s);
            return rc;
        }
        *out_scene = s;
        return OSMT_OK;
    }
    if (!split) { /* (split: the helper thread is copying these) */
        if (err == hipSuccess) err = up(s->d_jobs, b->jobs, b->n_jobs * sizeof(osmt_tile_job));
        if (err == hipSuccess) err = up(s->d_ops, b->ops, b->n_ops * sizeof(osmt_op));
        if (err == hipSuccess) err = up(s->d_rings, b->rings, b->n_rings * sizeof(osmt_ring));
        if (err == hipSuccess && ll) err = up(s->d_latlon, b->latlon, b->n_pts * 16);
        if (err == hipSuccess && nr) err = up(s->d_latlon, b->nodes, b->n_nodes * 16);
        if (err == hipSuccess && nr) err = up(s->d_node_refs, b->node_refs, b->n_pts * 4);
        if (err == hipSuccess && !ll && !nr) err = up(s->d_pts, b->points, b->n_pts * 8);
        if (err == hipSuccess) err = up(s->d_dashes, b->dashes, b->n_dashes * 8);
    }
    if (err == hipSuccess && host_pt_job) err = up(s->d_pt_job, pt_job.data(), b->n_pts * 4);
    if (err == hipSuccess && !host_pt_job && !split) err = osmt_launch_ptjob(s->d_jobs, s->n_jobs, s->d_pt_job, s->n_pts, st ? st : nullptr);
    if (err == hipSuccess) err = up(s->d_op_aux, op_aux.data(), b->n_ops * 4);
    if (err == hipSuccess) err = up(s->d_op_blk, op_blk.data(), b->n_ops * 4);
    if (err == hipSuccess) err = up(s->d_op_vseg, op_vseg.data(), b->n_ops * 4);
    if (err == hipSuccess) err = up(s->d_op_job, op_job.data(), b->n_ops * 4);
    if (err != hipSuccess) {
        cc.wait();
        (void)hipStreamSynchronize(st ? st : nullptr); /* copies into d_base may be on their way too */
        dev_free(ctx, s->d_base);
        scene_delete(s);
        return fail(OSMT_HIP_ERROR, "upload failed: %s", hipGetErrorString(err));
    }
    /* from here on the scene owns the front buffer: every later exit goes through osmt_scene_free (which waits for the stream) */
    s->d_front = d_front;
    fg.p = nullptr;
    mark(3);
    rc = cc.join(); /* no kernel has seen the coordinates yet; the helper's copies are on the stream */
    if (rc == OSMT_OK && split && !host_pt_job) { /* the point -> job kernel reads the jobs the helper copied: behind the join */
        err = osmt_launch_ptjob(s->d_jobs, s->n_jobs, s->d_pt_job, s->n_pts, st);
        if (err != hipSuccess) rc = fail(OSMT_HIP_ERROR, "upload failed: %s", hipGetErrorString(err));
    }
    mark(4);
    if (rc == OSMT_OK) rc = scene_size_arenas(ctx, s, n_fills, allow_guess);
    mark(5);
    report();
    if (rc != OSMT_OK) {
        osmt_scene_free(s);
        return rc;
    }
    *out_scene = s;
    return OSMT_OK;
}

int osmt_validate_batch(const osmt_batch* b) {
    return guarded([&] { return validate_batch(b); });
}

/* Public scenes: the upload runs on a pooled private stream (never the NULL stream, whose copies order themselves
 * against other workers' streams) and is complete when the call returns; the scene is then free to be rendered on any
 * stream of the caller. */
static int osmt_scene_upload_body(osmt_ctx* ctx, const osmt_batch* b, osmt_scene** out_scene) {
    if (!ctx || !out_scene) return fail(OSMT_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    HIP_TRY(stream_acquire(ctx, &st));
    int rc = scene_upload_impl(ctx, b, out_scene, st);
    if (rc == OSMT_OK) {
        osmt_scene* s = *out_scene;
        const hipError_t e = hipStreamSynchronize(st);
        s->own_stream = nullptr; /* from here on a public scene: waits go through its last-use events */
        stage_release(ctx, s->h_stage);
        s->h_stage = nullptr;
        if (e != hipSuccess) {
            osmt_scene_free(s);
            *out_scene = nullptr;
            rc = fail(OSMT_HIP_ERROR, "upload failed: %s", hipGetErrorString(e));
        }
    }
    stream_release(ctx, st);
    return rc;
}

int osmt_scene_upload(osmt_ctx* ctx, const osmt_batch* b, osmt_scene** out_scene) {
    return guarded([&] { return osmt_scene_upload_body(ctx, b, out_scene); });
}

void osmt_scene_free(osmt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    /* in-flight kernels may still read the scene and a cached buffer can be handed to the next upload at once:
     * wait for the scene's own stream (internal per-call scenes) or for the last launch on every stream the scene
     * was rendered on (public scenes) — not for the device: other workers' streams keep running */
    (void)scene_wait_idle(s);
    dev_free(s->ctx, s->d_base);
    dev_free(s->ctx, s->d_front);
    dev_free(s->ctx, s->d_arena);
    dev_free(s->ctx, s->d_lab_base);
    stage_release(s->ctx, s->h_stage);
    scene_delete(s);
}

/* Drawer::draw_labels (drawer.rs:221-262) as data: validates, sizes each label's coverage window and
 * uploads.  Window of a label = stripes its draw_line calls can create inside labels_bb's rows
 * (tile_pixels.rs:67-72) x every column those stripes can hold a key in (+-2 cells of slack for the
 * rounding of eval_x_at_y, font/rasterizer.rs:37). */
static int osmt_scene_set_labels_body(osmt_ctx* ctx, osmt_scene* sc, const osmt_label_batch* lb) {
    if (!ctx || !sc || sc->ctx != ctx) return fail(OSMT_INVALID_ARG, "bad ctx/scene");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = sc->own_stream;
    hipStream_t pooled = nullptr; /* public scene: copies on a private stream, complete on return (never the NULL stream) */
    if (!st) {
        HIP_TRY(stream_acquire(ctx, &pooled));
        st = pooled;
    }
    struct release_pooled {
        osmt_ctx* c;
        hipStream_t s;
        ~release_pooled() {
            if (s) {
                (void)hipStreamSynchronize(s);
                stream_release(c, s);
            }
        }
    } pooled_guard{ctx, pooled};
    HIP_TRY(scene_wait_idle(sc)); /* the previous label pass of THIS scene may still be in flight; nobody else is waited for */
    dev_free(ctx, sc->d_lab_base);
    sc->d_lab_base = nullptr;
    sc->n_labels = sc->n_label_segs = 0;
    if (!lb || lb->n_labels == 0) return OSMT_OK;
    if (!lb->labels || !lb->job_label_off || (lb->n_segs && !lb->segs)) return fail(OSMT_INVALID_ARG, "NULL label pool");
    if (lb->n_labels >= 0xFFFFFFFFull || lb->n_segs >= 0xFFFFFFFFull) return fail(OSMT_INVALID_ARG, "label batch too large");
    if (lb->job_label_off[0] != 0 || lb->job_label_off[sc->n_jobs] != lb->n_labels)
        return fail(OSMT_INVALID_ARG, "job_label_off must run from 0 to n_labels over n_jobs + 1 entries");
    for (uint32_t j = 0; j < sc->n_jobs; ++j)
        if (lb->job_label_off[j] > lb->job_label_off[j + 1]) return fail(OSMT_INVALID_ARG, "job_label_off is not monotonic");
    const double LIM = 1048576.0; /* 2^20 */
    for (size_t i = 0; i < 4 * lb->n_segs; ++i)
        if (!(std::fabs(lb->segs[i]) <= LIM)) return fail(OSMT_UNSUPPORTED, "label segment %zu: coordinate not finite or |v| > 2^20", i / 4);

    std::vector<osmt_image_desc> images;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        images = ctx->images;
    }
    const int32_t W = (int32_t)(OSMT_TILE_SIZE * sc->scale);
    std::vector<osmt_labelinfo>& info = sc->h_lab_info; /* kept alive: the upload may be stream-ordered */
    std::vector<uint32_t>& wide = sc->h_lab_wide;
    std::vector<osmt_label_band>& bands = sc->h_lab_bands;
    bands.clear();
    info.assign(lb->n_labels, osmt_labelinfo{});
    wide.clear();
    size_t cells = 0, wide_cells = 0, bit_words = 0; /* bit_words < cells / 64 + bands <= 2^26 */
    for (uint32_t j = 0; j < sc->n_jobs; ++j) {
        for (uint32_t l = lb->job_label_off[j]; l < lb->job_label_off[j + 1]; ++l) {
            const osmt_label& in = lb->labels[l];
            osmt_labelinfo& o = info[l];
            memset(&o, 0, sizeof o);
            o.ry0 = 1;
            o.ry1 = 0;
            if (in.has_icon && in.image_id < images.size()) { /* icon missing from the cache: Some(0), no blit (labeler.rs:64-66) */
                if (!(std::fabs(in.icon_center_x) <= LIM) || !(std::fabs(in.icon_center_y) <= LIM))
                    return fail(OSMT_UNSUPPORTED, "label %u: icon centre not finite or |v| > 2^20", l);
                const osmt_image_desc& im = images[in.image_id];
                o.icon_w = im.width;
                o.icon_h = im.height;
                o.icon_off = im.offset;
                o.icon_x = (int32_t)(in.icon_center_x - ((double)im.width / 2.0)); /* get_start_coord (labeler.rs:92-95) */
                o.icon_y = (int32_t)(in.icon_center_y - ((double)im.height / 2.0));
            }
            o.has_text = in.has_text ? 1 : 0;
            memcpy(o.color, in.text_color, 3);
            if (!in.has_text || in.n_segs == 0) continue;
            if ((size_t)in.seg_off + in.n_segs > lb->n_segs) return fail(OSMT_INVALID_ARG, "label %u: segment range out of bounds", l);
            o.seg_off = in.seg_off;
            o.n_segs = in.n_segs;
            int32_t ry0 = INT32_MAX, ry1 = INT32_MIN, cx0 = INT32_MAX, cx1 = INT32_MIN;
            for (uint32_t k = 0; k < in.n_segs; ++k) {
                const double* q = lb->segs + 4 * ((size_t)in.seg_off + k);
                if (q[3] - q[1] == 0.0) continue; /* draw_line returns (font/rasterizer.rs:30-32) */
                int32_t a = (int32_t)std::floor(std::fmin(q[1], q[3])), b = (int32_t)std::floor(std::fmax(q[1], q[3]));
                a = std::max(a, -W);
                b = std::min(b, 2 * W - 1);
                if (a > b) continue; /* no stripe inside labels_bb */
                ry0 = std::min(ry0, a);
                ry1 = std::max(ry1, b);
                cx0 = std::min(cx0, (int32_t)std::floor(std::fmin(q[0], q[2])) - 2);
                cx1 = std::max(cx1, (int32_t)std::floor(std::fmax(q[0], q[2])) + 3);
            }
            if (ry0 > ry1) continue;
            const size_t rows = (size_t)(ry1 - ry0 + 1), cols = (size_t)(cx1 - cx0 + 1);
            if (rows * cols > ((size_t)1 << 24))
                return fail(OSMT_UNSUPPORTED, "label %u: coverage window of %zu x %zu cells is too large", l, cols, rows);
            o.ry0 = ry0;
            o.ry1 = ry1;
            o.cx0 = cx0;
            o.cols = (uint32_t)cols;
            o.plane_off = cells;
            cells += rows * cols;
            if (cols > OSMT_LABEL_LDS_CELLS) {
                if (wide_cells + 64 * cols >= 0xFFFFFFFFull) return fail(OSMT_UNSUPPORTED, "too many wide label windows");
                o.wide_off = (uint32_t)wide_cells;
                wide_cells += 64 * cols;
                wide.push_back(l);
            } else {
                const uint32_t band_rows = osmt_label_band_rows((uint32_t)cols);
                o.wide_off = (uint32_t)bit_words;
                for (uint32_t rb = 0; rb < rows; rb += band_rows) {
                    bands.push_back(osmt_label_band{l, rb});
                    bit_words += osmt_label_band_words((uint32_t)cols);
                }
            }
        }
    }
    if (cells > ((size_t)1 << 31)) return fail(OSMT_UNSUPPORTED, "label coverage windows need %zu cells (> 2^31)", cells);
    /* every band scans all of its label's draw_line calls: longest first (ties keep draw order) */
    std::stable_sort(bands.begin(), bands.end(),
                     [&](const osmt_label_band& a, const osmt_label_band& b) { return info[a.label].n_segs > info[b.label].n_segs; });
    /* workgroup b runs on XCD b mod 8, each with an L2 of its own: the bands of one label (neighbours in the sorted
     * list, all reading the same draw_line calls) are dealt to ONE residue class, G at a time, so that the calls come
     * from HBM once and from that XCD's L2 for the other bands */
    {
        constexpr size_t G = 16, CH = 8 * G;
        std::vector<osmt_label_band> dealt(bands.size());
        const size_t full = bands.size() / CH * CH;
        for (size_t i = 0; i < full; ++i) {
            const size_t c = i / CH, j = i % CH;
            dealt[c * CH + (j % G) * 8 + j / G] = bands[i];
        }
        for (size_t i = full; i < bands.size(); ++i) dealt[i] = bands[i];
        bands.swap(dealt);
    }

    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    const size_t EW = 3 * (size_t)W;
    const size_t words = (EW * EW + 31) / 32;
    const size_t o_info = carve(lb->n_labels * sizeof(osmt_labelinfo));
    const size_t o_off = carve(((size_t)sc->n_jobs + 1) * 4);
    const size_t o_segs = carve(lb->n_segs * 32);
    const size_t o_a = carve((cells + 1) * 8);
    const size_t o_s = carve((wide_cells + 1) * 8);
    const size_t o_bits = carve((bit_words + 2) * 8); /* + the word a funnel read may touch behind the last stream */
    const size_t o_wide = carve((wide.size() + 1) * 4);
    const size_t o_bands = carve((bands.size() + 1) * sizeof(osmt_label_band));
    const size_t o_bm = carve(words * 4 <= 96 * 1024 ? 4 : (size_t)sc->n_jobs * words * 4); /* scale 1: the map lives in LDS */
    const size_t o_tl = carve(lb->n_labels * sizeof(osmt_tile_label));
    const size_t o_tlc = carve((size_t)sc->n_jobs * 4);
    const size_t o_ok = carve(lb->n_labels);
    const size_t o_err = carve(4);
    hipError_t e = dev_alloc(ctx, (void**)&sc->d_lab_base, off + 256);
    if (e != hipSuccess) {
        sc->d_lab_base = nullptr;
        return fail(e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "hipMalloc(%zu) for labels failed: %s", off,
                    hipGetErrorString(e));
    }
    char* base = sc->d_lab_base;
    sc->d_lab = (osmt_labelinfo*)(base + o_info);
    sc->d_job_label_off = (uint32_t*)(base + o_off);
    sc->d_lab_segs = (double*)(base + o_segs);
    sc->d_lab_a = (double*)(base + o_a);
    sc->d_lab_s_wide = (double*)(base + o_s);
    sc->d_lab_bits = (unsigned long long*)(base + o_bits);
    sc->d_lab_wide = (uint32_t*)(base + o_wide);
    sc->n_lab_wide = (uint32_t)wide.size();
    sc->d_lab_bands = (osmt_label_band*)(base + o_bands);
    sc->n_lab_bands = (uint32_t)bands.size();
    sc->d_tile_labels = (osmt_tile_label*)(base + o_tl);
    sc->d_tile_label_cnt = (uint32_t*)(base + o_tlc);
    sc->d_lab_bitmap = (uint32_t*)(base + o_bm);
    sc->d_lab_ok = (uint8_t*)(base + o_ok);
    sc->d_lab_err = (uint32_t*)(base + o_err);
    auto up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        if (!bytes) return hipSuccess;
        return st ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) : hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    };
    e = up(sc->d_lab, info.data(), info.size() * sizeof(osmt_labelinfo));
    if (e == hipSuccess) e = up(sc->d_job_label_off, lb->job_label_off, ((size_t)sc->n_jobs + 1) * 4);
    if (e == hipSuccess) e = up(sc->d_lab_segs, lb->segs, lb->n_segs * 32);
    if (e == hipSuccess) e = up(sc->d_lab_wide, wide.data(), wide.size() * 4);
    if (e == hipSuccess) e = up(sc->d_lab_bands, bands.data(), bands.size() * sizeof(osmt_label_band));
    /* The verdicts and the error word are cleared by the label stage itself, on the render stream (osmt_launch_labels).
     * Until the first render with these labels they read "no label has been placed, no error": a status read between
     * osmt_scene_set_labels and the next render used to return whatever the recycled buffer held (found by the poisoned
     * allocator in round 5: error word 0xA5A5A5A5). */
    if (e == hipSuccess)
        e = st ? hipMemsetAsync(base + o_ok, 0, o_err + 4 - o_ok, st) : hipMemset(base + o_ok, 0, o_err + 4 - o_ok);
    if (e != hipSuccess) {
        dev_free(ctx, sc->d_lab_base);
        sc->d_lab_base = nullptr;
        return fail(OSMT_HIP_ERROR, "label upload failed: %s", hipGetErrorString(e));
    }
    sc->n_labels = (uint32_t)lb->n_labels;
    sc->n_label_segs = (uint32_t)lb->n_segs;
    return OSMT_OK;
}

int osmt_scene_set_labels(osmt_ctx* ctx, osmt_scene* sc, const osmt_label_batch* lb) {
    return guarded([&] { return osmt_scene_set_labels_body(ctx, sc, lb); });
}

static int osmt_scene_read_label_status_body(osmt_ctx* ctx, osmt_scene* sc, uint8_t* ok) {
    if (!ctx || !sc || sc->ctx != ctx) return fail(OSMT_INVALID_ARG, "bad ctx/scene");
    if (sc->n_labels == 0) return OSMT_OK;
    if (!ok) return fail(OSMT_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(scene_wait_idle(sc));
    uint32_t err = 0;
    HIP_TRY(copy_back(ctx, &err, sc->d_lab_err, 4));
    if (err) return fail(OSMT_HIP_ERROR, "label coverage window overflow (internal error %u)", err);
    HIP_TRY(copy_back(ctx, ok, sc->d_lab_ok, sc->n_labels));
    return OSMT_OK;
}

int osmt_scene_read_label_status(osmt_ctx* ctx, osmt_scene* sc, uint8_t* ok) {
    return guarded([&] { return osmt_scene_read_label_status_body(ctx, sc, ok); });
}

static int osmt_render_scene_body(osmt_ctx* ctx, osmt_scene* scene, void* d_out_rgba, size_t stride, void* stream) {
    return render_impl(ctx, scene, 7u, d_out_rgba, stride, false, stream);
}

int osmt_render_scene(osmt_ctx* ctx, osmt_scene* scene, void* d_out_rgba, size_t stride, void* stream) {
    return guarded([&] { return osmt_render_scene_body(ctx, scene, d_out_rgba, stride, stream); });
}

static int osmt_render_scene_f64_body(osmt_ctx* ctx, osmt_scene* scene, void* d_out_f64, void* stream) {
    return render_impl(ctx, scene, 7u, d_out_f64, 0, true, stream);
}

int osmt_render_scene_f64(osmt_ctx* ctx, osmt_scene* scene, void* d_out_f64, void* stream) {
    return guarded([&] { return osmt_render_scene_f64_body(ctx, scene, d_out_f64, stream); });
}

static int osmt_render_scene_stages_body(osmt_ctx* ctx, osmt_scene* scene, uint32_t stage_mask, void* d_out_rgba, size_t stride,
                             void* stream) {
    return render_impl(ctx, scene, stage_mask & 7u, d_out_rgba, stride, false, stream);
}

int osmt_render_scene_stages(osmt_ctx* ctx, osmt_scene* scene, uint32_t stage_mask, void* d_out_rgba, size_t stride,
                             void* stream) {
    return guarded([&] { return osmt_render_scene_stages_body(ctx, scene, stage_mask, d_out_rgba, stride, stream); });
}

static int osmt_scene_check_body(osmt_ctx* ctx, osmt_scene* sc) {
    if (!ctx || !sc || sc->ctx != ctx) return fail(OSMT_INVALID_ARG, "bad ctx/scene");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(scene_wait_idle(sc));
    return prepass_error_check(sc);
}

int osmt_scene_check(osmt_ctx* ctx, osmt_scene* scene) {
    return guarded([&] { return osmt_scene_check_body(ctx, scene); });
}

static int osmt_scene_read_points_body(osmt_ctx* ctx, osmt_scene* sc, int32_t* xy) {
    if (!ctx || !sc || !xy) return fail(OSMT_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(scene_wait_idle(sc));
    if (sc->n_pts) HIP_TRY(copy_back(ctx, xy, sc->d_pts, (size_t)sc->n_pts * 8));
    return OSMT_OK;
}

int osmt_scene_read_points(osmt_ctx* ctx, osmt_scene* sc, int32_t* xy) {
    return guarded([&] { return osmt_scene_read_points_body(ctx, sc, xy); });
}

int osmt_render_batch(osmt_ctx* ctx, const osmt_batch* batch, uint8_t* out_rgba, size_t stride) {
    return osmt_render_batch_labels(ctx, batch, nullptr, out_rgba, stride);
}

static int osmt_render_batch_labels_once(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgba, size_t stride,
                                         bool rgb, bool trusted, bool allow_guess);

/* the host-buffer render: arenas guessed from the recent densities first; a miss renders again with exact sizing */
static int osmt_render_batch_labels_body(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgba,
                                         size_t stride, bool rgb = false, bool trusted = false) {
    g_arena_guess_missed = false;
    int rc = osmt_render_batch_labels_once(ctx, batch, labels, out_rgba, stride, rgb, trusted, true);
    if (rc != OSMT_OK && g_arena_guess_missed) {
        g_arena_guess_missed = false;
        rc = osmt_render_batch_labels_once(ctx, batch, labels, out_rgba, stride, rgb, true, false); /* validated the first time */
    }
    return rc;
}

static int osmt_render_batch_labels_once(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgba, size_t stride,
                                         bool rgb, bool trusted, bool allow_guess) {
    if (!ctx || !out_rgba) return fail(OSMT_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = nullptr; /* the whole call lives on its own stream: concurrent callers overlap on the GPU */
    HIP_TRY(stream_acquire(ctx, &st));
    osmt_scene* sc = nullptr;
    int rc = scene_upload_impl(ctx, batch, &sc, st, trusted, allow_guess);
    if (rc != OSMT_OK) {
        stream_release(ctx, st);
        return rc;
    }
    if (labels) {
        rc = osmt_scene_set_labels(ctx, sc, labels);
        if (rc != OSMT_OK) {
            osmt_scene_free(sc);
            stream_release(ctx, st);
            return rc;
        }
    }
    const size_t W = (size_t)OSMT_TILE_SIZE * batch->scale;
    const size_t tile_bytes = W * W * 4;                   /* what k_raster writes */
    const size_t host_bytes = rgb ? W * W * 3 : tile_bytes; /* what crosses PCIe */
    if (stride < host_bytes) {
        osmt_scene_free(sc);
        stream_release(ctx, st);
        return fail(OSMT_INVALID_ARG, rgb ? "out_tile_stride_bytes < W*H*3" : "out_tile_stride_bytes < W*H*4");
    }
    /* Output in pinned host memory (osmt_host_alloc / hipHostMalloc / a registered range) and a batch worth
     * splitting: kernels of chunk k overlap the D2H copy of chunk k-1 on a second stream. */
    bool pinned = false;
    if (batch->n_jobs >= 2u * std::max<uint32_t>(1u, 128u / (batch->scale * batch->scale))) { /* the query is not free: only where it can pay */
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, out_rgba) == hipSuccess)
            pinned = at.type == hipMemoryTypeHost;
        else
            (void)hipGetLastError();
    }
    const uint32_t chunk = std::max<uint32_t>(1u, 128u / (batch->scale * batch->scale));
    if (pinned && batch->n_jobs >= 2u * chunk) {
        hipStream_t s_k = st, s_c = nullptr; /* kernels stay on the call's stream (the scene was uploaded there) */
        hipEvent_t done[2] = {nullptr, nullptr}, freed[2] = {nullptr, nullptr};
        char* d_out = nullptr;
        hipError_t e = stream_acquire(ctx, &s_c);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&freed[k], hipEventDisableTiming);
        }
        char* d_rgb = nullptr;
        /* RGB8: k_raster packs the triples itself (no RGBA8 framebuffer, no packing kernel) */
        if (e == hipSuccess && !rgb) e = dev_alloc(ctx, (void**)&d_out, 2 * (size_t)chunk * tile_bytes);
        if (e == hipSuccess && rgb) e = dev_alloc(ctx, (void**)&d_rgb, 2 * (size_t)chunk * host_bytes);
        if (e != hipSuccess) rc = fail(e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "pipeline setup failed: %s", hipGetErrorString(e));
        if (rc == OSMT_OK) rc = render_impl(ctx, sc, 1u | 2u | 8u, nullptr, tile_bytes, false, s_k);
        uint32_t c = 0;
        for (uint32_t first = 0; rc == OSMT_OK && first < batch->n_jobs; first += chunk, ++c) {
            const uint32_t cnt = (uint32_t)std::min<size_t>(chunk, batch->n_jobs - first);
            const int k = (int)(c & 1u);
            char* dst = rgb ? d_rgb + (size_t)k * chunk * host_bytes : d_out + (size_t)k * chunk * tile_bytes;
            if (c >= 2) e = hipStreamWaitEvent(s_k, freed[k], 0);
            if (e == hipSuccess) rc = render_impl(ctx, sc, 4u | 16u | (rgb ? 32u : 0u), dst, host_bytes, false, s_k, first, cnt);
            if (rc != OSMT_OK) break;
            const char* src = dst;
            if (e == hipSuccess) e = hipEventRecord(done[k], s_k);
            if (e == hipSuccess) e = hipStreamWaitEvent(s_c, done[k], 0);
            if (e == hipSuccess) {
                if (stride == host_bytes) /* tightly packed tiles: one linear copy (2D copies are slower) */
                    e = hipMemcpyAsync(out_rgba + (size_t)first * stride, src, (size_t)cnt * host_bytes, hipMemcpyDeviceToHost, s_c);
                else
                    e = hipMemcpy2DAsync(out_rgba + (size_t)first * stride, stride, src, host_bytes, host_bytes, cnt,
                                         hipMemcpyDeviceToHost, s_c);
            }
            if (e == hipSuccess) e = hipEventRecord(freed[k], s_c);
            if (e != hipSuccess) rc = fail(OSMT_HIP_ERROR, "pipelined readback failed: %s", hipGetErrorString(e));
        }
        if (s_k) (void)hipStreamSynchronize(s_k);
        if (rc == OSMT_OK) rc = label_error_check(sc, s_k);
        if (s_c) {
            e = hipStreamSynchronize(s_c);
            if (e != hipSuccess && rc == OSMT_OK) rc = fail(OSMT_HIP_ERROR, "pipelined readback failed: %s", hipGetErrorString(e));
        }
        for (int k = 0; k < 2; ++k) {
            if (done[k]) (void)hipEventDestroy(done[k]);
            if (freed[k]) (void)hipEventDestroy(freed[k]);
        }
        osmt_scene_free(sc);
        dev_free(ctx, d_out);
        dev_free(ctx, d_rgb);
        stream_release(ctx, s_c);
        stream_release(ctx, st);
        return rc;
    }
    void* d_out = nullptr;
    /* A FEW tiles into PINNED caller memory (osmt_host_alloc / hipHostMalloc: mapped into the device's address space): k_raster
     * writes the pixels straight there — no device framebuffer, no copy.  Round 5 measured why it matters: asynchronous
     * device-to-host copies into pinned buffers from several threads at once queue up behind each other in the runtime (one-tile
     * requests, 4 threads: 10.9 k tiles/s with pinned caller buffers against 35.5 k with pageable ones, which the runtime
     * copies synchronously on the calling thread; profiles/r05_f_worker_pinned.txt).  The worker entry's gathered groups land
     * in the library's own pinned staging, so they take this path too: 16 tiles = the groups that 32 request threads form
     * (98 k tiles/s against 73 k with 8 tiles, the same with 32; profiles/r05_h_zero_copy_threshold.txt).
     * OSMT_ZERO_COPY_TILES: most tiles of such a call (0: never). */
    bool zero_copy = false;
    static const uint32_t zc_max = [] {
        const char* v = getenv("OSMT_ZERO_COPY_TILES");
        return (uint32_t)(v ? std::min(std::max(atoi(v), 0), 64) : 16);
    }();
    if (batch->n_jobs && batch->n_jobs <= zc_max && ((uintptr_t)out_rgba & 3u) == 0u && (stride & 3u) == 0u) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, out_rgba) == hipSuccess) {
            if (at.type == hipMemoryTypeHost && at.devicePointer) {
                d_out = at.devicePointer;
                zero_copy = true;
            }
        } else {
            (void)hipGetLastError(); /* pageable memory: not an error */
        }
    }
    if (batch->n_jobs && !zero_copy) {
        hipError_t e = dev_alloc(ctx, &d_out, batch->n_jobs * host_bytes); /* RGB8: written by k_raster as packed triples */
        if (e != hipSuccess) {
            osmt_scene_free(sc);
            stream_release(ctx, st);
            return fail(OSMT_OOM, "hipMalloc(output) failed: %s", hipGetErrorString(e));
        }
    }
    rc = render_impl(ctx, sc, 7u | (rgb ? 32u : 0u), d_out ? d_out : (void*)4, zero_copy ? stride : host_bytes, false, st);
    if (rc == OSMT_OK && batch->n_jobs) {
        hipError_t e = hipSuccess;
        const void* src = d_out;
        if (rc == OSMT_OK && e == hipSuccess && !zero_copy) {
            if (stride == host_bytes)
                e = hipMemcpyAsync(out_rgba, src, batch->n_jobs * host_bytes, hipMemcpyDeviceToHost, st);
            else
                e = hipMemcpy2DAsync(out_rgba, stride, src, host_bytes, host_bytes, batch->n_jobs, hipMemcpyDeviceToHost, st);
        }
        if (rc == OSMT_OK && e == hipSuccess) e = batch->n_jobs <= 64 ? stream_sync_small(st) : hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(OSMT_HIP_ERROR, "readback failed: %s", hipGetErrorString(e));
    }
    if (rc == OSMT_OK) rc = label_error_check(sc, st);
    osmt_scene_free(sc); /* waits for the call's stream */
    if (!zero_copy) dev_free(ctx, d_out);
    stream_release(ctx, st);
    return rc;
}

int osmt_render_batch_rgb(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgb, size_t stride) {
    return guarded([&] { return osmt_render_batch_labels_body(ctx, batch, labels, out_rgb, stride, true); });
}

int osmt_render_batch_labels(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgba,
                             size_t stride) {
    return guarded([&] { return osmt_render_batch_labels_body(ctx, batch, labels, out_rgba, stride); });
}

/* ---- PNG files straight from the GPU -------------------------------------- */
static uint32_t ihdr_crc(uint32_t W, uint32_t H) {
    /* CRC-32 of "IHDR" + width, height, 8, 2, 0, 0, 0 (bitwise; once per call) */
    uint8_t b[17] = {'I', 'H', 'D', 'R', (uint8_t)(W >> 24), (uint8_t)(W >> 16), (uint8_t)(W >> 8), (uint8_t)W,
                     (uint8_t)(H >> 24), (uint8_t)(H >> 16), (uint8_t)(H >> 8), (uint8_t)H, 8, 2, 0, 0, 0};
    uint32_t c = 0xFFFFFFFFu;
    for (uint8_t v : b) {
        c ^= v;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    }
    return ~c;
}

size_t osmt_png_device_bound(uint32_t W, uint32_t H) {
    /* 43 header bytes + one deflate block (its constant header, then at most PNG_LMAX bits per filtered byte, EOB) +
     * Adler, CRC, IEND */
    const size_t bits = PNG_BLOCK_HDR_BITS + (size_t)H * (3 * (size_t)W + 1) * PNG_LMAX + PNG_LMAX;
    /* + slack: the fast kernel stages its eight row bands in the slot behind the header words, each rounded up to words (3 words per band) */
    return align_up(43 + (bits + 7) / 8 + 4 + 4 + 12 + 8 + 4 * PNG_HEAD_WORDS + 128, 256);
}

static int osmt_encode_png_device_body(osmt_ctx* ctx, const void* d_rgba, size_t tile_stride, uint32_t n, uint32_t W, uint32_t H, void* d_png,
                           size_t png_stride, uint32_t* d_len, void* stream) {
    if (!ctx) return fail(OSMT_INVALID_ARG, "NULL argument");
    if (n == 0) return OSMT_OK;
    if (!d_rgba || !d_png || !d_len) return fail(OSMT_INVALID_ARG, "NULL device pointer");
    if (W == 0 || H == 0 || W > 1024 || (W % 4) != 0) return fail(OSMT_UNSUPPORTED, "PNG width %u not supported (multiple of 4, <= 1024)", W);
    if (tile_stride < (size_t)W * H * 4 || (tile_stride % 4) != 0) return fail(OSMT_INVALID_ARG, "tile stride too small / not a multiple of 4");
    if (png_stride < osmt_png_device_bound(W, H) || (png_stride % 4) != 0) return fail(OSMT_INVALID_ARG, "png_stride < osmt_png_device_bound()");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(osmt_launch_png(d_rgba, tile_stride, n, W, H, ihdr_crc(W, H), d_png, png_stride, d_len, (hipStream_t)stream));
    return OSMT_OK;
}

int osmt_encode_png_device(osmt_ctx* ctx, const void* d_rgba, size_t tile_stride, uint32_t n, uint32_t W, uint32_t H, void* d_png,
                           size_t png_stride, uint32_t* d_len, void* stream) {
    return guarded([&] { return osmt_encode_png_device_body(ctx, d_rgba, tile_stride, n, W, H, d_png, png_stride, d_len, stream); });
}

/* One call, its own pipeline.  The pre-pass runs once for the whole batch; then the tiles go through raster -> PNG
 * encode -> file lengths in CHUNKS on the call's stream, all enqueued up front.  The host walks the chunks behind the
 * GPU: as soon as a chunk's lengths are back it sums the file offsets and queues that chunk's compaction and read-back
 * on a second stream — the PCIe transfer of chunk c runs under the kernels of chunk c + 1. */
/* One batch on its way through the PNG pipeline: everything osmt_render_batch_png_begin queued and what
 * osmt_render_batch_png_end needs to finish it. */
struct osmt_png_job {
    osmt_ctx* ctx = nullptr;
    hipStream_t st = nullptr, s_c = nullptr; /* the job's own stream; the compaction / read-back stream (more than one chunk) */
    osmt_scene* sc = nullptr;
    char* d = nullptr;         /* framebuffers of two chunks, PNG slots, lengths, offsets */
    char* d_blob = nullptr;    /* compacted files, ONLY when a chunk's dead framebuffers cannot take them (see png_end_body) */
    uint32_t* h_len = nullptr; /* pinned: lengths come back asynchronously, offsets go out */
    std::vector<hipEvent_t> ev;
    uint32_t n = 0, W = 0, chunk = 0, n_chunks = 0;
    size_t slot = 0, tile_bytes = 0, o_rgba = 0, o_png = 0, o_len = 0, o_off = 0;
    int rc = OSMT_OK; /* a failure of the first half, reported by the second */
    std::string err;
    const osmt_batch* batch = nullptr; /* the caller's arrays (valid until _end returns): a missed arena guess renders them again */
    const osmt_label_batch* labels = nullptr;
};

static void png_job_release(osmt_png_job* j) {
    if (!j) return;
    osmt_ctx* ctx = j->ctx;
    for (hipEvent_t v : j->ev)
        if (v) (void)hipEventDestroy(v);
    if (j->sc) osmt_scene_free(j->sc); /* waits for the job's stream */
    dev_free(ctx, j->d);
    dev_free(ctx, j->d_blob);
    if (j->h_len) stage_release(ctx, j->h_len);
    if (j->s_c) stream_release(ctx, j->s_c);
    if (j->st) stream_release(ctx, j->st);
    delete j;
    /* the job's own reference (png_begin_body), dropped LAST: a job that outlives osmt_destroy — ended or dropped after the
     * caller closed the context — would otherwise free its scene, with it the last reference, and then hand buffers and streams
     * back to a context that ctx_teardown has deleted */
    ctx_release(ctx);
}

/* First half.  The pre-pass runs once for the whole batch; then the tiles go through raster -> PNG encode -> file lengths
 * in CHUNKS on the job's stream, all enqueued here; nothing is waited for. */
static int png_begin_body(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, osmt_png_job** out_job, bool allow_guess = true) {
    if (!ctx || !out_job) return fail(OSMT_INVALID_ARG, "NULL argument");
    *out_job = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    osmt_png_job* j = new (std::nothrow) osmt_png_job();
    if (!j) return fail(OSMT_OOM, "out of host memory");
    j->ctx = ctx;
    ctx->refs.fetch_add(1); /* like a scene: the context outlives its jobs (released at the end of png_job_release) */
    hipError_t e = stream_acquire(ctx, &j->st); /* the whole job lives on its own stream */
    if (e != hipSuccess) {
        delete j;
        ctx_release(ctx);
        return fail(OSMT_HIP_ERROR, "stream: %s", hipGetErrorString(e));
    }
    j->batch = batch;
    j->labels = labels;
    int rc = scene_upload_impl(ctx, batch, &j->sc, j->st, false, allow_guess);
    if (rc == OSMT_OK && labels) rc = osmt_scene_set_labels(ctx, j->sc, labels);
    const uint32_t n = j->n = batch ? (uint32_t)batch->n_jobs : 0u;
    if (rc == OSMT_OK && n) {
        const uint32_t W = j->W = OSMT_TILE_SIZE * batch->scale;
        const size_t tile_bytes = (size_t)W * W * 4, slot = j->slot = osmt_png_device_bound(W, W);
        /* the encoder is one workgroup per tile: below ~2 workgroups per CU it is latency-bound and a chunk costs more than
         * its overlap brings (measured on 1024 tiles: 1 chunk 4.21 ms, 2 chunks 4.10, 4 chunks 4.82, 8 chunks 7.07) */
        const uint32_t min_chunk = std::max<uint32_t>(1u, 512u / (batch->scale * batch->scale));
        uint32_t chunk = n;
        if (n >= 2u * min_chunk) chunk = std::max<uint32_t>(min_chunk, (n + 1u) / 2u);
        /* OSMT_PNG_CHUNKS (diagnostic: force the number of chunks): read ONCE — getenv on every call from worker threads is
         * not safe against a host that calls setenv — and clamped, so that a large value cannot turn one call into n events,
         * n tiny encode launches and n copies */
        static const int forced_chunks = [] {
            const char* v = getenv("OSMT_PNG_CHUNKS");
            return v ? std::min(std::max(atoi(v), 1), 16) : 0;
        }();
        if (forced_chunks) chunk = std::max<uint32_t>(1u, (n + (uint32_t)forced_chunks - 1u) / (uint32_t)forced_chunks);
        j->chunk = chunk;
        const uint32_t n_chunks = j->n_chunks = (n + chunk - 1u) / chunk;
        size_t off = 0;
        auto carve = [&](size_t bytes) {
            const size_t o = off;
            off = align_up(off + bytes, 256);
            return o;
        };
        const size_t o_rgba = j->o_rgba = carve((size_t)std::min<uint32_t>(2u, n_chunks) * chunk * tile_bytes); /* two chunks of framebuffers, alternating */
        j->tile_bytes = tile_bytes;
        j->o_png = carve((size_t)n * slot);
        j->o_len = carve((size_t)n * 4);
        j->o_off = carve((size_t)n * 8);
        e = dev_alloc(ctx, (void**)&j->d, off);
        if (e != hipSuccess) rc = fail(OSMT_OOM, "hipMalloc(%zu) failed: %s", off, hipGetErrorString(e));
        if (rc == OSMT_OK) {
            j->h_len = (uint32_t*)stage_acquire(ctx, (size_t)n * 12 + 16);
            if (!j->h_len) rc = fail(OSMT_OOM, "pinned staging for %u file lengths", n);
        }
        if (rc == OSMT_OK && n_chunks > 1u) {
            e = stream_acquire(ctx, &j->s_c);
            if (e != hipSuccess) rc = fail(OSMT_HIP_ERROR, "stream: %s", hipGetErrorString(e));
        }
        j->ev.assign(n_chunks, nullptr);
        for (uint32_t c = 0; rc == OSMT_OK && c < n_chunks; ++c) {
            e = hipEventCreateWithFlags(&j->ev[c], hipEventDisableTiming);
            if (e != hipSuccess) rc = fail(OSMT_HIP_ERROR, "event: %s", hipGetErrorString(e));
        }
        /* ---- everything the GPU has to do, queued at once ---- */
        if (rc == OSMT_OK) rc = render_impl(ctx, j->sc, 1u | 2u | 8u, nullptr, tile_bytes, false, j->st);
        for (uint32_t c = 0; rc == OSMT_OK && c < n_chunks; ++c) {
            const uint32_t first = c * chunk, cnt = std::min(chunk, n - first);
            char* rgba = j->d + o_rgba + (size_t)(c & 1u) * chunk * tile_bytes;
            rc = render_impl(ctx, j->sc, 4u | 16u, rgba, tile_bytes, false, j->st, first, cnt);
            if (rc == OSMT_OK)
                rc = osmt_encode_png_device(ctx, rgba, tile_bytes, cnt, W, W, j->d + j->o_png + (size_t)first * slot, slot,
                                            (uint32_t*)(j->d + j->o_len) + first, j->st);
            if (rc == OSMT_OK) {
                e = hipMemcpyAsync(j->h_len + first, j->d + j->o_len + (size_t)first * 4, (size_t)cnt * 4, hipMemcpyDeviceToHost, j->st);
                if (e == hipSuccess) e = hipEventRecord(j->ev[c], j->st);
                if (e != hipSuccess) rc = fail(OSMT_HIP_ERROR, "PNG pipeline: %s", hipGetErrorString(e));
            }
        }
    }
    if (rc != OSMT_OK) {
        const std::string msg = osmt_last_error();
        png_job_release(j);
        return fail(rc, "%s", msg.c_str());
    }
    *out_job = j;
    return OSMT_OK;
}

/* Second half.  The host walks the chunks behind the GPU: as soon as a chunk's lengths are back it sums the file
 * offsets and queues that chunk's compaction and read-back on a second stream — the PCIe transfer of chunk c runs under
 * the kernels of chunk c + 1 (and under the kernels of the NEXT job, when the caller has already begun one).  Releases
 * the job whatever happens. */
static int png_end_body(osmt_png_job* j, uint8_t* out_png, size_t out_capacity, uint64_t* out_off) {
    if (!j) return fail(OSMT_INVALID_ARG, "NULL job");
    if (!out_off || (!out_png && out_capacity)) {
        png_job_release(j);
        return fail(OSMT_INVALID_ARG, "NULL argument");
    }
    osmt_ctx* ctx = j->ctx;
    int rc = OSMT_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) rc = fail(OSMT_HIP_ERROR, "hipSetDevice: %s", hipGetErrorString(e));
    const uint32_t n = j->n, chunk = j->chunk;
    std::vector<unsigned long long> offs((size_t)n + 1, 0ull);
    if (rc == OSMT_OK && n) {
        unsigned long long* const h_off = reinterpret_cast<unsigned long long*>(j->h_len + ((n + 1u) & ~1u)); /* chunk-relative offsets */
        hipStream_t s_copy = j->s_c ? j->s_c : j->st;
        char* const d = j->d;
        const size_t slot = j->slot;
        bool fits = true;
        for (uint32_t c = 0; rc == OSMT_OK && c < j->n_chunks; ++c) {
            const uint32_t first = c * chunk, cnt = std::min(chunk, n - first);
            e = hipEventSynchronize(j->ev[c]);
            if (e != hipSuccess) {
                rc = fail(OSMT_HIP_ERROR, "PNG pipeline: %s", hipGetErrorString(e));
                break;
            }
            for (uint32_t i = 0; i < cnt; ++i) {
                h_off[first + i] = offs[first + i] - offs[first];
                offs[first + i + 1] = offs[first + i] + j->h_len[first + i];
            }
            const size_t c_bytes = (size_t)(offs[first + cnt] - offs[first]);
            if (offs[first + cnt] > out_capacity) fits = false; /* keep summing: the caller learns the size it needs */
            if (!fits || c_bytes == 0) continue;
            /* The compacted files of a chunk go where its framebuffers were: they are dead once the chunk is encoded (the
             * event above), each chunk has its own as long as there are at most two, and map tiles compress to a fifth of
             * them.  Only files that would not fit (noise: the slot bound is 12 bits per byte) or a forced chunk count above
             * two (framebuffers re-used by chunk c + 2) get a buffer of their own. */
            char* blob = d + j->o_rgba + (size_t)(c & 1u) * chunk * j->tile_bytes;
            static const bool own_blob = [] { /* diagnostic: OSMT_PNG_OWN_BLOB=1 always compacts into a buffer of its own */
                const char* v = getenv("OSMT_PNG_OWN_BLOB");
                return v && atoi(v) != 0;
            }();
            if (own_blob || j->n_chunks > 2u || c_bytes > (size_t)cnt * j->tile_bytes) {
                if (!j->d_blob) {
                    e = dev_alloc(ctx, (void**)&j->d_blob, (size_t)n * slot);
                    if (e != hipSuccess) {
                        rc = fail(OSMT_OOM, "hipMalloc(%zu) failed: %s", (size_t)n * slot, hipGetErrorString(e));
                        break;
                    }
                }
                blob = j->d_blob + (size_t)first * slot;
            }
            e = hipMemcpyAsync(d + j->o_off + (size_t)first * 8, h_off + first, (size_t)cnt * 8, hipMemcpyHostToDevice, s_copy);
            if (e == hipSuccess)
                e = osmt_launch_png_compact(d + j->o_png + (size_t)first * slot, slot, (const uint32_t*)(d + j->o_len) + first,
                                            (const unsigned long long*)(d + j->o_off) + first, cnt, blob, s_copy);
            if (e == hipSuccess) e = hipMemcpyAsync(out_png + offs[first], blob, c_bytes, hipMemcpyDeviceToHost, s_copy);
            if (e != hipSuccess) rc = fail(OSMT_HIP_ERROR, "PNG readback failed: %s", hipGetErrorString(e));
        }
        if (j->s_c) {
            e = hipStreamSynchronize(j->s_c);
            if (e != hipSuccess && rc == OSMT_OK) rc = fail(OSMT_HIP_ERROR, "PNG readback failed: %s", hipGetErrorString(e));
        }
        e = hipStreamSynchronize(j->st);
        if (e != hipSuccess && rc == OSMT_OK) rc = fail(OSMT_HIP_ERROR, "PNG pipeline: %s", hipGetErrorString(e));
        if (rc == OSMT_OK && !fits) rc = fail(OSMT_INVALID_ARG, "out_capacity %zu < %llu bytes of PNG data", out_capacity, offs[n]);
    }
    for (uint32_t i = 0; i <= n; ++i) out_off[i] = offs[i];
    g_arena_guess_missed = false;
    if (rc == OSMT_OK) rc = label_error_check(j->sc, j->st);
    const std::string msg = rc != OSMT_OK ? std::string(osmt_last_error()) : std::string();
    const bool missed = rc != OSMT_OK && g_arena_guess_missed;
    const osmt_batch* batch = j->batch;
    const osmt_label_batch* labels = j->labels;
    png_job_release(j);
    if (missed) { /* the guessed arenas were too small for this batch: the whole job again with exact sizing */
        g_arena_guess_missed = false;
        osmt_png_job* again = nullptr;
        rc = png_begin_body(ctx, batch, labels, &again, false);
        if (rc != OSMT_OK) return rc;
        return png_end_body(again, out_png, out_capacity, out_off);
    }
    return rc != OSMT_OK ? fail(rc, "%s", msg.c_str()) : OSMT_OK;
}

int osmt_render_batch_png_begin(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, osmt_png_job** out_job) {
    return guarded([&] { return png_begin_body(ctx, batch, labels, out_job); });
}

int osmt_render_batch_png_end(osmt_png_job* job, uint8_t* out_png, size_t out_capacity, uint64_t* out_off) {
    return guarded([&] { return png_end_body(job, out_png, out_capacity, out_off); });
}

/* One call = its own pipeline: begin + end. */
int osmt_render_batch_png(osmt_ctx* ctx, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_png, size_t out_capacity,
                          uint64_t* out_off) {
    return guarded([&] {
        if (!ctx || !out_off || (!out_png && out_capacity)) return fail(OSMT_INVALID_ARG, "NULL argument");
        osmt_png_job* j = nullptr;
        const int rc = png_begin_body(ctx, batch, labels, &j);
        if (rc != OSMT_OK) return rc;
        return png_end_body(j, out_png, out_capacity, out_off);
    });
}

static int osmt_host_alloc_body(osmt_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return fail(OSMT_INVALID_ARG, "NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return OSMT_OK;
}

int osmt_host_alloc(osmt_ctx* ctx, size_t bytes, void** out) {
    return guarded([&] { return osmt_host_alloc_body(ctx, bytes, out); });
}

void osmt_host_free(osmt_ctx* ctx, void* p) {
    if (!ctx || !p) return;
    (void)hipSetDevice(ctx->device);
    (void)hipHostFree(p);
}

static int osmt_project_body(osmt_ctx* ctx, const double* latlon, size_t n, uint8_t zoom, uint32_t tx, uint32_t ty, double scale,
                 int32_t* xy) {
    if (!ctx || (n && (!latlon || !xy))) return fail(OSMT_INVALID_ARG, "NULL argument");
    if (n >= 0xFFFFFFFFull) return fail(OSMT_INVALID_ARG, "too many points");
    if (n == 0) return OSMT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    /* device buffers from the per-context cache, the whole call on a pooled private stream: concurrent worker threads
     * neither pay hipMalloc/hipFree per call nor wait for each other's work */
    char* d = nullptr;
    hipStream_t st = nullptr;
    HIP_TRY(stream_acquire(ctx, &st));
    const size_t in_bytes = align_up(n * 16, 256);
    hipError_t e = dev_alloc(ctx, (void**)&d, in_bytes + n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d, latlon, n * 16, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = osmt_launch_project_single((const double*)d, (uint32_t)n, zoom, tx, ty, scale, (int32_t*)(d + in_bytes), st);
    if (e == hipSuccess) e = hipMemcpyAsync(xy, d + in_bytes, n * 8, hipMemcpyDeviceToHost, st);
    const hipError_t es = hipStreamSynchronize(st); /* also on the error path: the buffer goes back to the cache */
    if (e == hipSuccess) e = es;
    dev_free(ctx, d);
    stream_release(ctx, st);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "osmt_project: %s", hipGetErrorString(e));
    return OSMT_OK;
}

int osmt_project(osmt_ctx* ctx, const double* latlon, size_t n, uint8_t zoom, uint32_t tx, uint32_t ty, double scale,
                 int32_t* xy) {
    return guarded([&] { return osmt_project_body(ctx, latlon, n, zoom, tx, ty, scale, xy); });
}

static int osmt_composite_device_body(osmt_ctx* ctx, const void* d_planes, const double canvas[4], uint32_t n, uint32_t L,
                          uint32_t W, uint32_t H, void* d_out, void* stream) {
    if (!ctx || !canvas) return fail(OSMT_INVALID_ARG, "NULL argument");
    if ((uint64_t)W * H >= 0xFFFFFFFFull) return fail(OSMT_INVALID_ARG, "tile too large");
    if (((uint64_t)W * H) % 64u) return fail(OSMT_INVALID_ARG, "W*H must be a multiple of 64 (tiles are (256*scale)^2)");
    if ((size_t)n * W * H && (!d_planes || !d_out)) return fail(OSMT_INVALID_ARG, "NULL device pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(osmt_launch_composite(d_planes, canvas, n, L, W * H, d_out, (hipStream_t)stream));
    return OSMT_OK;
}

int osmt_composite_device(osmt_ctx* ctx, const void* d_planes, const double canvas[4], uint32_t n, uint32_t L,
                          uint32_t W, uint32_t H, void* d_out, void* stream) {
    return guarded([&] { return osmt_composite_device_body(ctx, d_planes, canvas, n, L, W, H, d_out, stream); });
}

static int osmt_composite_body(osmt_ctx* ctx, const double* planes, const double canvas[4], uint32_t n, uint32_t L, uint32_t W,
                   uint32_t H, uint8_t* out_rgba) {
    if (!ctx || !canvas) return fail(OSMT_INVALID_ARG, "NULL argument");
    const size_t npx = (size_t)n * W * H;
    if (npx == 0) return OSMT_OK;
    if (!planes || !out_rgba) return fail(OSMT_INVALID_ARG, "NULL buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    char* d = nullptr;
    hipStream_t st = nullptr;
    HIP_TRY(stream_acquire(ctx, &st));
    const size_t in_bytes = align_up(npx * L * 32, 256);
    hipError_t e = dev_alloc(ctx, (void**)&d, in_bytes + npx * 4);
    if (e == hipSuccess && L) e = hipMemcpyAsync(d, planes, npx * L * 32, hipMemcpyHostToDevice, st);
    int rc = OSMT_OK;
    if (e == hipSuccess) rc = osmt_composite_device(ctx, d, canvas, n, L, W, H, d + in_bytes, st);
    if (e == hipSuccess && rc == OSMT_OK) e = hipMemcpyAsync(out_rgba, d + in_bytes, npx * 4, hipMemcpyDeviceToHost, st);
    const hipError_t es = hipStreamSynchronize(st);
    if (e == hipSuccess) e = es;
    dev_free(ctx, d);
    stream_release(ctx, st);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "osmt_composite: %s", hipGetErrorString(e));
    return rc;
}

int osmt_composite(osmt_ctx* ctx, const double* planes, const double canvas[4], uint32_t n, uint32_t L, uint32_t W,
                   uint32_t H, uint8_t* out_rgba) {
    return guarded([&] { return osmt_composite_body(ctx, planes, canvas, n, L, W, H, out_rgba); });
}

} /* extern "C" */

/* ======================= one node, several GPUs ======================= */
struct osmt_batch_shard {
    osmt_batch b;
    std::vector<osmt_tile_job> jobs;
    std::vector<osmt_op> ops;
    std::vector<osmt_ring> rings;
    std::vector<double> latlon;
    std::vector<int32_t> points;
    std::vector<uint32_t> node_refs;
    std::vector<double> dashes;
};

namespace {

/* RCCL entry points, resolved at the first use.  dlopen instead of a link-time dependency: inside a PyTorch process
 * the HIP runtime in use is PyTorch's own copy, and its own RCCL (already loaded) is the one built against it. */
typedef struct { char internal[OSMT_COMM_ID_BYTES]; } rccl_unique_id;
struct rccl_api {
    void* handle = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(void**, int, rccl_unique_id, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
constexpr int RCCL_UINT64 = 5; /* ncclUint64 */
constexpr int RCCL_SUM = 0;    /* ncclSum */

rccl_api* rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) { /* a copy that is already mapped wins */
            api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (api.handle) break;
        }
        for (const char* n : names) {
            if (api.handle) break;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!api.handle) {
            const char* e = dlerror();
            api.error = std::string("librccl could not be loaded: ") + (e ? e : "not found");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(api.handle, name);
            if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + name;
            return p;
        };
        api.GetUniqueId = (int (*)(rccl_unique_id*))sym("ncclGetUniqueId");
        api.CommInitRank = (int (*)(void**, int, rccl_unique_id, int))sym("ncclCommInitRank");
        api.CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
        api.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        api.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
        api.GroupStart = (int (*)())sym("ncclGroupStart");
        api.GroupEnd = (int (*)())sym("ncclGroupEnd");
        api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    });
    return &api;
}

int rccl_fail(const char* what, int rc) {
    rccl_api* r = rccl();
    return fail(OSMT_RCCL_ERROR, "%s failed: %s", what, (r->GetErrorString && rc) ? r->GetErrorString(rc) : (r->error.empty() ? "unknown" : r->error.c_str()));
}

#define RCCL_TRY(what, expr)                        \
    do {                                            \
        const int _rc = (expr);                     \
        if (_rc != 0) return rccl_fail(what, _rc);  \
    } while (0)

int rccl_ready() {
    rccl_api* r = rccl();
    if (!r->handle || !r->error.empty()) return fail(OSMT_RCCL_ERROR, "%s", r->error.empty() ? "librccl not available" : r->error.c_str());
    return OSMT_OK;
}

int comm_buffers(osmt_ctx* ctx) {
    if (ctx->d_count) return OSMT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc((void**)&ctx->d_count, 4 * sizeof(unsigned long long)));
    return OSMT_OK;
}

}  // namespace

namespace {
void comm_destroy(osmt_ctx* ctx) {
    if (ctx->comm) {
        rccl_api* r = rccl();
        if (r->CommDestroy) (void)r->CommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    if (ctx->d_count) (void)hipFree(ctx->d_count);
    ctx->d_count = nullptr;
}
}  // namespace

extern "C" {

static int shard_create_body(const osmt_batch* b, uint32_t rank, uint32_t world, osmt_batch_shard** out, bool trusted = false) {
    if (!out) return fail(OSMT_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (world == 0 || rank >= world) return fail(OSMT_INVALID_ARG, "rank %u not in 0..%u", rank, world);
    int rc = trusted ? OSMT_OK : validate_batch(b);
    if (rc != OSMT_OK) return rc;
    osmt_batch_shard* s = new (std::nothrow) osmt_batch_shard();
    if (!s) return fail(OSMT_OOM, "out of host memory");
    const bool ll = b->coord_kind == OSMT_COORD_LATLON_F64, nr = b->coord_kind == OSMT_COORD_NODE_REF;
    for (size_t j = rank; j < b->n_jobs; j += world) { /* tile i -> shard i mod world (http_server.rs:105-108) */
        osmt_tile_job job = b->jobs[j];
        const uint32_t old_pt = job.pt_off, new_pt = (uint32_t)(ll ? s->latlon.size() / 2 : nr ? s->node_refs.size() : s->points.size() / 2);
        const uint32_t old_op = job.op_off;
        job.op_off = (uint32_t)s->ops.size();
        job.pt_off = new_pt;
        for (uint32_t k = 0; k < job.n_ops; ++k) {
            osmt_op op = b->ops[old_op + k];
            if (op.kind != OSMT_OP_NONE) {
                const uint32_t old_ring = op.ring_off;
                op.ring_off = (uint32_t)s->rings.size();
                for (uint32_t r = 0; r < op.n_rings; ++r) {
                    osmt_ring ring = b->rings[old_ring + r];
                    ring.first_pt = ring.first_pt - old_pt + new_pt;
                    s->rings.push_back(ring);
                }
                if (op.kind == OSMT_OP_STROKE && op.has_dashes) {
                    const uint32_t old_d = op.dashes_off;
                    op.dashes_off = (uint32_t)s->dashes.size();
                    s->dashes.insert(s->dashes.end(), b->dashes + old_d, b->dashes + old_d + op.n_dashes);
                }
            }
            s->ops.push_back(op);
        }
        if (ll) s->latlon.insert(s->latlon.end(), b->latlon + 2 * (size_t)old_pt, b->latlon + 2 * ((size_t)old_pt + job.n_pts));
        else if (nr) s->node_refs.insert(s->node_refs.end(), b->node_refs + old_pt, b->node_refs + old_pt + job.n_pts);
        else s->points.insert(s->points.end(), b->points + 2 * (size_t)old_pt, b->points + 2 * ((size_t)old_pt + job.n_pts));
        s->jobs.push_back(job);
    }
    memset(&s->b, 0, sizeof s->b);
    s->b.jobs = s->jobs.data();
    s->b.n_jobs = s->jobs.size();
    s->b.ops = s->ops.data();
    s->b.n_ops = s->ops.size();
    s->b.rings = s->rings.data();
    s->b.n_rings = s->rings.size();
    s->b.coord_kind = b->coord_kind;
    s->b.scale = b->scale;
    s->b.latlon = ll ? s->latlon.data() : nullptr;
    s->b.points = (!ll && !nr) ? s->points.data() : nullptr;
    s->b.n_pts = ll ? s->latlon.size() / 2 : nr ? s->node_refs.size() : s->points.size() / 2;
    s->b.dashes = s->dashes.empty() ? nullptr : s->dashes.data();
    s->b.n_dashes = s->dashes.size();
    s->b.nodes = nr ? b->nodes : nullptr;
    s->b.n_nodes = nr ? b->n_nodes : 0;
    s->b.node_refs = nr ? s->node_refs.data() : nullptr;
    *out = s;
    return OSMT_OK;
}

int osmt_batch_shard_create(const osmt_batch* b, uint32_t rank, uint32_t world, osmt_batch_shard** out) {
    return guarded([&] { return shard_create_body(b, rank, world, out); });
}

const osmt_batch* osmt_batch_shard_get(const osmt_batch_shard* s) { return s ? &s->b : nullptr; }

void osmt_batch_shard_free(osmt_batch_shard* s) { delete s; }

static int comm_unique_id_body(uint8_t* id) {
    if (!id) return fail(OSMT_INVALID_ARG, "NULL argument");
    int rc = rccl_ready();
    if (rc != OSMT_OK) return rc;
    rccl_unique_id u;
    RCCL_TRY("ncclGetUniqueId", rccl()->GetUniqueId(&u));
    memcpy(id, u.internal, OSMT_COMM_ID_BYTES);
    return OSMT_OK;
}

int osmt_comm_unique_id(uint8_t id[OSMT_COMM_ID_BYTES]) {
    return guarded([&] { return comm_unique_id_body(id); });
}

static int comm_init_rank_body(osmt_ctx* ctx, const uint8_t* id, uint32_t rank, uint32_t nranks) {
    if (!ctx || !id) return fail(OSMT_INVALID_ARG, "NULL argument");
    if (nranks == 0 || rank >= nranks) return fail(OSMT_INVALID_ARG, "rank %u not in 0..%u", rank, nranks);
    int rc = rccl_ready();
    if (rc != OSMT_OK) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    comm_destroy(ctx);
    rc = comm_buffers(ctx);
    if (rc != OSMT_OK) return rc;
    rccl_unique_id u;
    memcpy(u.internal, id, OSMT_COMM_ID_BYTES);
    RCCL_TRY("ncclCommInitRank", rccl()->CommInitRank(&ctx->comm, (int)nranks, u, (int)rank));
    ctx->comm_rank = rank;
    ctx->comm_size = nranks;
    return OSMT_OK;
}

int osmt_comm_init_rank(osmt_ctx* ctx, const uint8_t id[OSMT_COMM_ID_BYTES], uint32_t rank, uint32_t nranks) {
    return guarded([&] { return comm_init_rank_body(ctx, id, rank, nranks); });
}

static int comm_init_local_body(osmt_ctx* const* ctxs, uint32_t n) {
    if (!ctxs || n == 0) return fail(OSMT_INVALID_ARG, "no contexts");
    int rc = rccl_ready();
    if (rc != OSMT_OK) return rc;
    std::vector<int> devs(n);
    std::vector<void*> comms(n, nullptr);
    for (uint32_t i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(OSMT_INVALID_ARG, "context %u is NULL", i);
        for (uint32_t k = 0; k < i; ++k)
            if (ctxs[k]->device == ctxs[i]->device) return fail(OSMT_INVALID_ARG, "contexts %u and %u share device %d", k, i, ctxs[i]->device);
        devs[i] = ctxs[i]->device;
        comm_destroy(ctxs[i]);
        rc = comm_buffers(ctxs[i]);
        if (rc != OSMT_OK) return rc;
    }
    RCCL_TRY("ncclCommInitAll", rccl()->CommInitAll(comms.data(), (int)n, devs.data()));
    for (uint32_t i = 0; i < n; ++i) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_size = n;
    }
    return OSMT_OK;
}

int osmt_comm_init_local(osmt_ctx* const* ctxs, uint32_t n) {
    return guarded([&] { return comm_init_local_body(ctxs, n); });
}

static int allreduce_local_body(osmt_ctx* const* ctxs, uint32_t n, const uint64_t* locals, uint64_t* out) {
    if (!ctxs || !locals || !out || n == 0) return fail(OSMT_INVALID_ARG, "NULL argument");
    for (uint32_t i = 0; i < n; ++i)
        if (!ctxs[i] || !ctxs[i]->comm || ctxs[i]->comm_size != n) return fail(OSMT_INVALID_ARG, "context %u has no communicator of size %u (osmt_comm_init_local)", i, n);
    rccl_api* r = rccl();
    std::vector<hipStream_t> st(n, nullptr);
    int rc = OSMT_OK;
    for (uint32_t i = 0; i < n && rc == OSMT_OK; ++i) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess || stream_acquire(ctxs[i], &st[i]) != hipSuccess ||
            hipMemcpyAsync(ctxs[i]->d_count, &locals[i], 8, hipMemcpyHostToDevice, st[i]) != hipSuccess)
            rc = fail(OSMT_HIP_ERROR, "tile-count staging failed on device %d", ctxs[i]->device);
    }
    if (rc == OSMT_OK) {
        int e = r->GroupStart();
        for (uint32_t i = 0; i < n && e == 0; ++i) {
            (void)hipSetDevice(ctxs[i]->device);
            e = r->AllReduce(ctxs[i]->d_count, ctxs[i]->d_count + 1, 1, RCCL_UINT64, RCCL_SUM, ctxs[i]->comm, st[i]);
        }
        const int e2 = r->GroupEnd();
        if (e != 0 || e2 != 0) rc = rccl_fail("ncclAllReduce", e ? e : e2);
    }
    unsigned long long got = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!st[i]) continue;
        (void)hipSetDevice(ctxs[i]->device);
        if (rc == OSMT_OK && i == 0 && hipMemcpyAsync(&got, ctxs[0]->d_count + 1, 8, hipMemcpyDeviceToHost, st[0]) != hipSuccess)
            rc = fail(OSMT_HIP_ERROR, "tile-count read-back failed");
        if (hipStreamSynchronize(st[i]) != hipSuccess && rc == OSMT_OK) rc = fail(OSMT_HIP_ERROR, "tile-count reduction failed on device %d", ctxs[i]->device);
        stream_release(ctxs[i], st[i]);
    }
    if (rc == OSMT_OK) *out = got;
    return rc;
}

int osmt_allreduce_tile_count_local(osmt_ctx* const* ctxs, uint32_t n, const uint64_t* locals, uint64_t* out) {
    return guarded([&] { return allreduce_local_body(ctxs, n, locals, out); });
}

static int allreduce_body(osmt_ctx* ctx, uint64_t local, uint64_t* out) {
    if (!ctx || !out) return fail(OSMT_INVALID_ARG, "NULL argument");
    if (!ctx->comm) return fail(OSMT_INVALID_ARG, "the context has no communicator (osmt_comm_init_rank / osmt_comm_init_local)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    HIP_TRY(stream_acquire(ctx, &st));
    int rc = OSMT_OK;
    unsigned long long got = 0;
    hipError_t e = hipMemcpyAsync(ctx->d_count, &local, 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const int n = rccl()->AllReduce(ctx->d_count, ctx->d_count + 1, 1, RCCL_UINT64, RCCL_SUM, ctx->comm, st);
        if (n != 0) rc = rccl_fail("ncclAllReduce", n);
    }
    if (e == hipSuccess && rc == OSMT_OK) e = hipMemcpyAsync(&got, ctx->d_count + 1, 8, hipMemcpyDeviceToHost, st);
    const hipError_t es = hipStreamSynchronize(st);
    stream_release(ctx, st);
    if (rc == OSMT_OK && (e != hipSuccess || es != hipSuccess)) rc = fail(OSMT_HIP_ERROR, "tile-count reduction: %s", hipGetErrorString(e != hipSuccess ? e : es));
    if (rc == OSMT_OK) *out = got;
    return rc;
}

int osmt_allreduce_tile_count(osmt_ctx* ctx, uint64_t local, uint64_t* out) {
    return guarded([&] { return allreduce_body(ctx, local, out); });
}

static int allreduce_enqueue_body(osmt_ctx* ctx, uint64_t local, void* stream) {
    if (!ctx) return fail(OSMT_INVALID_ARG, "NULL argument");
    if (!ctx->comm) return fail(OSMT_INVALID_ARG, "the context has no communicator (osmt_comm_init_rank / osmt_comm_init_local)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    /* the addend goes in as two 32-bit fills: stream-ordered, no host buffer that would have to outlive the call */
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(ctx->d_count + 2), (int)(uint32_t)local, 1, st));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)((char*)(ctx->d_count + 2) + 4), (int)(uint32_t)(local >> 32), 1, st));
    const int n = rccl()->AllReduce(ctx->d_count + 2, ctx->d_count + 3, 1, RCCL_UINT64, RCCL_SUM, ctx->comm, st);
    if (n != 0) return rccl_fail("ncclAllReduce", n);
    return OSMT_OK;
}

int osmt_allreduce_tile_count_enqueue(osmt_ctx* ctx, uint64_t local, void* stream) {
    return guarded([&] { return allreduce_enqueue_body(ctx, local, stream); });
}

static int allreduce_result_body(osmt_ctx* ctx, void* stream, uint64_t* out) {
    if (!ctx || !out) return fail(OSMT_INVALID_ARG, "NULL argument");
    if (!ctx->comm) return fail(OSMT_INVALID_ARG, "the context has no communicator (osmt_comm_init_rank / osmt_comm_init_local)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    unsigned long long got = 0;
    HIP_TRY(hipMemcpyAsync(&got, ctx->d_count + 3, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *out = got;
    return OSMT_OK;
}

int osmt_allreduce_tile_count_result(osmt_ctx* ctx, void* stream, uint64_t* out) {
    return guarded([&] { return allreduce_result_body(ctx, stream, out); });
}

/* The labels of one shard: tiles rank, rank + world, ... keep their labels in draw order, segments re-packed. */
struct label_shard {
    osmt_label_batch b;
    std::vector<osmt_label> labels;
    std::vector<uint32_t> job_label_off;
    std::vector<double> segs;
};

static int label_shard_build(const osmt_label_batch* lb, size_t n_jobs, uint32_t rank, uint32_t world, label_shard* out) {
    if (!lb->labels || !lb->job_label_off || (lb->n_segs && !lb->segs)) return fail(OSMT_INVALID_ARG, "NULL label pool");
    if (lb->job_label_off[0] != 0 || lb->job_label_off[n_jobs] != lb->n_labels)
        return fail(OSMT_INVALID_ARG, "job_label_off must run from 0 to n_labels over n_jobs + 1 entries");
    out->job_label_off.push_back(0u);
    for (size_t j = rank; j < n_jobs; j += world) {
        const uint32_t l0 = lb->job_label_off[j], l1 = lb->job_label_off[j + 1];
        if (l0 > l1 || l1 > lb->n_labels) return fail(OSMT_INVALID_ARG, "job_label_off is not monotonic");
        for (uint32_t l = l0; l < l1; ++l) {
            osmt_label lab = lb->labels[l];
            if (lab.n_segs) {
                if ((size_t)lab.seg_off + lab.n_segs > lb->n_segs) return fail(OSMT_INVALID_ARG, "label %u: segment range out of bounds", l);
                const uint32_t old = lab.seg_off;
                lab.seg_off = (uint32_t)(out->segs.size() / 4);
                out->segs.insert(out->segs.end(), lb->segs + 4 * (size_t)old, lb->segs + 4 * ((size_t)old + lab.n_segs));
            }
            out->labels.push_back(lab);
        }
        out->job_label_off.push_back((uint32_t)out->labels.size());
    }
    out->b.labels = out->labels.data();
    out->b.n_labels = out->labels.size();
    out->b.job_label_off = out->job_label_off.data();
    out->b.segs = out->segs.empty() ? nullptr : out->segs.data();
    out->b.n_segs = out->segs.size() / 4;
    return OSMT_OK;
}

static int render_batch_multi_body(osmt_ctx* const* ctxs, uint32_t n, const osmt_batch* batch, const osmt_label_batch* labels, uint32_t flags,
                                   uint8_t* out, size_t stride, uint64_t* out_count) {
    if (!ctxs || n == 0 || !batch) return fail(OSMT_INVALID_ARG, "NULL argument");
    for (uint32_t i = 0; i < n; ++i)
        if (!ctxs[i]) return fail(OSMT_INVALID_ARG, "context %u is NULL", i);
    if (flags & ~(uint32_t)OSMT_MULTI_RGB8) return fail(OSMT_INVALID_ARG, "unknown flags 0x%x", flags);
    const bool rgb = (flags & OSMT_MULTI_RGB8) != 0;
    /* Only what needs the whole batch is checked here, serially (O(n_jobs log n_jobs)); every GPU's thread validates the jobs
     * of its own shard before it packs them.  (Round 3 validated the whole batch up front: 6 ms of one thread for the
     * 10 000-tile batch while one GPU's share of the work is ~8 ms — Amdahl capped eight GPUs near 4.9x.) */
    static const bool trace_multi = getenv("OSMT_TRACE_MULTI") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    int rc = validate_batch_global(batch);
    if (rc != OSMT_OK) return rc;
    if (batch->n_jobs && !out) return fail(OSMT_INVALID_ARG, "output pointer is NULL");
    const size_t W = (size_t)OSMT_TILE_SIZE * batch->scale;
    if (stride < W * W * (rgb ? 3 : 4)) return fail(OSMT_INVALID_ARG, rgb ? "out_tile_stride_bytes < W*H*3" : "out_tile_stride_bytes < W*H*4");
    if (labels && labels->n_labels == 0) labels = nullptr;
    /* one host thread per GPU: shard, upload, kernels and read-back of the devices run side by side; shard d writes
     * tiles d, d + n, ... = base out + d * stride with a tile pitch of n * stride */
    std::vector<int> rcs(n, OSMT_OK);
    std::vector<std::string> msgs(n);
    std::vector<uint64_t> counts(n, 0);
    struct joiner { /* a failed thread start must not leave joinable threads behind (std::terminate) */
        std::vector<std::thread> th;
        ~joiner() {
            for (auto& t : th)
                if (t.joinable()) t.join();
        }
    } pool;
    pool.th.reserve(n);
    const auto t_forked = std::chrono::steady_clock::now();
    for (uint32_t d = 0; d < n; ++d) {
        pool.th.emplace_back([&, d] {
            osmt_batch_shard* sh = nullptr;
            int r = guarded([&] {
                for (size_t j = d; j < batch->n_jobs; j += n) { /* the shard's own jobs */
                    const int v = validate_job(batch, j);
                    if (v != OSMT_OK) return v;
                }
                const size_t nn = batch->coord_kind == OSMT_COORD_NODE_REF ? batch->n_nodes : 0; /* its slice of the shared node table */
                return validate_nodes(batch, nn * d / n, nn * (d + 1) / n);
            });
            if (r == OSMT_OK) r = guarded([&] { return shard_create_body(batch, d, n, &sh, true); });
            label_shard ls;
            if (r == OSMT_OK && labels) r = guarded([&] { return label_shard_build(labels, batch->n_jobs, d, n, &ls); });
            if (r == OSMT_OK && sh->b.n_jobs)
                r = guarded([&] {
                    return osmt_render_batch_labels_body(ctxs[d], &sh->b, labels ? &ls.b : nullptr, out + (size_t)d * stride, (size_t)n * stride, rgb, true);
                });
            if (r == OSMT_OK) counts[d] = sh->b.n_jobs;
            if (r != OSMT_OK) msgs[d] = osmt_last_error(); /* thread-local: carry it to the caller's thread */
            rcs[d] = r;
            osmt_batch_shard_free(sh);
        });
    }
    for (auto& t : pool.th) t.join();
    const auto t_joined = std::chrono::steady_clock::now();
    for (uint32_t d = 0; d < n; ++d)
        if (rcs[d] != OSMT_OK) { /* lowest shard first; a failure in one shard's validation has stopped only that shard */
            /* a malformed batch must leave the output untouched by contract?  No: like osmt_render_batch, a failing call
             * leaves the buffer unspecified */
            return fail(rcs[d], "GPU %u (device %d): %s", d, ctxs[d]->device, msgs[d].c_str());
        }
    uint64_t total = 0;
    bool have_comm = n > 1;
    for (uint32_t d = 0; d < n; ++d) have_comm = have_comm && ctxs[d]->comm && ctxs[d]->comm_size == n;
    if (have_comm) {
        rc = allreduce_local_body(ctxs, n, counts.data(), &total);
        if (rc != OSMT_OK) return rc;
    } else {
        for (uint64_t c : counts) total += c;
    }
    if (total != batch->n_jobs) return fail(OSMT_HIP_ERROR, "tile count %llu != %zu jobs", (unsigned long long)total, batch->n_jobs);
    if (out_count) *out_count = total;
    if (trace_multi) { /* OSMT_TRACE_MULTI=1 (diagnostic): the serial share of the call, one line on stderr */
        const auto t_end = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
            return std::chrono::duration<double, std::micro>(b2 - a).count();
        };
        fprintf(stderr, "osmt multi: %u contexts, %zu tiles: serial head %.0f us, parallel %.0f us, serial tail %.0f us\n", n, batch->n_jobs,
                us(t_begin, t_forked), us(t_forked, t_joined), us(t_joined, t_end));
    }
    return OSMT_OK;
}

int osmt_render_batch_multi(osmt_ctx* const* ctxs, uint32_t n, const osmt_batch* batch, uint8_t* out, size_t stride, uint64_t* out_count) {
    return guarded([&] { return render_batch_multi_body(ctxs, n, batch, nullptr, 0u, out, stride, out_count); });
}

int osmt_render_batch_multi_ex(osmt_ctx* const* ctxs, uint32_t n, const osmt_batch* batch, const osmt_label_batch* labels, uint32_t flags,
                               uint8_t* out, size_t stride, uint64_t* out_count) {
    return guarded([&] { return render_batch_multi_body(ctxs, n, batch, labels, flags, out, stride, out_count); });
}

}  // extern "C"

/* ---- the per-request entry: worker handles that gather concurrent one-tile requests -------------------------------
 * The reference's server hands ONE tile per request to each of available_parallelism() worker threads, every one with
 * its own TilePixels (src/http_server.rs:50-83,105-108,134-181).  Sixteen threads that each run the whole chain — five
 * pre-pass launches, a raster launch for 128 waves, a copy, a synchronisation — for one tile queue up behind each other
 * on the device (round 3: 17 k tiles/s, p99 9.6 ms from 16 workers).  The GPU renders 16 tiles in the time of one, so
 * the requests are gathered ("group commit"): a request that finds fewer than OSMT_WORKER_INFLIGHT groups on the
 * device becomes a leader at once and takes EVERY request that is waiting (up to 64 tiles) with it; requests that
 * arrive while the device is busy wait for the next leader.  No timer: a lone request never waits for company, and
 * the group size follows the load.  The leader merges the display lists (a valid batch: op ranges partition the merged
 * pool, every request was validated by its own thread), renders them with ONE launch sequence into pinned staging and
 * every requester copies its own tiles out in parallel.  A group of one skips all of that and is osmt_render_batch_rgb. */
struct coalesce_group {
    osmt_ctx* ctx = nullptr;
    void* stage = nullptr;
    std::atomic<int> pending{0}; /* requesters that still have to copy their tiles out of `stage` */
};

struct coalesce_req {
    const osmt_batch* b = nullptr;
    const osmt_label_batch* lb = nullptr;
    uint8_t* out = nullptr;
    size_t stride = 0;
    bool taken = false, done = false, ran = false; /* ran: rendered on its own (rc is its verdict) */
    int rc = OSMT_OK;
    std::string err;
    coalesce_group* grp = nullptr; /* where the pixels are (NULL: already in `out`, or failed) */
    const uint8_t* src = nullptr;
};

struct osmt_worker {
    osmt_ctx* ctx = nullptr;
};

namespace {

constexpr size_t CO_MAX_TILES = 64;

int co_max_in_flight() {
    static const int v = [] {
        const char* e = getenv("OSMT_WORKER_INFLIGHT");
        /* 16 / 32 native threads, one-tile requests: 48 / 64 k tiles/s with 1, 45 / 72 k with 2, 35 / 53 k with 3, 28 / 49 k with 4,
         * 20 / 37 k with 6 (profiles/r04_j_worker_inflight.txt): every group in flight is a host thread in the runtime, and
         * they queue up behind each other's calls */
        return e ? std::min(std::max(atoi(e), 1), 16) : 2;
    }();
    return v;
}

/* the display lists (and label lists) of several requests as ONE batch, in request order */
struct merged_batch {
    std::vector<osmt_tile_job> jobs;
    std::vector<osmt_op> ops;
    std::vector<osmt_ring> rings;
    std::vector<double> latlon; /* per point, or the concatenated node tables */
    std::vector<int32_t> points;
    std::vector<uint32_t> node_refs;
    std::vector<double> dashes;
    std::vector<osmt_label> labels;
    std::vector<uint32_t> job_label_off;
    std::vector<double> segs;
    osmt_batch b;
    osmt_label_batch lb;
    bool has_labels = false;
};

void merge_requests(const std::vector<coalesce_req*>& reqs, merged_batch* m) {
    const osmt_batch* b0 = reqs[0]->b;
    const bool ll = b0->coord_kind == OSMT_COORD_LATLON_F64, nr = b0->coord_kind == OSMT_COORD_NODE_REF;
    for (const coalesce_req* r : reqs)
        if (r->lb && r->lb->n_labels) m->has_labels = true;
    if (m->has_labels) m->job_label_off.push_back(0u);
    for (const coalesce_req* r : reqs) {
        const osmt_batch* b = r->b;
        const uint32_t op0 = (uint32_t)m->ops.size(), ring0 = (uint32_t)m->rings.size(), dash0 = (uint32_t)m->dashes.size();
        const uint32_t pt0 = (uint32_t)(ll ? m->latlon.size() / 2 : nr ? m->node_refs.size() : m->points.size() / 2);
        const uint32_t node0 = nr ? (uint32_t)(m->latlon.size() / 2) : 0u;
        for (size_t j = 0; j < b->n_jobs; ++j) {
            osmt_tile_job job = b->jobs[j];
            job.op_off += op0;
            job.pt_off += pt0;
            m->jobs.push_back(job);
        }
        for (size_t o = 0; o < b->n_ops; ++o) {
            osmt_op op = b->ops[o];
            if (op.kind != OSMT_OP_NONE) {
                op.ring_off += ring0;
                if (op.kind == OSMT_OP_STROKE && op.has_dashes) op.dashes_off += dash0;
            }
            m->ops.push_back(op);
        }
        for (size_t k = 0; k < b->n_rings; ++k) {
            osmt_ring ring = b->rings[k];
            ring.first_pt += pt0;
            m->rings.push_back(ring);
        }
        if (ll) {
            m->latlon.insert(m->latlon.end(), b->latlon, b->latlon + 2 * b->n_pts);
        } else if (nr) {
            m->latlon.insert(m->latlon.end(), b->nodes, b->nodes + 2 * b->n_nodes);
            for (size_t i = 0; i < b->n_pts; ++i) m->node_refs.push_back(b->node_refs[i] + node0);
        } else {
            m->points.insert(m->points.end(), b->points, b->points + 2 * b->n_pts);
        }
        if (b->n_dashes) m->dashes.insert(m->dashes.end(), b->dashes, b->dashes + b->n_dashes);
        if (m->has_labels) {
            const osmt_label_batch* lb = r->lb;
            const uint32_t lab0 = (uint32_t)m->labels.size(), seg0 = (uint32_t)(m->segs.size() / 4);
            if (lb && lb->n_labels) {
                for (size_t i = 0; i < lb->n_labels; ++i) {
                    osmt_label l = lb->labels[i];
                    l.seg_off += seg0;
                    m->labels.push_back(l);
                }
                m->segs.insert(m->segs.end(), lb->segs, lb->segs + 4 * lb->n_segs);
                for (size_t j = 0; j < b->n_jobs; ++j) m->job_label_off.push_back(lab0 + lb->job_label_off[j + 1]);
            } else {
                for (size_t j = 0; j < b->n_jobs; ++j) m->job_label_off.push_back(lab0);
            }
        }
    }
    memset(&m->b, 0, sizeof m->b);
    m->b.jobs = m->jobs.data();
    m->b.n_jobs = m->jobs.size();
    m->b.ops = m->ops.data();
    m->b.n_ops = m->ops.size();
    m->b.rings = m->rings.data();
    m->b.n_rings = m->rings.size();
    m->b.coord_kind = b0->coord_kind;
    m->b.scale = b0->scale;
    m->b.n_pts = ll ? m->latlon.size() / 2 : nr ? m->node_refs.size() : m->points.size() / 2;
    m->b.latlon = ll ? m->latlon.data() : nullptr;
    m->b.points = (!ll && !nr) ? m->points.data() : nullptr;
    m->b.nodes = nr ? m->latlon.data() : nullptr;
    m->b.n_nodes = nr ? m->latlon.size() / 2 : 0;
    m->b.node_refs = nr ? m->node_refs.data() : nullptr;
    m->b.dashes = m->dashes.data();
    m->b.n_dashes = m->dashes.size();
    memset(&m->lb, 0, sizeof m->lb);
    m->lb.labels = m->labels.data();
    m->lb.n_labels = m->labels.size();
    m->lb.job_label_off = m->job_label_off.data();
    m->lb.segs = m->segs.data();
    m->lb.n_segs = m->segs.size() / 4;
}

/* label batches are validated where they are attached (osmt_scene_set_labels); here only what merging itself reads */
int label_batch_shape_ok(const osmt_batch* b, const osmt_label_batch* lb) {
    if (!lb || !lb->n_labels) return OSMT_OK;
    if (!lb->labels || !lb->job_label_off || (lb->n_segs && !lb->segs)) return fail(OSMT_INVALID_ARG, "label batch: NULL pool");
    if (lb->n_labels >= 0x7FFFFFFFull || lb->n_segs >= 0x7FFFFFFFull) return fail(OSMT_INVALID_ARG, "label batch too large");
    if (lb->job_label_off[0] != 0u) return fail(OSMT_INVALID_ARG, "job_label_off[0] must be 0");
    for (size_t j = 0; j < b->n_jobs; ++j)
        if (lb->job_label_off[j + 1] < lb->job_label_off[j] || lb->job_label_off[j + 1] > lb->n_labels)
            return fail(OSMT_INVALID_ARG, "job_label_off not monotonic / out of range at tile %zu", j);
    if (lb->job_label_off[b->n_jobs] != lb->n_labels) return fail(OSMT_INVALID_ARG, "job_label_off[n_jobs] != n_labels");
    /* a label's draw_line calls must lie in ITS request's pool: re-based into the merged pool an out-of-range seg_off would
     * land in another request's calls and pass the merged check instead of failing alone */
    for (size_t i = 0; i < lb->n_labels; ++i)
        if ((unsigned long long)lb->labels[i].seg_off + lb->labels[i].n_segs > lb->n_segs)
            return fail(OSMT_INVALID_ARG, "label %zu: seg_off + n_segs beyond the batch's draw_line calls", i);
    return OSMT_OK;
}

/* renders the requests of one group; fills rc / err / grp / src of every request (the caller marks them done) */
void coalesce_run_group(osmt_ctx* ctx, const std::vector<coalesce_req*>& reqs) {
    auto run_alone = [&](coalesce_req* r) {
        /* its requester has validated it: trusted, like the merged batch (no second pass over the ops and points) */
        r->rc = guarded([&] {
            return osmt_render_batch_labels_body(ctx, r->b, (r->lb && r->lb->n_labels) ? r->lb : nullptr, r->out, r->stride, true, true);
        });
        r->ran = true;
        if (r->rc != OSMT_OK) r->err = osmt_last_error();
    };
    if (reqs.size() == 1) {
        run_alone(reqs[0]);
        return;
    }
    const size_t W = (size_t)OSMT_TILE_SIZE * reqs[0]->b->scale, tile_rgb = W * W * 3;
    size_t n_tiles = 0;
    for (const coalesce_req* r : reqs) n_tiles += r->b->n_jobs;
    int rc = OSMT_OK;
    void* stage = nullptr;
    /* OSMT_TRACE_WORKER=1 (diagnostic): one stderr line per gathered group — requests, tiles, merge and render time */
    static const bool trace_worker = getenv("OSMT_TRACE_WORKER") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto t1 = t0;
    try {
        merged_batch m;
        merge_requests(reqs, &m);
        stage = stage_acquire(ctx, n_tiles * tile_rgb);
        t1 = std::chrono::steady_clock::now();
        if (!stage) {
            rc = OSMT_OOM;
        } else {
            rc = osmt_render_batch_labels_body(ctx, &m.b, m.has_labels ? &m.lb : nullptr, (uint8_t*)stage, tile_rgb, true, true);
        }
    } catch (...) {
        rc = OSMT_OOM;
    }
    if (trace_worker) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "osmt worker group: %zu requests, %zu tiles, merge+staging %.0f us, render %.0f us, rc %d\n", reqs.size(), n_tiles,
                std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count(), rc);
    }
    if (rc != OSMT_OK) {
        /* whatever went wrong (one request's labels refused, no pinned memory): every request on its own, so that a bad
         * one fails alone */
        if (stage) stage_release(ctx, stage);
        for (coalesce_req* r : reqs) run_alone(r);
        return;
    }
    coalesce_group* g = new (std::nothrow) coalesce_group();
    if (!g) {
        stage_release(ctx, stage);
        for (coalesce_req* r : reqs) run_alone(r);
        return;
    }
    g->ctx = ctx;
    g->stage = stage;
    g->pending.store((int)reqs.size());
    size_t first = 0;
    for (coalesce_req* r : reqs) {
        r->grp = g;
        r->src = (const uint8_t*)stage + first * tile_rgb;
        r->rc = OSMT_OK;
        first += r->b->n_jobs;
    }
}

int worker_render_body(osmt_worker* w, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgb, size_t stride) {
    if (!w || !w->ctx || !out_rgb) return fail(OSMT_INVALID_ARG, "NULL argument");
    osmt_ctx* ctx = w->ctx;
    int rc = validate_batch(batch); /* by the requesting thread: the merged batch is trusted */
    if (rc != OSMT_OK) return rc;
    rc = label_batch_shape_ok(batch, labels);
    if (rc != OSMT_OK) return rc;
    const size_t W = (size_t)OSMT_TILE_SIZE * batch->scale, tile_rgb = W * W * 3;
    if (stride < tile_rgb) return fail(OSMT_INVALID_ARG, "out_tile_stride_bytes < W*H*3");
    if (batch->n_jobs == 0) return OSMT_OK;
    if (batch->n_jobs > CO_MAX_TILES) return osmt_render_batch_rgb(ctx, batch, labels, out_rgb, stride); /* a batch of its own anyway */

    coalesce_req me;
    me.b = batch;
    me.lb = labels;
    me.out = out_rgb;
    me.stride = stride;
    std::vector<coalesce_req*> group;
    group.reserve(CO_MAX_TILES); /* every request has at least one tile: push_back below cannot throw */
    {
        std::unique_lock<std::mutex> lk(ctx->co_mu);
        ctx->co_queue.push_back(&me);
        for (;;) {
            if (me.done) break;
            if (!me.taken && ctx->co_in_flight < co_max_in_flight()) {
                /* leader: everything that waits and fits, in arrival order (its own request is in there) */
                size_t tiles = 0;
                /* the merged pools are indexed with 32 bits and rendered without a second validation: a request that
                 * would take any pool past 2^31 entries waits for the next leader (alone it is what its own validation saw) */
                unsigned long long pool[7] = {0, 0, 0, 0, 0, 0, 0};
                const osmt_batch* b0 = ctx->co_queue.front()->b;
                while (!ctx->co_queue.empty()) {
                    coalesce_req* r = ctx->co_queue.front();
                    if (r->b->scale != b0->scale || r->b->coord_kind != b0->coord_kind) break; /* the next leader's */
                    if (!group.empty() && tiles + r->b->n_jobs > CO_MAX_TILES) break;
                    const unsigned long long add[7] = {r->b->n_ops, r->b->n_rings, r->b->n_pts, r->b->n_nodes, r->b->n_dashes,
                                                       r->lb ? r->lb->n_labels : 0ull, r->lb ? r->lb->n_segs : 0ull};
                    bool fits32 = true;
                    for (int k = 0; k < 7; ++k) fits32 = fits32 && pool[k] + add[k] < 0x7FFFFFFFull;
                    if (!group.empty() && !fits32) break;
                    for (int k = 0; k < 7; ++k) pool[k] += add[k];
                    tiles += r->b->n_jobs;
                    r->taken = true;
                    group.push_back(r);
                    ctx->co_queue.pop_front();
                }
                ++ctx->co_in_flight;
                lk.unlock();
                /* whatever happens in there, the group's members are marked done and the slot is given back: followers
                 * wait on co_cv with no timeout */
                try {
                    coalesce_run_group(ctx, group);
                } catch (...) {
                    for (coalesce_req* r : group)
                        if (!r->grp && !r->ran) r->rc = OSMT_OOM; /* (err stays empty: assigning a string could throw again) */
                }
                lk.lock();
                --ctx->co_in_flight;
                for (coalesce_req* r : group) r->done = true;
                group.clear();
                ctx->co_cv.notify_all();
                continue; /* its own request may have been behind an incompatible one: look again */
            }
            ctx->co_cv.wait(lk);
        }
    }
    if (me.rc != OSMT_OK) return fail(me.rc, "%s", me.err.c_str());
    if (me.grp) {
        /* the requester's own share of the read-back: pinned staging -> its buffer, all requesters in parallel */
        for (size_t j = 0; j < batch->n_jobs; ++j) memcpy(out_rgb + j * stride, me.src + j * tile_rgb, tile_rgb);
        coalesce_group* g = me.grp;
        if (g->pending.fetch_sub(1) == 1) {
            stage_release(g->ctx, g->stage);
            delete g;
        }
    }
    return OSMT_OK;
}

}  // namespace

extern "C" {

int osmt_worker_create(osmt_ctx* ctx, osmt_worker** out_worker) {
    return guarded([&] {
        if (!ctx || !out_worker) return fail(OSMT_INVALID_ARG, "NULL argument");
        osmt_worker* w = new (std::nothrow) osmt_worker();
        if (!w) return fail(OSMT_OOM, "out of host memory");
        w->ctx = ctx;
        ctx->refs.fetch_add(1); /* like a scene: the context outlives its workers */
        *out_worker = w;
        return (int)OSMT_OK;
    });
}

void osmt_worker_destroy(osmt_worker* w) {
    if (!w) return;
    osmt_ctx* ctx = w->ctx;
    delete w;
    if (ctx) ctx_release(ctx);
}

int osmt_worker_render(osmt_worker* w, const osmt_batch* batch, const osmt_label_batch* labels, uint8_t* out_rgb, size_t out_tile_stride_bytes) {
    return guarded([&] { return worker_render_body(w, batch, labels, out_rgb, out_tile_stride_bytes); });
}

}  // extern "C"

extern "C" {

static int hbm_copy_probe_body(osmt_ctx* ctx, size_t bytes, uint32_t iters, double* out_copy, double* out_read) {
    if (!ctx || !out_copy) return fail(OSMT_INVALID_ARG, "NULL argument");
    if (bytes < 16 || iters == 0) return fail(OSMT_INVALID_ARG, "nothing to copy");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n16 = bytes / 16;
    char* d = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    HIP_TRY(stream_acquire(ctx, &st));
    hipError_t e = dev_alloc(ctx, (void**)&d, 2 * n16 * 16);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0x5A, 2 * n16 * 16, st);
    for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipEventCreate(&ev[k]);
    if (e == hipSuccess) e = osmt_launch_copy16(d, d + n16 * 16, n16, false, st); /* warm-up */
    if (e == hipSuccess) e = hipEventRecord(ev[0], st);
    for (uint32_t i = 0; i < iters && e == hipSuccess; ++i) e = osmt_launch_copy16(d, d + n16 * 16, n16, false, st);
    if (e == hipSuccess) e = hipEventRecord(ev[1], st);
    for (uint32_t i = 0; i < iters && e == hipSuccess; ++i) e = osmt_launch_copy16(d, d + n16 * 16, n16, true, st);
    if (e == hipSuccess) e = hipEventRecord(ev[2], st);
    const hipError_t es = hipStreamSynchronize(st);
    if (e == hipSuccess) e = es;
    float ms_copy = 0.f, ms_read = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_copy, ev[0], ev[1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_read, ev[1], ev[2]);
    for (int k = 0; k < 3; ++k)
        if (ev[k]) (void)hipEventDestroy(ev[k]);
    dev_free(ctx, d);
    stream_release(ctx, st);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OSMT_OOM : OSMT_HIP_ERROR, "osmt_hbm_copy_probe: %s", hipGetErrorString(e));
    *out_copy = 2.0 * (double)(n16 * 16) * iters / ((double)ms_copy * 1e-3) / 1e9;
    if (out_read) *out_read = (double)(n16 * 16) * iters / ((double)ms_read * 1e-3) / 1e9;
    return OSMT_OK;
}

int osmt_debug_poison_enabled(void) { return poison_alloc() ? 1 : 0; }

int osmt_hbm_copy_probe(osmt_ctx* ctx, size_t bytes, uint32_t iters, double* out_copy, double* out_read) {
    return guarded([&] { return hbm_copy_probe_body(ctx, bytes, iters, out_copy, out_read); });
}

} /* extern "C" */
