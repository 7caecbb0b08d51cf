// The reference's server shape (src/http_server.rs:50-83,105-108): T worker threads, each answering requests for ONE tile.
// Compares the batch entry called per request (osmt_render_batch_rgb, what round 3 measured) with the gathering
// per-request entry (osmt_worker_render): throughput, p50 / p99 latency per request, one JSON line per configuration.
// Build + run: tools/worker_bench.sh [threads...]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../include/osmtile.h"

struct Tiles {
    std::vector<osmt_tile_job> jobs;
    std::vector<osmt_op> ops;
    std::vector<osmt_ring> rings;
    std::vector<int32_t> pts;
    osmt_batch batch{};
};

// config-2-like content: 50 octagons + 40 five-segment polylines per tile, integer points
static void make(Tiles& t, int n, unsigned seed) {
    std::mt19937 rng(seed);
    auto U = [&](int a, int b) { return (int)(rng() % (unsigned)(b - a)) + a; };
    for (int j = 0; j < n; ++j) {
        osmt_tile_job job{};
        job.zoom = 15, job.x = 19000 + j, job.y = 10000, job.has_canvas = 1;
        job.canvas_rgb[0] = 0xF1, job.canvas_rgb[1] = 0xEE, job.canvas_rgb[2] = 0xE8;
        job.op_off = (uint32_t)t.ops.size(), job.pt_off = (uint32_t)(t.pts.size() / 2);
        for (int k = 0; k < 90; ++k) {
            osmt_op op{};
            op.ring_off = (uint32_t)t.rings.size(), op.n_rings = 1;
            op.color[0] = (uint8_t)rng(), op.color[1] = (uint8_t)rng(), op.color[2] = (uint8_t)rng();
            osmt_ring r{(uint32_t)(t.pts.size() / 2), 0};
            if (k < 50) {
                op.kind = OSMT_OP_FILL_COLOR, op.opacity = k % 3 ? 1.0 : 0.6;
                const int cx = U(-32, 288), cy = U(-32, 288), rad = U(8, 48);
                static const int dx[8] = {100, 71, 0, -71, -100, -71, 0, 71}, dy[8] = {0, 71, 100, 71, 0, -71, -100, -71};
                for (int v = 0; v <= 8; ++v) t.pts.push_back(cx + rad * dx[v % 8] / 100), t.pts.push_back(cy + rad * dy[v % 8] / 100);
                r.n_pts = 9;
            } else {
                op.kind = OSMT_OP_STROKE, op.opacity = 1.0, op.width = 0.5 + (rng() % 8);
                op.cap = k % 3 == 0 ? OSMT_CAP_ROUND : OSMT_CAP_NONE;
                int x = U(-16, 272), y = U(-16, 272);
                for (int v = 0; v < 6; ++v) t.pts.push_back(x), t.pts.push_back(y), x += U(-48, 48), y += U(-48, 48);
                r.n_pts = 6;
            }
            t.rings.push_back(r);
            t.ops.push_back(op);
        }
        job.n_ops = 90, job.n_pts = (uint32_t)(t.pts.size() / 2) - job.pt_off;
        t.jobs.push_back(job);
    }
    t.batch.jobs = t.jobs.data(), t.batch.n_jobs = t.jobs.size();
    t.batch.ops = t.ops.data(), t.batch.n_ops = t.ops.size();
    t.batch.rings = t.rings.data(), t.batch.n_rings = t.rings.size();
    t.batch.coord_kind = OSMT_COORD_POINT_I32, t.batch.scale = 1;
    t.batch.points = t.pts.data(), t.batch.n_pts = t.pts.size() / 2;
}

int main(int argc, char** argv) {
    osmt_ctx* ctx = nullptr;
    osmt_config cfg{0, 0};
    if (osmt_create(&cfg, &ctx) != OSMT_OK) return printf("osmt_create: %s\n", osmt_last_error()), 1;
    std::vector<int> threads;
    for (int i = 1; i < argc; ++i) threads.push_back(atoi(argv[i]));
    if (threads.empty()) threads = {1, 4, 16, 64};
    const size_t tile_rgb = 256 * 256 * 3;
    // reference pixels of every thread's tile from the batch entry: the gathered path must return the same bytes
    for (int mode = 0; mode < 2; ++mode) {
        for (int T : threads) {
            std::vector<Tiles> work(T);
            for (int t = 0; t < T; ++t) make(work[t], 1, 17u * (unsigned)t + 1u);
            std::vector<std::vector<uint8_t>> want(T, std::vector<uint8_t>(tile_rgb));
            for (int t = 0; t < T; ++t)
                if (osmt_render_batch_rgb(ctx, &work[t].batch, nullptr, want[t].data(), tile_rgb) != OSMT_OK)
                    return printf("reference render: %s\n", osmt_last_error()), 1;
            const int calls = T >= 16 ? 300 : 600;
            std::atomic<int> bad{0}, differ{0};
            std::vector<std::vector<float>> lat(T);
            auto body = [&](int t, int n_calls, bool record) {
                osmt_worker* w = nullptr;
                if (osmt_worker_create(ctx, &w) != OSMT_OK) { ++bad; return; }
                /* OSMT_BENCH_PINNED=1: the caller's output buffer is pinned (osmt_host_alloc), as in bench.py's latency legs */
                std::vector<uint8_t> out_pageable(tile_rgb);
                void* out_pinned = nullptr;
                if (getenv("OSMT_BENCH_PINNED") && osmt_host_alloc(ctx, tile_rgb, &out_pinned) != OSMT_OK) { ++bad; return; }
                uint8_t* const out_p = out_pinned ? (uint8_t*)out_pinned : out_pageable.data();
                struct view { uint8_t* p; uint8_t* data() const { return p; } } out{out_p};
                for (int c = 0; c < n_calls; ++c) {
                    const auto a = std::chrono::steady_clock::now();
                    const int rc = mode == 0 ? osmt_render_batch_rgb(ctx, &work[t].batch, nullptr, out.data(), tile_rgb)
                                             : osmt_worker_render(w, &work[t].batch, nullptr, out.data(), tile_rgb);
                    const auto b = std::chrono::steady_clock::now();
                    if (rc != OSMT_OK) ++bad;
                    if (record) lat[t].push_back(std::chrono::duration<float, std::micro>(b - a).count());
                    if (c % 16 == 0 && memcmp(out.data(), want[t].data(), tile_rgb) != 0) ++differ;
                }
                osmt_worker_destroy(w);
                if (out_pinned) osmt_host_free(ctx, out_pinned);
            };
            {
                std::vector<std::thread> th;  // warm-up: streams, staging buffers, device buffers of every size
                for (int t = 0; t < T; ++t) th.emplace_back(body, t, 20, false);
                for (auto& x : th) x.join();
            }
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(body, t, calls, true);
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::vector<float> all;
            for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
            std::sort(all.begin(), all.end());
            printf("{\"entry\": \"%s\", \"threads\": %d, \"tiles_per_s\": %.0f, \"p50_us\": %.1f, \"p99_us\": %.1f, \"max_us\": %.1f, \"errors\": %d, "
                   "\"differing\": %d}\n",
                   mode == 0 ? "osmt_render_batch_rgb" : "osmt_worker_render", T, (double)T * calls / dt, all[all.size() / 2],
                   all[std::min(all.size() - 1, (size_t)(all.size() * 0.99))], all.back(), (int)bad, (int)differ);
            fflush(stdout);
        }
    }
    osmt_destroy(ctx);
    return 0;
}
