// The reference's server shape without a Python GIL in the way: T host threads, each calling
// osmt_render_batch (n tiles per call, pageable buffers) on ONE context, like the worker pool of
// src/http_server.rs:50-83.  Build + run: tools/bench_host_threads_native.sh
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "../include/osmtile.h"

struct Tiles {
    std::vector<osmt_tile_job> jobs;
    std::vector<osmt_op> ops;
    std::vector<osmt_ring> rings;
    std::vector<int32_t> pts;
    std::vector<double> dashes;
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

int main() {
    osmt_ctx* ctx = nullptr;
    osmt_config cfg{0, 0};
    if (osmt_create(&cfg, &ctx) != OSMT_OK) return printf("osmt_create: %s\n", osmt_last_error()), 1;
    for (int n : {1, 16}) {
        for (int T : {1, 4, 16, 64}) {
            std::vector<Tiles> work(T);
            for (int t = 0; t < T; ++t) make(work[t], n, 17u * (unsigned)t + (unsigned)n);
            const int calls = n == 1 ? 400 : 100;
            std::atomic<int> bad{0};
            auto body = [&](int t) {
                std::vector<uint8_t> out((size_t)n * 256 * 256 * 4);
                for (int c = 0; c < calls; ++c)
                    if (osmt_render_batch(ctx, &work[t].batch, out.data(), 256 * 256 * 4) != OSMT_OK) ++bad;
            };
            body(0);  // warm-up
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(body, t);
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("n=%3d tiles/call, %3d threads: %9.0f tiles/s  (%.3f ms per call per thread)%s\n", n, T, (double)T * calls * n / dt,
                   dt / calls * 1e3, bad ? "  ERRORS" : "");
        }
    }
    osmt_destroy(ctx);
    return 0;
}
