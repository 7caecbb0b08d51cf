// Host build of the closed forms the HIP kernels use (osm_renderer_amd/csrc/osmt_geom.h) plus
// sizeof/offsetof probes of the C ABI structs, for the CPU-side tests.
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "../include/osmtile.h"
#include "../osm_renderer_amd/csrc/osmt_geom.h"

extern "C" {
int shim_fill_row_extent(int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, int32_t y, int32_t* xmin, int32_t* xmax) {
    return osmt_fill_row_extent(p1x, p1y, p2x, p2y, y, xmin, xmax);
}
// all rows of one edge at once: out[(y - y_lo)] = {present, xmin, xmax}
void shim_fill_rows(int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, int32_t y_lo, int32_t y_hi, int32_t* out) {
    for (int32_t y = y_lo; y <= y_hi; ++y) {
        int32_t a = 0, b = 0;
        const int r = osmt_fill_row_extent(p1x, p1y, p2x, p2y, y, &a, &b);
        out[3 * (y - y_lo) + 0] = r;
        out[3 * (y - y_lo) + 1] = a;
        out[3 * (y - y_lo) + 2] = b;
    }
}
// all steps k = 0..b of one segment: out[k] = {c, pe, has_extra, pe_extra}
void shim_stroke_steps(int32_t a, int32_t b, int32_t* out) {
    for (int32_t k = 0; k <= b; ++k) osmt_stroke_step(a, b, k, &out[4 * k], &out[4 * k + 1], &out[4 * k + 2], &out[4 * k + 3]);
}
void shim_stroke_steps24(int32_t a, int32_t b, int32_t* out) {
    for (int32_t k = 0; k <= b; ++k) osmt_stroke_step24(a, b, k, &out[4 * k], &out[4 * k + 1], &out[4 * k + 2], &out[4 * k + 3]);
}
// all extra events of a segment: out[i] = {c, k, pe}; returns the count E(b)
int32_t shim_extra_events(int32_t a, int32_t b, int32_t* out, int32_t* counts /* [b+1]: E(K) */) {
    for (int32_t K = 0; K <= b; ++K) counts[K] = osmt_extra_count(a, b, K);
    const int32_t n = counts[b];
    for (int32_t m = 1; m <= n; ++m) osmt_extra_event(a, b, m, &out[3 * (m - 1)], &out[3 * (m - 1) + 1], &out[3 * (m - 1) + 2]);
    return n;
}
double shim_fmod_pos(double x, double y) { return osmt_fmod_pos(x, y); }
/* item ranges of (segment, rectangle): out[8] = k_lo0, k_n0, k_lo1, k_n1, m_lo0, n_x0, m_lo1, n_x1; returns the item count */
uint32_t shim_seg_ranges(int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, double len, double ft, int32_t rx0, int32_t ry0, int32_t rx1,
                         int32_t ry1, int32_t* out) {
    osmt_item_ranges q;
    const uint32_t n = osmt_seg_ranges(p1x, p1y, p2x, p2y, len, ft, rx0, ry0, rx1, ry1, &q);
    out[0] = q.k_lo0; out[1] = q.k_n0; out[2] = q.k_lo1; out[3] = q.k_n1;
    out[4] = q.m_lo0; out[5] = q.n_x0; out[6] = q.m_lo1; out[7] = q.n_x1;
    return n;
}
/* osmt_div_exact against the hardware division on n cases; returns the number of mismatches, first one in bad[0..1] */
size_t shim_div_exact_check(const double* num, const double* den, size_t n, double* bad) {
    size_t miss = 0;
    for (size_t i = 0; i < n; ++i) {
        const double want = num[i] / den[i];
        const double got = osmt_div_exact(num[i], den[i], 1.0 / den[i]);
        if (!(want == got)) {
            if (!miss) {
                bad[0] = num[i];
                bad[1] = den[i];
            }
            ++miss;
        }
    }
    return miss;
}
/* the same on the kernel's own operand shapes, generated here (fast): d = sqrt(dx^2 + dy^2) of integer deltas,
 * n = |integer| (cross products), `rounds` x 2^20 cases from a SplitMix64 stream */
size_t shim_div_exact_sweep(uint64_t seed, size_t rounds, double* bad) {
    size_t miss = 0;
    uint64_t st = seed;
    auto next = [&]() {
        uint64_t z = (st += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    for (size_t k = 0; k < rounds << 20; ++k) {
        const uint64_t a = next(), b = next();
        const int sh = (int)(a & 31);                 /* deltas of every magnitude up to 2^29 */
        const double dx = (double)((a >> 8) & ((1ull << (sh < 29 ? sh + 1 : 29)) - 1));
        const double dy = (double)((a >> 40) & ((1ull << ((sh * 7) % 30)) - 1));
        double d = sqrt(dx * dx + dy * dy);
        if (d < 1.0) d = 1.0;
        const int sn = (int)(b & 63);
        const double nn = (double)(int64_t)((b >> 6) & ((1ull << (sn < 58 ? sn : 58)) - 1)); /* |cross| up to 2^58, rounded like `as f64` */
        const double want = nn / d;
        const double got = osmt_div_exact(nn, d, 1.0 / d);
        if (!(want == got)) {
            if (!miss) {
                bad[0] = nn;
                bad[1] = d;
            }
            ++miss;
        }
    }
    return miss;
}
int64_t shim_udiv(int64_t n, int64_t d) { return osmt_udiv(n, d); }
size_t shim_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(osmt_op);
        case 1: return sizeof(osmt_ring);
        case 2: return sizeof(osmt_tile_job);
        case 3: return sizeof(osmt_batch);
        case 4: return sizeof(osmt_config);
        case 5: return sizeof(osmt_label);
        case 6: return sizeof(osmt_label_batch);
        case 10: return offsetof(osmt_op, opacity);
        case 11: return offsetof(osmt_op, width);
        case 12: return offsetof(osmt_op, n_dashes);
        case 13: return offsetof(osmt_op, image_id);
        case 20: return offsetof(osmt_tile_job, n_ops);
        case 21: return offsetof(osmt_tile_job, pt_off);
        case 30: return offsetof(osmt_batch, coord_kind);
        case 31: return offsetof(osmt_batch, latlon);
        case 32: return offsetof(osmt_batch, dashes);
        case 33: return offsetof(osmt_batch, nodes);
        case 34: return offsetof(osmt_batch, node_refs);
        case 40: return offsetof(osmt_label, image_id);
        case 41: return offsetof(osmt_label, n_segs);
        case 42: return offsetof(osmt_label, icon_center_x);
        case 43: return offsetof(osmt_label_batch, job_label_off);
        case 44: return offsetof(osmt_label_batch, n_segs);
    }
    return 0;
}
}
