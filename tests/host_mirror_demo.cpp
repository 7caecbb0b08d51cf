// Drives the C++ mirror of the reference's draw interface (osm_renderer_amd/host/osmt_draw.hpp)
// exactly the way Drawer::draw_to_pixels drives TilePixels (src/draw/drawer.rs:60-131):
// reset -> per area {fill_contour | draw_lines ; bump_generation} -> blend -> to_rgb_triples.
// Writes the RGB triples of two tiles (one via TilePixels::to_rgb_triples, both via TileBatch)
// to the file given as argv[1]; tests/test_gpu_host_mirror.py compares them with the oracle.
#include <cstdio>

#include "../osm_renderer_amd/host/osmt_draw.hpp"

using namespace osmt;

static PointPairs pairs(std::initializer_list<Point> pts) {
    PointPairs out;
    const Point* prev = nullptr;
    for (const Point& p : pts) {
        if (prev) out.push_back({*prev, p});
        prev = &p;
    }
    return out;
}

static void draw_tile_a(TilePixels& px) {
    px.reset(Color{241, 238, 232});
    fill_contour(pairs({{10, 10}, {200, 30}, {150, 220}, {20, 180}, {10, 10}}), Filler::from_color(Color{200, 40, 40}), 0.6, px);
    px.bump_generation();
    px.bump_generation();  // an area whose style draws nothing (drawer.rs:218)
    PointPairs mp = pairs({{60, 60}, {120, 70}, {100, 130}, {60, 60}});
    PointPairs hole = pairs({{300, 300}, {310, 300}, {305, 320}, {300, 300}});
    mp.insert(mp.end(), hole.begin(), hole.end());  // two rings of one multipolygon
    fill_contour(mp, Filler::from_color(Color{20, 40, 220}), 1.0, px);
    px.bump_generation();
    draw_lines(pairs({{5, 250}, {90, 120}, {180, 200}, {250, 20}}), 6.0, Color{10, 120, 10}, 0.8,
               std::vector<double>{9.0, 4.0}, LineCap::Round, false, px);
    px.bump_generation();
    draw_lines(pairs({{0, 0}, {255, 255}}), 1.5, Color{0, 0, 0}, 1.0, std::nullopt, std::nullopt, false, px);
    px.bump_generation();
    px.blend_unfinished_pixels(false);
    // label pass (drawer.rs:107-125): text only; a second label that collides with it; a curve
    auto square = [](Rasterizer& r, double x0, double y0, double x1, double y1) {
        r.draw_line(x0, y0, x0, y1);
        r.draw_line(x0, y1, x1, y1);
        r.draw_line(x1, y1, x1, y0);
        r.draw_line(x1, y0, x0, y0);
    };
    {
        Rasterizer r(Color{102, 102, 255});
        square(r, 40.25, 40.5, 70.75, 52.125);
        r.save_to_figure(px);
        px.bump_label_generation(true);
    }
    {
        Rasterizer r(Color{255, 0, 0});
        square(r, 60, 45, 90, 60);
        r.save_to_figure(px);
        px.bump_label_generation(true);
    }
    {
        Rasterizer r(Color{0, 0, 0});
        r.draw_line(150, 150, 150, 180);
        r.draw_quad(150, 180, 190, 165, 150, 150);
        r.save_to_figure(px);
        px.bump_label_generation(true);
    }
    px.blend_unfinished_pixels(true);
}

static void draw_tile_b(TilePixels& px) {
    px.reset(std::nullopt);
    draw_lines(pairs({{30, 30}, {220, 60}}), 12.0, Color{255, 200, 0}, 0.5, std::nullopt, LineCap::Square, true, px);
    px.bump_generation();
    px.blend_unfinished_pixels(false);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    try {
        Context ctx(0);
        TilePixels a(ctx, 1), b(ctx, 1);
        draw_tile_a(a);
        draw_tile_b(b);
        RgbTriples single = a.to_rgb_triples(Tile{15, 19807, 10243});
        TileBatch batch(ctx, 1);
        batch.add(Tile{15, 19807, 10243}, a);
        batch.add(Tile{15, 19808, 10243}, b);
        std::vector<TileRenderedPixels> both = batch.render();
        FILE* f = fopen(argv[1], "wb");
        if (!f) return 3;
        auto dump = [&](const RgbTriples& t) {
            for (auto& [r, g, bl] : t) {
                const unsigned char px[3] = {r, g, bl};
                fwrite(px, 1, 3, f);
            }
        };
        dump(single);
        dump(both[0].triples);
        dump(both[1].triples);
        fclose(f);
        if (argc > 2) {  // the first tile again as a PNG file written by the GPU (Drawer::draw_tile)
            std::vector<std::vector<uint8_t>> png = batch.render_png();
            FILE* g = fopen(argv[2], "wb");
            if (!g) return 6;
            fwrite(png[0].data(), 1, png[0].size(), g);
            fclose(g);
        }
        // error behaviour: an invalid scale surfaces as osmt::Error, not a crash
        try {
            TilePixels bad(ctx, 99);
            bad.reset(std::nullopt);
            bad.to_rgb_triples();
            return 4;
        } catch (const Error& e) {
            if (e.code != OSMT_INVALID_ARG) return 5;
        }
    } catch (const Error& e) {
        fprintf(stderr, "osmt error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
