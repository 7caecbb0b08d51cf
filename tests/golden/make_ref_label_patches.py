#!/usr/bin/env python
"""Builds tests/golden/ref_label_patches.json: a label of the reference's OWN golden image
tests/rendered/17_expected.png (z17 mosaic, tile col 0 / row 2) together with the label display list
(osmt_label: icon + Rasterizer::draw_line calls) that re-synthesises it, plus the same station at z14 seen from
the tile below it (see main()).

The label is the metro station node "Арбатская": icon symbols/station.png (mapnik.mapcss:6073-6075) and the
text rule mapnik.mapcss:6123-6131 (font-size 11, text-color #6666ff; nodes are placed with
TextPosition::Center, drawer.rs:252-259).  NOTHING is fitted except the node's integer pixel position, and that
is read off the icon: the 9x9 icon sits at x 73..81, y 193..201 of the tile, so get_start_coord
(labeler.rs:92-95) gives the centre (78, 198).  Everything else is computed:

  * glyph outlines, advance widths and vertical metrics come from the reference's font
    src/draw/font/NotoSans-Regular.ttf through the reader below, a restatement of the stb_truetype crate
    (Cargo.lock pins 0.3.1; not vendored in the reference) for exactly the calls font/text_placer.rs makes:
    FontInfo::find_glyph_index (cmap format 4/12), get_glyph_h_metrics, get_glyph_kern_advance (the font has
    no `kern` table: always 0), get_v_metrics, scale_for_pixel_height (f32), get_glyph_shape (simple and
    compound glyphs, the stbtt on/off-curve walk with its integer midpoints);
  * the glyph walk and placement follow TextPlacer::place / Glyph::rasterize (font/text_placer.rs:24-160,
    232-259); curves are flattened like Rasterizer::draw_quad (font/rasterizer.rs:90-113) with libm's hypot
    (osm_renderer_amd.labels.flatten_quad).

The oracle's label pass then reproduces every pixel of the text (rows 202..214) and of the icon with ZERO
differences; only the dashed subway line that crosses the icon's rows left and right of it (a different way,
not part of this label) is masked out.  This pins font/rasterizer.rs, tile_pixels.rs:131-162,205-223 and
labeler.rs:91-106 of the oracle to the reference's real output.

Run in the build container only (reads /root/reference); the JSON it writes is the fixture.
"""
import json
import os
import struct
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from osm_renderer_amd import labels  # noqa: E402

REF = "/root/reference"


class Font:
    def __init__(self, data):
        self.d = data
        n = struct.unpack_from('>H', data, 4)[0]
        self.t = {}
        for i in range(n):
            tag, _, off, ln = struct.unpack_from('>4sIII', data, 12 + 16 * i)
            self.t[tag.decode()] = (off, ln)
        self.head = self.t['head'][0]; self.hhea = self.t['hhea'][0]; self.hmtx = self.t['hmtx'][0]
        self.loca = self.t['loca'][0]; self.glyf = self.t['glyf'][0]
        self.kern = self.t.get('kern', (0, 0))[0]
        self.index_to_loc_format = self.u16(self.head + 50)
        self.num_glyphs = self.u16(self.t['maxp'][0] + 4)
        cmap = self.t['cmap'][0]
        self.index_map = 0
        for i in range(self.u16(cmap + 2)):
            rec = cmap + 4 + 8 * i
            plat, enc, off = self.u16(rec), self.u16(rec + 2), self.u32(rec + 4)
            if plat == 3 and enc in (1, 10): self.index_map = cmap + off
            elif plat == 0: self.index_map = cmap + off
    def u8(self, o): return self.d[o]
    def i8(self, o): return struct.unpack_from('>b', self.d, o)[0]
    def u16(self, o): return struct.unpack_from('>H', self.d, o)[0]
    def i16(self, o): return struct.unpack_from('>h', self.d, o)[0]
    def u32(self, o): return struct.unpack_from('>I', self.d, o)[0]

    def find_glyph_index(self, cp):
        im = self.index_map; fmt = self.u16(im)
        if fmt == 4:
            if cp > 0xFFFF: return 0
            segcount = self.u16(im + 6) >> 1
            search_range = self.u16(im + 8) >> 1
            entry_selector = self.u16(im + 10)
            range_shift = self.u16(im + 12) >> 1
            end_count = im + 14
            search = end_count
            if cp >= self.u16(search + range_shift * 2): search += range_shift * 2
            search -= 2
            while entry_selector:
                search_range >>= 1
                end = self.u16(search + search_range * 2)
                if cp > end: search += search_range * 2
                entry_selector -= 1
            search += 2
            item = (search - end_count) >> 1
            start = self.u16(im + 14 + segcount * 2 + 2 + 2 * item)
            if cp < start: return 0
            offset = self.u16(im + 14 + segcount * 6 + 2 + 2 * item)
            if offset == 0:
                return (cp + self.i16(im + 14 + segcount * 4 + 2 + 2 * item)) & 0xFFFF
            return self.u16(offset + (cp - start) * 2 + im + 14 + segcount * 6 + 2 + 2 * item)
        if fmt in (12, 13):
            ngroups = self.u32(im + 12); low, high = 0, ngroups
            while low < high:
                mid = low + ((high - low) >> 1)
                sc, ec = self.u32(im + 16 + mid * 12), self.u32(im + 16 + mid * 12 + 4)
                if cp < sc: high = mid
                elif cp > ec: low = mid + 1
                else:
                    sg = self.u32(im + 16 + mid * 12 + 8)
                    return sg + cp - sc if fmt == 12 else sg
            return 0
        raise NotImplementedError(fmt)

    def h_metrics(self, g):
        n = self.u16(self.hhea + 34)
        if g < n: return self.i16(self.hmtx + 4 * g), self.i16(self.hmtx + 4 * g + 2)
        return self.i16(self.hmtx + 4 * (n - 1)), self.i16(self.hmtx + 4 * n + 2 * (g - n))

    def v_metrics(self):
        return self.i16(self.hhea + 4), self.i16(self.hhea + 6), self.i16(self.hhea + 8)  # ascent, descent, line_gap

    def kern_advance(self, g1, g2):
        k = self.kern
        if not k: return 0
        if self.u16(k + 2) < 1: return 0
        if self.u16(k + 8) != 1: return 0
        l, r = 0, self.u16(k + 10) - 1
        needle = (g1 << 16) | g2
        while l <= r:
            m = (l + r) >> 1
            straw = self.u32(k + 18 + m * 6)
            if needle < straw: r = m - 1
            elif needle > straw: l = m + 1
            else: return self.i16(k + 22 + m * 6)
        return 0

    def scale_for_pixel_height(self, h):
        a, d, _ = self.v_metrics()
        return float(np.float32(h) / np.float32(a - d))

    def glyf_offset(self, g):
        if g >= self.num_glyphs: return None
        if self.index_to_loc_format == 0:
            g1 = self.glyf + self.u16(self.loca + g * 2) * 2; g2 = self.glyf + self.u16(self.loca + g * 2 + 2) * 2
        else:
            g1 = self.glyf + self.u32(self.loca + g * 4); g2 = self.glyf + self.u32(self.loca + g * 4 + 4)
        return None if g1 == g2 else g1

    def glyph_shape(self, gi):
        """list of (type, x, y, cx, cy), type in 'M','L','Q' (stbtt vmove / vline / vcurve), or None"""
        g = self.glyf_offset(gi)
        if g is None: return None
        ncont = self.i16(g)
        V = []
        def i16c(v):  # wrap to i16 like the (i16) casts
            v &= 0xFFFF
            return v - 0x10000 if v & 0x8000 else v
        if ncont > 0:
            end_pts = g + 10
            ins = self.u16(g + 10 + ncont * 2)
            p = g + 10 + ncont * 2 + 2 + ins
            n = 1 + self.u16(end_pts + ncont * 2 - 2)
            flags = []; fc = 0; fl = 0
            for _ in range(n):
                if fc == 0:
                    fl = self.u8(p); p += 1
                    if fl & 8: fc = self.u8(p); p += 1
                else: fc -= 1
                flags.append(fl)
            xs = []; x = 0
            for fl in flags:
                if fl & 2:
                    dx = self.u8(p); p += 1
                    x += dx if fl & 16 else -dx
                elif not (fl & 16):
                    x = x + self.i16(p); p += 2
                x = i16c(x); xs.append(x)
            ys = []; y = 0
            for fl in flags:
                if fl & 4:
                    dy = self.u8(p); p += 1
                    y += dy if fl & 32 else -dy
                elif not (fl & 32):
                    y = y + self.i16(p); p += 2
                y = i16c(y); ys.append(y)
            def close(was_off, start_off, sx, sy, scx, scy, cx, cy):
                if start_off:
                    if was_off: V.append(('Q', (cx + scx) >> 1, (cy + scy) >> 1, cx, cy))
                    V.append(('Q', sx, sy, scx, scy))
                else:
                    if was_off: V.append(('Q', sx, sy, cx, cy))
                    else: V.append(('L', sx, sy, 0, 0))
            next_move = 0; j = 0; was_off = False; start_off = False
            sx = sy = cx = cy = scx = scy = 0
            i = 0
            while i < n:
                fl, x, y = flags[i], xs[i], ys[i]
                if next_move == i:
                    if i != 0: close(was_off, start_off, sx, sy, scx, scy, cx, cy)
                    start_off = not (fl & 1)
                    if start_off:
                        scx, scy = x, y
                        if not (flags[i + 1] & 1):
                            sx = (x + xs[i + 1]) >> 1; sy = (y + ys[i + 1]) >> 1
                        else:
                            sx, sy = xs[i + 1], ys[i + 1]; i += 1
                    else:
                        sx, sy = x, y
                    V.append(('M', sx, sy, 0, 0))
                    was_off = False
                    next_move = 1 + self.u16(end_pts + j * 2); j += 1
                else:
                    if not (fl & 1):
                        if was_off: V.append(('Q', (cx + x) >> 1, (cy + y) >> 1, cx, cy))
                        cx, cy = x, y; was_off = True
                    else:
                        if was_off: V.append(('Q', x, y, cx, cy))
                        else: V.append(('L', x, y, 0, 0))
                        was_off = False
                i += 1
            close(was_off, start_off, sx, sy, scx, scy, cx, cy)
            return V
        if ncont < 0:  # compound (stbtt: numberOfContours == -1; the port tests < 0)
            f32 = np.float32
            comp = g + 10
            more = True
            while more:
                flags = self.u16(comp); gidx = self.u16(comp + 2); comp += 4
                mtx = [f32(1), f32(0), f32(0), f32(1), f32(0), f32(0)]
                if flags & 2:
                    if flags & 1:
                        mtx[4] = f32(self.i16(comp)); mtx[5] = f32(self.i16(comp + 2)); comp += 4
                    else:
                        mtx[4] = f32(self.i8(comp)); mtx[5] = f32(self.i8(comp + 1)); comp += 2
                else:
                    raise NotImplementedError('matching points')
                if flags & (1 << 3):
                    mtx[0] = mtx[3] = f32(self.i16(comp)) / f32(16384.0); comp += 2
                elif flags & (1 << 6):
                    mtx[0] = f32(self.i16(comp)) / f32(16384.0); mtx[3] = f32(self.i16(comp + 2)) / f32(16384.0); comp += 4
                elif flags & (1 << 7):
                    mtx[0] = f32(self.i16(comp)) / f32(16384.0); mtx[1] = f32(self.i16(comp + 2)) / f32(16384.0)
                    mtx[2] = f32(self.i16(comp + 4)) / f32(16384.0); mtx[3] = f32(self.i16(comp + 6)) / f32(16384.0); comp += 8
                m = np.sqrt(mtx[0] * mtx[0] + mtx[1] * mtx[1]); n = np.sqrt(mtx[2] * mtx[2] + mtx[3] * mtx[3])
                sub = self.glyph_shape(gidx)
                def cast(v):  # f32 -> i16, Rust `as` (saturating, truncating)
                    v = float(v)
                    return int(max(-32768.0, min(32767.0, v)))
                for (t, x, y, cx, cy) in (sub or []):
                    fx, fy, fcx, fcy = f32(x), f32(y), f32(cx), f32(cy)
                    nx = cast(m * (mtx[0] * fx + mtx[2] * fy + mtx[4])); ny = cast(n * (mtx[1] * fx + mtx[3] * fy + mtx[5]))
                    ncx = cast(m * (mtx[0] * fcx + mtx[2] * fcy + mtx[4])); ncy = cast(n * (mtx[1] * fcx + mtx[3] * fcy + mtx[5]))
                    V.append((t, nx, ny, ncx, ncy))
                more = bool(flags & (1 << 5))
            return V
        return None if ncont == 0 else V


def center_text_segments(font, text, font_size, cx, cy, y_offset):
    """TextPlacer::place, TextPosition::Center, one row (font/text_placer.rs:41-56,103-155)."""
    scale = font.scale_for_pixel_height(font_size)
    asc, desc, gap = [v * scale for v in font.v_metrics()]
    glyphs, prev = [], None
    for ch in text:
        g = font.find_glyph_index(ord(ch))
        w = float(font.h_metrics(g)[0]) * scale
        if prev is not None:
            w += float(font.kern_advance(prev, g)) * scale
        glyphs.append((w, font.glyph_shape(g)))
        prev = g
    row_width = 0.0
    for w, _ in glyphs:
        row_width += w
    row_height = asc - desc + gap
    cur_y = cy
    if y_offset > 0:
        cur_y += float(y_offset)
    else:
        cur_y -= row_height * 1.0 / 2.0
    cur_x = cx - row_width / 2.0
    out = []
    for w, shape in glyphs:
        baseline, xo = cur_y + asc, cur_x
        if shape:
            labels.glyph_segments(shape, scale, lambda p, xo=xo, baseline=baseline: (xo + p[0], baseline - p[1]), out)
        cur_x += w
    return np.array(out, dtype=np.float64).reshape(-1, 4)


def main():
    font = Font(open(os.path.join(REF, "src/draw/font/NotoSans-Regular.ttf"), "rb").read())
    im = np.array(Image.open(os.path.join(REF, "tests/rendered/17_expected.png")).convert("RGB"))
    tile = im[512:768, 0:256]
    icon = np.array(Image.open(os.path.join(REF, "tests/mapcss/symbols/station.png")).convert("RGBA"))
    cx, cy = 78, 198
    segs = center_text_segments(font, "Арбатская", 11.0, float(cx), float(cy), icon.shape[0] // 2)
    x0, x1, y0, y1 = 50, 111, 190, 215
    mask = np.zeros((y1 - y0 + 1, x1 - x0 + 1), dtype=bool)
    mask[202 - y0 :, :] = True          # the text rows and the canvas around them
    mask[: 193 - y0, :] = True          # canvas above the icon
    mask[193 - y0 : 202 - y0, 73 - x0 : 82 - x0] = True  # the icon itself (drawn over the dashed line)
    out = {
        "_provenance": __doc__,
        "station": {
            "source": "tests/rendered/17_expected.png, mosaic tile (col 0, row 2), tile-relative pixel coordinates",
            "window_x0_x1_y0_y1": [x0, x1, y0, y1],
            "canvas": [0xF1, 0xEE, 0xE8],
            "icon_rgba": icon.tolist(),
            "icon_center": [float(cx), float(cy)],
            "text_color": [0x66, 0x66, 0xFF],
            "segs": segs.tolist(),
            "mask_rows": ["".join("1" if v else "0" for v in row) for row in mask],
            "expected_rgb": tile[y0 : y1 + 1, x0 : x1 + 1].tolist(),
        },
    }
    # The same station at z14 (tests/rendered/14_expected.png): icon symbols/station_small.png (mapnik.mapcss:6065-6067,
    # 6x6, found at x 135..140, y 246..251 of mosaic tile (0, 0) -> node at (138, 249)), font-size 9 (mapnik.mapcss:6113-
    # 6121).  The label hangs over the tile's lower edge: the tile BELOW, mosaic tile (col 0, row 1), sees the same node
    # at (138, 249 - 256 = -7) inside its 3x3 label area (tile_pixels.rs:67-72) and draws the glyphs' lowest rows.  Rows
    # 1..4 of that tile left of x = 147 hold nothing but canvas and those glyph rows (row 0 is the mosaic's red grid line).
    im14 = np.array(Image.open(os.path.join(REF, "tests/rendered/14_expected.png")).convert("RGB"))
    tile14 = im14[256:512, 0:256]
    icon14 = np.array(Image.open(os.path.join(REF, "tests/mapcss/symbols/station_small.png")).convert("RGBA"))
    cx, cy = 138, 249 - 256
    segs14 = center_text_segments(font, "Арбатская", 9.0, float(cx), float(cy), icon14.shape[0] // 2)
    x0, x1, y0, y1 = 100, 146, 1, 4
    out["station_z14_from_the_tile_above"] = {
        "source": "tests/rendered/14_expected.png, mosaic tile (col 0, row 1), tile-relative pixel coordinates",
        "window_x0_x1_y0_y1": [x0, x1, y0, y1],
        "canvas": [0xF1, 0xEE, 0xE8],
        "icon_rgba": icon14.tolist(),
        "icon_center": [float(cx), float(cy)],
        "text_color": [0x66, 0x66, 0xFF],
        "segs": segs14.tolist(),
        "mask_rows": ["1" * (x1 - x0 + 1) for _ in range(y0, y1 + 1)],
        "expected_rgb": tile14[y0 : y1 + 1, x0 : x1 + 1].tolist(),
    }
    with open(os.path.join(HERE, "ref_label_patches.json"), "w") as f:
        json.dump(out, f)
    w14 = tile14[y0 : y1 + 1, x0 : x1 + 1]
    print("station: draw_line calls", len(segs), "mask px", int(mask.sum()), "| z14:", len(segs14), "calls,",
          int((w14 != np.array([0xF1, 0xEE, 0xE8])).any(-1).sum()), "covered px")


if __name__ == "__main__":
    main()
