"""Label lists: the flat form of what Drawer::draw_labels walks (drawer.rs:221-262).

One osmt_label == one Labeler::label_entity call (labeler.rs:16-38): an optional icon blit and an
optional text, the text given as the Rasterizer::draw_line calls (font/rasterizer.rs:27-88) the
reference's glyph walk makes.  Curves are flattened HERE (host side) exactly like
Rasterizer::draw_quad (font/rasterizer.rs:90-113) with libm's hypot, which is what f64::hypot is.
"""
import ctypes as C
import ctypes.util

import numpy as np

from . import abi

LABEL_DTYPE = np.dtype(
    [
        ("has_icon", "u1"),
        ("has_text", "u1"),
        ("text_color", "u1", (3,)),
        ("_pad", "u1", (3,)),
        ("image_id", "u4"),
        ("seg_off", "u4"),
        ("n_segs", "u4"),
        ("_reserved", "u4"),
        ("icon_center_x", "f8"),
        ("icon_center_y", "f8"),
    ]
)
assert LABEL_DTYPE.itemsize == 40

_libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.hypot.restype = C.c_double
_libm.hypot.argtypes = [C.c_double, C.c_double]


def flatten_quad(x0, y0, x1, y1, x2, y2, out):
    """Rasterizer::draw_quad (font/rasterizer.rs:90-113): appends the draw_line calls to `out`."""
    d01 = _libm.hypot(abs(x0 - x1), abs(y0 - y1))
    d12 = _libm.hypot(abs(x1 - x2), abs(y1 - y2))
    d02 = _libm.hypot(abs(x0 - x2), abs(y0 - y2))
    if (d01 + d12) <= 1.0001 * d02:
        out.append((x0, y0, x2, y2))
        return
    m01x, m01y = (x0 + x1) / 2.0, (y0 + y1) / 2.0
    m12x, m12y = (x1 + x2) / 2.0, (y1 + y2) / 2.0
    mx, my = (m01x + m12x) / 2.0, (m01y + m12y) / 2.0
    flatten_quad(x0, y0, m01x, m01y, mx, my, out)
    flatten_quad(mx, my, m12x, m12y, x2, y2, out)


def glyph_segments(vertices, scale, tr, out):
    """Glyph::rasterize (font/text_placer.rs:232-259).  vertices: stb_truetype-style list of
    (type, x, y, cx, cy) in font units with type 'M' (MoveTo), 'L' (LineTo), 'Q' (CurveTo)."""
    frm = (0.0, 0.0)
    for t, x, y, cx, cy in vertices:
        to = (float(x) * scale, float(y) * scale)
        if t == "L":
            p1, p0 = tr(frm), tr(to)
            out.append((p0[0], p0[1], p1[0], p1[1]))
        elif t == "Q":
            mid = (float(cx) * scale, float(cy) * scale)
            p2, p1, p0 = tr(frm), tr(mid), tr(to)
            flatten_quad(p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], out)
        frm = to
    return out


class LabelList:
    """Labels of a batch of tiles (osmt_label_batch) backed by numpy arrays."""

    def __init__(self, labels, job_label_off, segs):
        self.labels = np.ascontiguousarray(labels, dtype=LABEL_DTYPE)
        self.job_label_off = np.ascontiguousarray(job_label_off, dtype=np.uint32)
        self.segs = np.ascontiguousarray(segs, dtype=np.float64).reshape(-1, 4)

    @property
    def n_jobs(self):
        return len(self.job_label_off) - 1

    def as_batch(self):
        b = abi.LabelBatch()
        b.labels = self.labels.ctypes.data_as(C.POINTER(abi.Label))
        b.n_labels = len(self.labels)
        b.job_label_off = self.job_label_off.ctypes.data_as(C.POINTER(C.c_uint32))
        b.segs = self.segs.ctypes.data_as(C.POINTER(C.c_double)) if len(self.segs) else None
        b.n_segs = len(self.segs)
        return b

    def algorithmic_bytes(self):
        """bytes the label pass must read at least once: 40 per label + 32 per draw_line call."""
        return 40 * len(self.labels) + 32 * len(self.segs)

    def subset(self, idx):
        return concat_labels([self._single(i) for i in idx])

    def _single(self, i):
        a, b = int(self.job_label_off[i]), int(self.job_label_off[i + 1])
        lab = self.labels[a:b].copy()
        segs = []
        cur = 0
        for l in lab:
            n = int(l["n_segs"])
            segs.append(self.segs[int(l["seg_off"]) : int(l["seg_off"]) + n])
            l["seg_off"] = cur if n else 0
            cur += n
        segs = np.concatenate(segs) if segs else np.zeros((0, 4))
        return LabelList(lab, [0, len(lab)], segs)


def concat_labels(lists):
    labels, offs, segs = [], [0], []
    seg_cur = 0
    for ll in lists:
        lab = ll.labels.copy()
        lab["seg_off"][lab["n_segs"] > 0] += seg_cur
        labels.append(lab)
        segs.append(ll.segs)
        seg_cur += len(ll.segs)
        base = offs[-1]
        offs.extend((base + ll.job_label_off[1:].astype(np.int64)).tolist())
    return LabelList(
        np.concatenate(labels) if labels else np.zeros(0, LABEL_DTYPE),
        offs,
        np.concatenate(segs) if segs else np.zeros((0, 4)),
    )


class TileLabels:
    """Builder for the labels of ONE tile, in draw order."""

    def __init__(self):
        self._labels = []
        self._segs = []

    def label(self, icon=None, text=None):
        """icon: (image_id, center_x, center_y) or None; text: (color_rgb, segs[n][4]) or None.
        text with zero segments is a text that drew nothing (still `has_text`)."""
        l = np.zeros((), LABEL_DTYPE)
        if icon is not None:
            l["has_icon"] = 1
            l["image_id"] = icon[0]
            l["icon_center_x"], l["icon_center_y"] = float(icon[1]), float(icon[2])
        if text is not None:
            color, segs = text
            segs = np.asarray(segs, dtype=np.float64).reshape(-1, 4)
            l["has_text"] = 1
            l["text_color"] = color
            l["seg_off"] = sum(len(s) for s in self._segs) if len(segs) else 0
            l["n_segs"] = len(segs)
            self._segs.append(segs)
        self._labels.append(l)
        return self

    def build(self):
        labels = np.array(self._labels, dtype=LABEL_DTYPE) if self._labels else np.zeros(0, LABEL_DTYPE)
        segs = np.concatenate(self._segs) if self._segs else np.zeros((0, 4))
        return LabelList(labels, [0, len(labels)], segs)


# ---- synthetic glyphs (TrueType-like quadratic outlines in a 1000-unit em) ----------------------
def _ring(cx, cy, rx, ry, ccw):
    """8 quadratic arcs approximating an ellipse, as (type, x, y, cx, cy) stb-style vertices."""
    k = 1.0 / np.cos(np.pi / 8)
    pts = []
    for i in range(9):
        a = 2 * np.pi * (i % 8) / 8 * (1 if ccw else -1)
        pts.append((cx + rx * np.cos(a), cy + ry * np.sin(a)))
    v = [("M", int(pts[0][0]), int(pts[0][1]), 0, 0)]
    for i in range(8):
        a = (2 * np.pi * (i + 0.5) / 8) * (1 if ccw else -1)
        c = (cx + k * rx * np.cos(a), cy + k * ry * np.sin(a))
        v.append(("Q", int(pts[i + 1][0]), int(pts[i + 1][1]), int(c[0]), int(c[1])))
    return v


def _poly(points):
    v = [("M", points[0][0], points[0][1], 0, 0)]
    for p in points[1:] + [points[0]]:
        v.append(("L", p[0], p[1], 0, 0))
    return v


SYNTH_GLYPHS = [
    # (advance, vertices)
    (620, _ring(310, 360, 250, 370, True) + _ring(310, 360, 150, 270, False)),  # "o"
    (280, _poly([(90, 0), (190, 0), (190, 720), (90, 720)])),  # "l"
    (560, _poly([(80, 0), (500, 0), (500, 90), (180, 90), (180, 720), (80, 720)])),  # "L"
    (600, _poly([(40, 0), (140, 0), (300, 560), (460, 0), (560, 0), (350, 720), (250, 720)])
     + _poly([(215, 200), (385, 200), (360, 290), (240, 290)][::-1])),  # "A"-like with a hole
    (260, []),  # space
]


def synth_text(rng, x, y, font_px, n_glyphs, angle=0.0):
    """Segments of a run of synthetic glyphs starting at (x, y) = left end of the baseline, rotated by `angle`."""
    scale = font_px / 1000.0
    out = []
    s, c = np.sin(angle), np.cos(angle)
    pen = 0.0
    for _ in range(n_glyphs):
        adv, verts = SYNTH_GLYPHS[int(rng.integers(0, len(SYNTH_GLYPHS)))]
        px = pen

        def tr(p, px=px):
            gx, gy = px + p[0], p[1]
            return (x + gx * c + gy * s, y + gx * s - gy * c)

        glyph_segments(verts, scale, tr, out)
        pen += adv * scale
    return np.array(out, dtype=np.float64).reshape(-1, 4), pen


def make_labels(n_tiles, labels_per_tile=24, scale=1, seed=7, n_images=0, image_sizes=None, text_frac=0.85,
                icon_frac=0.4, line_frac=0.3):
    """Synthetic label workload: per tile `labels_per_tile` labels scattered over the 3x3-tile label area
    (labels_bb, tile_pixels.rs:67-72) so that collisions happen both inside and outside the tile."""
    rng = np.random.default_rng(seed)
    W = 256 * scale
    out = []
    for _ in range(n_tiles):
        tl = TileLabels()
        for _ in range(labels_per_tile):
            cx = float(rng.integers(-W // 2, W + W // 2)) + float(rng.integers(0, 2)) * 0.5
            cy = float(rng.integers(-W // 2, W + W // 2)) + float(rng.integers(0, 4)) * 0.25
            icon = None
            y_off = 0.0
            if n_images and rng.random() < icon_frac:
                img = int(rng.integers(0, n_images))
                icon = (img, cx, cy)
                y_off = float(image_sizes[img][0] // 2)
            text = None
            if rng.random() < text_frac:
                font_px = float(rng.choice([9.0, 10.0, 11.0, 12.0, 14.0])) * scale
                n_gl = int(rng.integers(3, 13))
                color = tuple(int(v) for v in rng.integers(0, 256, size=3))
                if rng.random() < line_frac:
                    ang = float(rng.uniform(-1.2, 1.2))
                    segs, _ = synth_text(rng, cx, cy, font_px, n_gl, ang)
                else:
                    probe, width = synth_text(np.random.default_rng(0), 0.0, 0.0, font_px, 0)
                    st = rng.bit_generator.state
                    _, width = synth_text(rng, 0.0, 0.0, font_px, n_gl)
                    rng.bit_generator.state = st
                    segs, _ = synth_text(rng, cx - width / 2.0, cy + y_off + 0.8 * font_px, font_px, n_gl)
                text = (color, segs)
            tl.label(icon=icon, text=text)
        out.append(tl.build())
    return concat_labels(out)
