"""An independent, dictionary-based Python restatement of the reference's area path — fill_contour (src/draw/fill.rs),
draw_lines / draw_line (src/draw/line.rs), OpacityCalculator (src/draw/opacity_calculator.rs), Point::dist /
push_away_from (src/draw/point.rs) and the TilePixels generation machinery (src/draw/tile_pixels.rs:107-129,150-158,
164-181,205-223) — written from the Rust source separately from oracle/osm_oracle.cpp, for differential tests of the
oracle on the parts no golden image covers (Square / Butt caps, use_caps_for_dashes = false, odd dash lists)."""
import math

NAN = float("nan")


def fmax(a, b):  # f64::max: NaN-ignoring
    return b if a != a else (a if b != b else max(a, b))


def fmin(a, b):
    return b if a != a else (a if b != b else min(a, b))


def rsqrt(v):
    return math.sqrt(v) if v >= 0.0 else NAN  # NaN stays NaN through the comparison below


def rround(v):  # f64::round: half away from zero
    r = math.floor(abs(v))
    if abs(v) - r >= 0.5:
        r += 1
    return int(math.copysign(r, v))


def dist(a, b):
    dx, dy = float(a[0] - b[0]), float(a[1] - b[1])
    return math.sqrt(dx * dx + dy * dy)


def push_away_from(p, other, by):
    d = by / dist(p, other)
    return (p[0] + rround(float(p[0] - other[0]) * d), p[1] + rround(float(p[1] - other[1]) * d))


class Pixels:
    def __init__(self, canvas, W=256):
        self.W = W
        self.canvas = tuple(1.0 * (c / 255.0) for c in canvas) + (1.0,) if canvas is not None else (0.0, 0.0, 0.0, 1.0)
        self.px, self.nxt, self.gen = {}, {}, 0

    def _blend(self, k):
        if k in self.nxt:
            c, _ = self.nxt.pop(k)
            old = self.px.get(k, self.canvas)
            self.px[k] = tuple(c[i] + (1.0 - c[3]) * old[i] for i in range(4))

    def set_pixel(self, x, y, c):
        if x < 0 or x > self.W - 1 or y < 0 or y > self.W - 1:
            return
        k = (x, y)
        if k in self.nxt and self.nxt[k][1] == self.gen:
            if c[3] > self.nxt[k][0][3]:
                self.nxt[k] = (c, self.gen)
            return
        self._blend(k)
        self.nxt[k] = (c, self.gen)

    def finish(self):
        for k in list(self.nxt):
            self._blend(k)

    def rgb(self):
        import numpy as np

        out = np.zeros((self.W, self.W, 3), dtype=np.uint8)
        base = self.canvas
        for y in range(self.W):
            for x in range(self.W):
                p = self.px.get((x, y), base)
                for i in range(3):
                    m = 0.0 if p[3] == 0.0 else p[i] / p[3]
                    v = 255.0 * m
                    out[y, x, i] = 0 if v != v else max(0, min(255, int(v)))
        return out


def from_color(color, o):
    return (o * (color[0] / 255.0), o * (color[1] / 255.0), o * (color[2] / 255.0), o)


def fill_contour(pairs, color, opacity, px):
    rows = {}  # IndexMap<i32, IndexMap<usize, Edge>>: python dicts keep insertion order
    for idx, (p1, p2) in enumerate(pairs):
        dx, dy = abs(p2[0] - p1[0]), -abs(p2[1] - p1[1])
        sx = 1 if p1[0] < p2[0] else -1
        sy = 1 if p1[1] < p2[1] else -1
        err = dx + dy
        cur = [p1[0], p1[1]]
        while True:
            is_start, is_end = tuple(cur) == tuple(p1), tuple(cur) == tuple(p2)
            poisoned = (p1[1] <= p2[1]) if is_start else ((p2[1] <= p1[1]) if is_end else False)
            if 0 <= cur[1] <= px.W - 1:
                e = rows.setdefault(cur[1], {}).setdefault(idx, [cur[0], cur[0], poisoned])
                e[0], e[1], e[2] = min(e[0], cur[0]), max(e[1], cur[0]), e[2] or poisoned
            if is_end:
                break
            e2 = 2 * err
            if e2 >= dy:
                err += dy
                cur[0] += sx
            if e2 <= dx:
                err += dx
                cur[1] += sy
    for y, edges in rows.items():
        good = sorted((e for e in edges.values() if not e[2]), key=lambda e: e[0])  # stable
        i = 0
        while i + 1 < len(good):
            for x in range(max(good[i][0], 0), min(good[i + 1][1], px.W - 1) + 1):
                px.set_pixel(x, y, from_color(color, opacity))
            i += 2


class OC:
    def __init__(self, hlw, dashes, cap):
        self.hlw, self.segs, self.total, self.trav = hlw, [], 0.0, 0.0
        if dashes is not None:
            non_trivial = cap in ("round", "square")
            for idx in list(range(len(dashes))) + [0]:
                d = dashes[idx]
                start = self.total
                if idx != 0 or not self.segs:
                    self.total += d
                if idx % 2 != 0:
                    continue
                end = start + d
                orig = (start, end) if cap == "round" else None
                if non_trivial:
                    start -= hlw
                    end += hlw
                mid = (start + end) / 2.0
                self.segs.append((fmin(start - 0.5, mid - 1.0), fmin(start + 0.5, mid), fmax(end - 0.5, mid), fmax(end + 0.5, mid + 1.0),
                                  fmin(end - start, 1.0), orig))

    def calculate(self, cd, sd):
        if not self.segs:
            op, cap_d = 1.0, None
        else:
            r = self.trav + sd
            if self.total > 0.0:
                r = math.fmod(r, self.total)
            op, cap_d = 0.0, None
            for (sf, st, ef, et, mul, orig) in self.segs:
                if r < sf or r > et:
                    continue
                base = (r - sf) / (st - sf) if r <= st else (1.0 if r < ef else (et - r) / (et - ef))
                op = fmax(op, mul * base)
                if orig is not None:
                    dcap = orig[0] - r if r < orig[0] else (0.0 if r <= orig[1] else r - orig[1])
                    if cap_d is None or dcap < cap_d:
                        cap_d = dcap
        c = cap_d if cap_d is not None else 0.0
        h = rsqrt(self.hlw * self.hlw - c * c)
        ff, ft = fmax(h - 0.5, 0.0), fmax(h + 0.5, 1.0)
        mul = fmin(2.0 * h, 1.0)
        v = 1.0 if cd < ff else ((ft - cd) / (ft - ff) if cd < ft else 0.0)
        cdo = mul * v
        return fmin(op, cdo), cdo > 0.0


def draw_line(p1, p2, color, op0, oc, px):
    if tuple(p1) == tuple(p2):
        return
    inc = lambda a, b: 1 if a <= b else -1
    dx, dy = abs(p2[0] - p1[0]), abs(p2[1] - p1[1])
    swap = dx > dy
    sw = (lambda a, b: (b, a)) if swap else (lambda a, b: (a, b))
    mn, mx = sw(p1[0], p1[1])
    mn_last, mx_last = sw(p2[0], p2[1])
    mn_d, mx_d = sw(dx, dy)
    mn_inc, mx_inc = sw(inc(p1[0], p2[0]), inc(p1[1], p2[1]))

    def upd(e):
        if e + 2 * mn_d > mx_d:
            return e - 2 * mx_d + 2 * mn_d, True
        return e + 2 * mn_d, False

    cconst = p2[0] * p1[1] - p2[1] * p1[0]
    sdx, sdy = p2[0] - p1[0], p2[1] - p1[1]
    den = math.sqrt(float(dy) * float(dy) + float(dx) * float(dx))

    def perps(mn, mx, p_error):
        for mul in (1, -1):
            p_mn, p_mx, e = mx, mn, mul * p_error
            while True:
                x, y = sw(p_mx, p_mn)
                cd = abs(float(cconst + sdy * x - sdx * y)) / den
                ld = dist((x, y), p1)
                sd = math.sqrt(fmax(ld * ld - cd * cd, 0.0))
                o, inl = oc.calculate(cd, sd)
                if not inl:
                    break
                px.set_pixel(x, y, from_color(color, op0 * o))
                e, corrected = upd(e)
                if corrected:
                    p_mn -= mul * mx_inc
                p_mx += mul * mn_inc

    error = p_error = 0
    while True:
        perps(mn, mx, p_error)
        if mn == mn_last and mx == mx_last:
            break
        error, c = upd(error)
        if c:
            mn += mn_inc
            p_error, c2 = upd(p_error)
            if c2:
                perps(mn, mx, p_error)
        mx += mx_inc


def draw_lines(pairs, width, color, opacity, dashes, cap, use_caps_for_dashes, px):
    hw = width / 2.0
    oc = OC(hw, dashes, cap if use_caps_for_dashes else None)
    oc_caps = OC(hw, [0.0], cap)
    has_caps = cap in ("round", "square")
    first = True
    for i, (p1, p2) in enumerate(pairs):
        draw_line(p1, p2, color, opacity, oc, px)
        oc.trav += dist(p1, p2)
        if tuple(p1) != tuple(p2) and has_caps:
            if first:
                draw_line(p1, push_away_from(p1, p2, hw), color, opacity, oc_caps, px)
            if i + 1 == len(pairs):
                draw_line(p2, push_away_from(p2, p1, hw), color, opacity, oc_caps, px)
        first = False
