"""The closed forms used by the HIP kernels (osmt_geom.h, built for the host) against the
literal integer walks of the reference as restated in the oracle."""
import ctypes as C
import random

import numpy as np

from tests import _shim


def _rows_from_walk(oracle, p1, p2):
    """fill.rs:51-104: per row x extent + poison flag, by literally walking."""
    w = oracle.fill_edge_walk(p1, p2, cap=1 << 14)
    rows = {}
    for x, y in w.tolist():
        r = rows.setdefault(y, [x, x])
        r[0] = min(r[0], x)
        r[1] = max(r[1], x)
    poisoned = p1[1] if p1[1] <= p2[1] else p2[1]  # row of the smaller-y endpoint (both when equal)
    if p1[1] == p2[1]:
        return {}
    return {y: tuple(v) for y, v in rows.items() if y != poisoned}


def _check_edge(oracle, p1, p2):
    want = _rows_from_walk(oracle, p1, p2)
    y_lo, y_hi = min(p1[1], p2[1]) - 2, max(p1[1], p2[1]) + 2
    out = np.zeros((y_hi - y_lo + 1, 3), dtype=np.int32)
    _shim.lib().shim_fill_rows(p1[0], p1[1], p2[0], p2[1], y_lo, y_hi, out.ctypes.data_as(C.POINTER(C.c_int32)))
    got = {y_lo + i: (int(r[1]), int(r[2])) for i, r in enumerate(out) if r[0]}
    assert got == want, (p1, p2)


def test_fill_row_extent_exhaustive_small(oracle):
    for dx in range(-24, 25):
        for dy in range(-24, 25):
            _check_edge(oracle, (3, -7), (3 + dx, -7 + dy))


def test_fill_row_extent_random(oracle):
    rnd = random.Random(11)
    for _ in range(1500):
        p1 = (rnd.randint(-600, 600), rnd.randint(-600, 600))
        p2 = (rnd.randint(-600, 600), rnd.randint(-600, 600))
        _check_edge(oracle, p1, p2)
    for _ in range(300):  # steep / shallow / long
        p1 = (rnd.randint(-5000, 5000), rnd.randint(-50, 50))
        p2 = (rnd.randint(-5000, 5000), rnd.randint(-50, 50))
        _check_edge(oracle, p1, p2)
        _check_edge(oracle, (p1[1], p1[0]), (p2[1], p2[0]))


def _main_loop(a, b):
    """line.rs:143-157 literally: list of (c, p_error, has_extra, p_error_extra) per main step."""
    error = pe = c = 0
    out = []

    def upd(e):
        return (e - 2 * b + 2 * a, True) if e + 2 * a > b else (e + 2 * a, False)

    for k in range(b + 1):
        rec = [c, pe, 0, 0]
        if k < b:
            error, corr = upd(error)
            if corr:
                c += 1
                pe, corr2 = upd(pe)
                if corr2:
                    rec[2], rec[3] = 1, pe
        out.append(rec)
    assert c == a
    return out


def test_stroke_step_closed_form():
    rnd = random.Random(5)
    cases = [(a, b) for b in range(1, 40) for a in range(0, b + 1)]
    cases += [(rnd.randint(0, b), b) for b in (rnd.randint(1, 4000) for _ in range(400))]
    cases += [(2047, 2047), (2046, 2047), (1, 2047), (0, 2047), (1024, 2047)]
    for a, b in cases:
        out = np.zeros((b + 1, 4), dtype=np.int32)
        _shim.lib().shim_stroke_steps(a, b, out.ctypes.data_as(C.POINTER(C.c_int32)))
        assert out.tolist() == _main_loop(a, b), (a, b)
        if b < 2048:  # the 32-bit variant used for short segments
            out24 = np.zeros((b + 1, 4), dtype=np.int32)
            _shim.lib().shim_stroke_steps24(a, b, out24.ctypes.data_as(C.POINTER(C.c_int32)))
            assert np.array_equal(out24, out), (a, b)


def test_udiv_matches_integer_division():
    rnd = random.Random(9)
    for _ in range(20000):
        d = rnd.randint(1, 1 << rnd.randint(1, 31))
        n = rnd.randint(0, 1 << rnd.randint(1, 60))
        assert _shim.lib().shim_udiv(n, d) == n // d
    for n, d in [(0, 1), ((1 << 52) - 1, 1), ((1 << 52) - 1, 3), (1 << 52, 7), ((1 << 59), (1 << 29) - 1)]:
        assert _shim.lib().shim_udiv(n, d) == n // d


def test_fmod_pos_is_bit_exact_with_libm():
    import math
    import struct

    rnd = random.Random(21)
    f = _shim.lib().shim_fmod_pos
    cases = []
    for _ in range(60000):
        y = rnd.choice([6.0, 18.0, 12.0, 3.0, 7.5, 0.1, 1e-3, rnd.uniform(0.01, 500.0)])
        x = rnd.choice([rnd.uniform(0, 10 * y), rnd.uniform(0, 1e6), y * rnd.randint(0, 1000), y * rnd.randint(0, 1000) * (1 + 2**-52)])
        cases.append((x, y))
    cases += [(0.0, 1.0), (5.0, 5.0), (4.999999999999999, 5.0), (1e15, 3.0), (0.3, 0.1), (0.7, 0.1), (2**52 - 1.0, 1.0)]
    for x, y in cases:
        assert struct.pack("d", f(x, y)) == struct.pack("d", math.fmod(x, y)), (x, y)


def test_extra_perpendicular_events_enumeration():
    """osmt_extra_count / osmt_extra_event against the events of the literal main loop."""
    rnd = random.Random(31)
    cases = [(a, b) for b in range(1, 36) for a in range(0, b + 1)]
    cases += [(rnd.randint(0, b), b) for b in (rnd.randint(1, 5000) for _ in range(300))]
    cases += [(2047, 2047), (2046, 2047), (1, 2047), (2048, 2048), (1500, 2049)]
    for a, b in cases:
        loop = _main_loop(a, b)
        events = [(rec[0] + 1, k, rec[3]) for k, rec in enumerate(loop) if rec[2]]  # (c after update, k, p_error)
        out = np.zeros((b + 2, 3), dtype=np.int32)
        counts = np.zeros(b + 1, dtype=np.int32)
        n = _shim.lib().shim_extra_events(a, b, out.ctypes.data_as(C.POINTER(C.c_int32)), counts.ctypes.data_as(C.POINTER(C.c_int32)))
        assert n == len(events), (a, b)
        assert [tuple(r) for r in out[:n].tolist()] == events, (a, b)
        want_counts = np.cumsum([0] + [rec[2] for rec in loop[:-1]])  # events at steps k < K
        assert counts.tolist() == want_counts.tolist(), (a, b)


def test_div_exact_equals_the_hardware_division():
    """osmt_div_exact(n, d, RN(1/d)) == n / d bit for bit: the kernel's center_distance (line.rs:116-118) on the
    kernel's operand shapes (d = |integer vector|, n = |integer cross product| as f64), plus adversarial operands."""
    import ctypes as C

    import numpy as np

    from tests import _shim

    L = _shim.lib()
    bad = (C.c_double * 2)()
    miss = L.shim_div_exact_sweep(0x5EED, 24, bad)  # 24 x 2^20 cases of the kernel's shapes
    assert miss == 0, (miss, bad[0], bad[1])
    rnd = np.random.default_rng(11)
    # adversarial: mantissas of all ones / near powers of two, quotients next to rounding boundaries
    m = np.concatenate([
        rnd.integers(1, 1 << 53, size=400000).astype(np.float64),
        (np.float64(1 << 53) - rnd.integers(1, 4096, size=100000)).astype(np.float64),
        (np.float64(1 << 52) + rnd.integers(0, 4096, size=100000)).astype(np.float64),
    ])
    num = m * np.exp2(rnd.integers(-20, 8, size=len(m)))
    d_int = rnd.integers(1, 1 << 30, size=len(m)).astype(np.float64)
    den = np.where(rnd.random(len(m)) < 0.5, np.sqrt(d_int * d_int + rnd.integers(0, 1 << 30, size=len(m)).astype(np.float64) ** 2), d_int)
    den = np.maximum(den, 1.0)
    # the feather quotient (ft0 - cd) / fd0: fd0 = (h + 0.5) - (h - 0.5) is 1.0 up to an ulp, numerators of either sign
    hw = rnd.random(len(m)) * 40.0
    fd = (np.maximum(hw + 0.5, 1.0) - np.maximum(hw - 0.5, 0.0))
    feather = (np.maximum(hw + 0.5, 1.0) - rnd.random(len(m)) * 45.0)
    for nn, dd in ((num, den), (np.rint(num), den), (den * rnd.integers(1, 1000, size=len(m)), den), (feather, fd)):
        nn = np.ascontiguousarray(nn, dtype=np.float64)
        dd = np.ascontiguousarray(dd, dtype=np.float64)
        miss = L.shim_div_exact_check(nn.ctypes.data_as(C.POINTER(C.c_double)), dd.ctypes.data_as(C.POINTER(C.c_double)), len(nn), bad)
        assert miss == 0, (miss, bad[0], bad[1])


def _literal_runs(p1, p2, hw):
    """line.rs:65-158 walked literally for an un-dashed line of half-width hw: one entry per perpendicular run,
    (kind, index, side, pixels set) with kind 0 = main run of step `index`, 1 = extra run of event `index` (1-based)."""
    import math

    inc = lambda a, b: 1 if a <= b else -1
    dx, dy = abs(p2[0] - p1[0]), abs(p2[1] - p1[1])
    swap = dx > dy
    sw = (lambda a, b: (b, a)) if swap else (lambda a, b: (a, b))
    mn, mx = sw(p1[0], p1[1])
    mn_last, mx_last = sw(p2[0], p2[1])
    mn_d, mx_d = sw(dx, dy)
    mn_inc, mx_inc = sw(inc(p1[0], p2[0]), inc(p1[1], p2[1]))
    upd = lambda e: (e - 2 * mx_d + 2 * mn_d, True) if e + 2 * mn_d > mx_d else (e + 2 * mn_d, False)
    cconst = p2[0] * p1[1] - p2[1] * p1[0]
    sdx, sdy = p2[0] - p1[0], p2[1] - p1[1]
    den = math.sqrt(float(dy) * float(dy) + float(dx) * float(dx))
    hl = math.sqrt(hw * hw)
    ff, ft = max(hl - 0.5, 0.0), max(hl + 0.5, 1.0)
    mul0 = min(2.0 * hl, 1.0)
    runs = []

    def perps(kind, index, mn, mx, p_error):
        for side, mul in ((0, 1), (1, -1)):
            p_mn, p_mx, e = mx, mn, mul * p_error
            pix = []
            while True:
                x, y = sw(p_mx, p_mn)
                cd = abs(float(cconst + sdy * x - sdx * y)) / den
                v = 1.0 if cd < ff else ((ft - cd) / (ft - ff) if cd < ft else 0.0)
                if not mul0 * v > 0.0:
                    break
                pix.append((x, y))
                e, corrected = upd(e)
                if corrected:
                    p_mn -= mul * mx_inc
                p_mx += mul * mn_inc
            runs.append((kind, index, side, pix))

    error = p_error = 0
    k = m = 0
    while True:
        perps(0, k, mn, mx, p_error)
        if mn == mn_last and mx == mx_last:
            break
        error, c = upd(error)
        if c:
            mn += mn_inc
            p_error, c2 = upd(p_error)
            if c2:
                m += 1
                perps(1, m, mn, mx, p_error)
        mx += mx_inc
        k += 1
    return runs


def test_seg_ranges_list_every_run_that_draws_into_the_rectangle():
    """osmt_seg_ranges (what k_stroke_bin records per (segment, sub-tile)) must contain every perpendicular run of the
    literal walk that sets a pixel inside the rectangle — for every slope, direction, width and rectangle position —
    and is allowed to list more (it is a cull, the walk decides).  Also reports how tight the cull is."""
    import ctypes as C
    import math

    import numpy as np

    from tests import _shim

    L = _shim.lib()
    rnd = np.random.default_rng(2024)
    out = (C.c_int32 * 8)()
    listed = needed = 0
    for trial in range(3000):
        span = int(rnd.choice([3, 12, 40, 90]))
        p1 = (int(rnd.integers(-20, 60)), int(rnd.integers(-20, 60)))
        p2 = (p1[0] + int(rnd.integers(-span, span + 1)), p1[1] + int(rnd.integers(-span, span + 1)))
        if trial % 9 == 0:
            p2 = (p1[0] + int(rnd.integers(-span, span + 1)), p1[1])  # axis-parallel
        if trial % 11 == 0:
            d = int(rnd.integers(-span, span + 1))
            p2 = (p1[0] + d, p1[1] + d)  # exact diagonal
        if p1 == p2:
            continue
        hw = float(rnd.choice([0.0, 0.05, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0, 4.5, 7.3, -2.0]))
        runs = _literal_runs(p1, p2, hw)
        ln = math.sqrt(float((p2[0] - p1[0]) ** 2 + (p2[1] - p1[1]) ** 2))
        ft = max(abs(hw) + 0.5, 1.0)
        for _ in range(6):
            rx0, ry0 = int(rnd.integers(-40, 80)) & ~31, int(rnd.integers(-40, 80)) & ~15
            rx1, ry1 = rx0 + 31, ry0 + 15
            n = L.shim_seg_ranges(p1[0], p1[1], p2[0], p2[1], ln, ft, rx0, ry0, rx1, ry1, out)
            k_lo = (out[0], out[2]); k_n = (out[1], out[3]); m_lo = (out[4], out[6]); n_x = (out[5], out[7])
            assert n == sum(k_n) + sum(n_x)
            listed += n
            for kind, idx, side, pix in runs:
                if not any(rx0 <= x <= rx1 and ry0 <= y <= ry1 for x, y in pix):
                    continue
                needed += 1
                lo, cnt = (k_lo[side], k_n[side]) if kind == 0 else (m_lo[side], n_x[side])
                assert lo <= idx < lo + cnt, (p1, p2, hw, (rx0, ry0), kind, idx, side, list(out))
    assert needed > 25000, needed
    print(f"seg_ranges: {listed} items listed for {needed} runs that draw ({listed / needed:.2f}x)")
    # the cull is not required to be exact, but it should not list several times what draws
    assert listed < 2.5 * needed, (listed, needed)
