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
