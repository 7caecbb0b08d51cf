"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under osm_renderer_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from osm_renderer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


class Icon(C.Structure):
    _fields_ = [("rgba", C.POINTER(C.c_double)), ("width", C.c_uint32), ("height", C.c_uint32)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "osm_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        u8p = C.POINTER(C.c_uint8)
        L.orc_coords_to_xy.argtypes = [C.c_double, C.c_double, C.c_uint8, dp, dp]
        L.orc_coords_to_xy_tile_relative.argtypes = [C.c_double, C.c_double, C.c_uint8, C.c_uint32, C.c_uint32, dp, dp]
        L.orc_project_points.argtypes = [dp, C.c_size_t, C.c_uint8, C.c_uint32, C.c_uint32, C.c_double, ip]
        L.orc_coords_to_max_zoom_tile.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_push_away_from.argtypes = [ip, ip, C.c_double, ip]
        L.orc_pixels_new.restype = C.c_void_p
        L.orc_pixels_new.argtypes = [C.c_uint32]
        L.orc_pixels_free.argtypes = [C.c_void_p]
        L.orc_pixels_reset.argtypes = [C.c_void_p, C.c_int, C.c_uint8, C.c_uint8, C.c_uint8]
        L.orc_set_pixel.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dp]
        L.orc_bump_generation.argtypes = [C.c_void_p]
        L.orc_blend_unfinished_pixels.argtypes = [C.c_void_p]
        L.orc_dimension.argtypes = [C.c_void_p]
        L.orc_dimension.restype = C.c_uint32
        L.orc_to_rgb_triples.argtypes = [C.c_void_p, u8p]
        L.orc_read_pixels_f64.argtypes = [C.c_void_p, dp]
        L.orc_read_pending_alpha.argtypes = [C.c_void_p, C.c_uint64, dp]
        L.orc_fill_contour.argtypes = [C.c_void_p, ip, C.c_size_t, u8p, C.POINTER(Icon), C.c_double]
        L.orc_draw_lines.argtypes = [
            C.c_void_p, ip, C.c_size_t, C.c_double, u8p, C.c_double, dp, C.c_int, C.c_int, C.c_int,
        ]
        L.orc_opacity_calculate.argtypes = [
            C.c_double, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, dp, C.POINTER(C.c_int),
        ]
        L.orc_fill_edge_walk.argtypes = [ip, ip, ip, C.c_size_t]
        L.orc_fill_edge_walk.restype = C.c_size_t
        L.orc_render_job.argtypes = [C.POINTER(abi.Batch), C.c_size_t, C.POINTER(Icon), C.c_size_t, u8p, dp]
        L.orc_render_batch.argtypes = [
            C.POINTER(abi.Batch), C.c_size_t, C.c_size_t, C.POINTER(Icon), C.c_size_t, u8p, C.c_size_t, C.c_int,
        ]
        L.orc_render_job_labels.argtypes = [
            C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), C.c_size_t, C.POINTER(Icon), C.c_size_t, u8p, dp, u8p,
        ]
        L.orc_render_batch_labels.argtypes = [
            C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), C.c_size_t, C.c_size_t, C.POINTER(Icon), C.c_size_t, u8p,
            C.c_size_t, C.c_int, u8p,
        ]
        L.orc_pool_create.restype = C.c_void_p
        L.orc_pool_create.argtypes = [C.c_int, C.c_uint32]
        L.orc_pool_free.argtypes = [C.c_void_p]
        L.orc_pool_render.argtypes = [
            C.c_void_p, C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), C.c_size_t, C.c_size_t, C.POINTER(Icon), C.c_size_t,
            u8p, C.c_size_t, u8p,
        ]
        L.orc_rasterizer_pixels.argtypes = [dp, C.c_size_t, ip, dp, C.c_size_t]
        L.orc_rasterizer_pixels.restype = C.c_size_t
        L.orc_flatten_quad.argtypes = [dp, dp, C.c_size_t]
        L.orc_flatten_quad.restype = C.c_size_t
        L.orc_job_points.argtypes = [C.POINTER(abi.Batch), C.c_size_t, ip]
        L.orc_composite.argtypes = [dp, dp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_int]
        L.orc_icon_from_rgba8.argtypes = [u8p, C.c_size_t, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


# ---- projection ------------------------------------------------------------
def coords_to_xy(lat, lon, zoom):
    x, y = C.c_double(), C.c_double()
    lib().orc_coords_to_xy(lat, lon, zoom, C.byref(x), C.byref(y))
    return x.value, y.value


def coords_to_max_zoom_tile(lat, lon):
    x, y = C.c_uint32(), C.c_uint32()
    lib().orc_coords_to_max_zoom_tile(lat, lon, C.byref(x), C.byref(y))
    return x.value, y.value


def project_points(latlon, zoom, tx, ty, scale):
    latlon = np.ascontiguousarray(latlon, dtype=np.float64).reshape(-1, 2)
    out = np.empty((len(latlon), 2), dtype=np.int32)
    lib().orc_project_points(_dp(latlon), len(latlon), zoom, tx, ty, float(scale), _ip(out))
    return out


def push_away_from(p, other, by):
    a = np.array(p, dtype=np.int32)
    b = np.array(other, dtype=np.int32)
    o = np.zeros(2, dtype=np.int32)
    lib().orc_push_away_from(_ip(a), _ip(b), float(by), _ip(o))
    return int(o[0]), int(o[1])


def fill_edge_walk(p1, p2, cap=1 << 16):
    a = np.array(p1, dtype=np.int32)
    b = np.array(p2, dtype=np.int32)
    out = np.empty((cap, 2), dtype=np.int32)
    n = lib().orc_fill_edge_walk(_ip(a), _ip(b), _ip(out), cap)
    assert n <= cap
    return out[:n].copy()


def opacity_calculate(half_width, dashes, cap, traveled, cd, sd):
    op, inl = C.c_double(), C.c_int()
    if dashes is None:
        lib().orc_opacity_calculate(half_width, None, -1, cap, traveled, cd, sd, C.byref(op), C.byref(inl))
    else:
        d = np.ascontiguousarray(dashes, dtype=np.float64)
        lib().orc_opacity_calculate(half_width, _dp(d), len(d), cap, traveled, cd, sd, C.byref(op), C.byref(inl))
    return op.value, bool(inl.value)


# ---- canvas ------------------------------------------------------------------
def make_icons(images):
    """images: list of (h, w, 4) uint8 straight-alpha arrays -> (ctypes Icon array, keepalive)."""
    if not images:
        return None, 0, []
    arr = (Icon * len(images))()
    keep = []
    for i, img in enumerate(images):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, _ = img.shape
        pm = np.empty((h * w, 4), dtype=np.float64)
        lib().orc_icon_from_rgba8(_u8p(img), h * w, _dp(pm))
        keep.append(pm)
        arr[i].rgba = _dp(pm)
        arr[i].width = w
        arr[i].height = h
    return arr, len(images), keep


class Pixels:
    """draw::tile_pixels::TilePixels (reference: src/draw/tile_pixels.rs)."""

    def __init__(self, scale=1):
        self._p = lib().orc_pixels_new(scale)
        self.dim = lib().orc_dimension(self._p)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_pixels_free(self._p)
            self._p = None

    def reset(self, canvas=(241, 238, 232)):
        if canvas is None:
            lib().orc_pixels_reset(self._p, 0, 0, 0, 0)
        else:
            lib().orc_pixels_reset(self._p, 1, *canvas)

    def set_pixel(self, x, y, rgba):
        c = np.array(rgba, dtype=np.float64)
        lib().orc_set_pixel(self._p, x, y, _dp(c))

    def bump_generation(self):
        lib().orc_bump_generation(self._p)

    def blend_unfinished_pixels(self):
        lib().orc_blend_unfinished_pixels(self._p)

    def fill_contour(self, pairs, color, opacity=1.0, icon=None):
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 4)
        col = np.array(color, dtype=np.uint8)
        ic = None
        if icon is not None:
            arr, _, keep = make_icons([icon])
            ic = C.pointer(arr[0])
            self._keep = (arr, keep)
        lib().orc_fill_contour(self._p, _ip(pairs), len(pairs), _u8p(col), ic, float(opacity))

    def draw_lines(self, pairs, width, color, opacity=1.0, dashes=None, cap=abi.CAP_NONE, use_caps_for_dashes=False):
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 4)
        col = np.array(color, dtype=np.uint8)
        if dashes is None:
            lib().orc_draw_lines(self._p, _ip(pairs), len(pairs), float(width), _u8p(col), float(opacity), None, -1,
                                 cap, int(use_caps_for_dashes))
        else:
            d = np.ascontiguousarray(dashes, dtype=np.float64)
            lib().orc_draw_lines(self._p, _ip(pairs), len(pairs), float(width), _u8p(col), float(opacity), _dp(d),
                                 len(d), cap, int(use_caps_for_dashes))

    def to_rgb(self):
        out = np.empty((self.dim, self.dim, 3), dtype=np.uint8)
        lib().orc_to_rgb_triples(self._p, _u8p(out))
        return out

    def pixels_f64(self):
        out = np.empty((self.dim, self.dim, 4), dtype=np.float64)
        lib().orc_read_pixels_f64(self._p, _dp(out))
        return out

    def pending_alpha(self, gen=0):
        out = np.empty((self.dim, self.dim), dtype=np.float64)
        lib().orc_read_pending_alpha(self._p, gen, _dp(out))
        return out


def ring_to_pairs(ring):
    """point_pairs.rs:11-22: consecutive (P[i-1], P[i])."""
    ring = np.asarray(ring, dtype=np.int32).reshape(-1, 2)
    return np.concatenate([ring[:-1], ring[1:]], axis=1)


# ---- display lists -------------------------------------------------------------
def render_job(dl, job_idx=0, images=(), want_f64=False, labels=None, want_status=False):
    """labels: osm_renderer_amd.labels.LabelList covering the same jobs (drawer.rs:107-125), or None."""
    b = dl.as_batch()
    arr, n, keep = make_icons(list(images))
    out = np.empty((dl.dim, dl.dim, 4), dtype=np.uint8)
    f64 = np.empty((dl.dim, dl.dim, 4), dtype=np.float64) if want_f64 else None
    lb = labels.as_batch() if labels is not None else None
    status = np.zeros(len(labels.labels) if labels is not None else 0, dtype=np.uint8)
    rc = lib().orc_render_job_labels(C.byref(b), C.byref(lb) if lb is not None else None, job_idx, arr, n, _u8p(out),
                                     _dp(f64) if want_f64 else None, _u8p(status) if len(status) else None)
    assert rc == 0
    res = (out, f64) if want_f64 else out
    return (res, status) if want_status else res


def render_batch(dl, first=0, count=None, images=(), threads=1, labels=None, want_status=False):
    count = dl.n_jobs - first if count is None else count
    b = dl.as_batch()
    arr, n, keep = make_icons(list(images))
    out = np.empty((count, dl.dim, dl.dim, 4), dtype=np.uint8)
    lb = labels.as_batch() if labels is not None else None
    status = np.zeros(len(labels.labels) if labels is not None else 0, dtype=np.uint8)
    rc = lib().orc_render_batch_labels(C.byref(b), C.byref(lb) if lb is not None else None, first, count, arr, n, _u8p(out),
                                       dl.dim * dl.dim * 4, threads, _u8p(status) if len(status) else None)
    assert rc == 0
    return (out, status) if want_status else out


class Pool:
    """Persistent worker pool of the oracle: `threads` TilePixels allocated once, like the reference's server
    (http_server.rs:69-72).  render() writes into a caller-provided array and allocates nothing."""

    def __init__(self, threads, scale=1):
        self.threads = int(threads)
        self.scale = int(scale)
        self._p = lib().orc_pool_create(self.threads, self.scale)

    def close(self):
        if getattr(self, "_p", None):
            lib().orc_pool_free(self._p)
            self._p = None

    def __del__(self):
        self.close()

    def render(self, dl, out, first=0, count=None, labels=None, status=None, images=()):
        count = dl.n_jobs - first if count is None else count
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.shape[0] >= count
        b = dl.as_batch()
        arr, n, keep = make_icons(list(images))
        lb = labels.as_batch() if labels is not None else None
        rc = lib().orc_pool_render(self._p, C.byref(b), C.byref(lb) if lb is not None else None, first, count, arr, n,
                                   _u8p(out), dl.dim * dl.dim * 4, _u8p(status) if status is not None else None)
        assert rc == 0, rc
        return out


def rasterizer_pixels(segs):
    """font/rasterizer.rs draw_line* + save_to_figure: (xy [n][2] int32, total [n]) in visiting order."""
    segs = np.ascontiguousarray(segs, dtype=np.float64).reshape(-1, 4)
    cap = 1 << 16
    while True:
        xy = np.empty((cap, 2), dtype=np.int32)
        tot = np.empty(cap, dtype=np.float64)
        n = lib().orc_rasterizer_pixels(_dp(segs), len(segs), _ip(xy), _dp(tot), cap)
        if n <= cap:
            return xy[:n], tot[:n]
        cap = n


def flatten_quad(x0, y0, x1, y1, x2, y2):
    q = np.array([x0, y0, x1, y1, x2, y2], dtype=np.float64)
    cap = 1 << 14
    out = np.empty((cap, 4), dtype=np.float64)
    n = lib().orc_flatten_quad(_dp(q), _dp(out), cap)
    assert n <= cap
    return out[:n]


def job_points(dl, job_idx=0):
    b = dl.as_batch()
    n = int(dl.jobs[job_idx]["n_pts"])
    out = np.empty((n, 2), dtype=np.int32)
    lib().orc_job_points(C.byref(b), job_idx, _ip(out))
    return out


def composite(planes, canvas_rgba, threads=1):
    planes = np.ascontiguousarray(planes, dtype=np.float64)
    n, L, H, W, four = planes.shape
    assert four == 4
    cv = np.ascontiguousarray(canvas_rgba, dtype=np.float64)
    out = np.empty((n, H, W, 4), dtype=np.uint8)
    lib().orc_composite(_dp(planes), _dp(cv), n, L, W, H, _u8p(out), threads)
    return out
