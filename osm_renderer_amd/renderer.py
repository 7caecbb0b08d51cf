"""Host-side driver over the C ABI: contexts, HBM-resident scenes, device output.

torch is used for device buffers / streams only (plumbing): every pixel is
produced by the HIP kernels behind libosmtile.so.
"""
import ctypes as C

import numpy as np

from . import abi
from .display_list import DisplayList
from .lib import check, load


def _torch():
    import torch

    return torch


def _stream_ptr(stream=None):
    torch = _torch()
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)


class Scene:
    """A display-list batch resident in HBM (osmt_scene)."""

    def __init__(self, ctx, dl: DisplayList, labels=None):
        self.ctx = ctx
        self.dl = dl
        self.n_jobs = dl.n_jobs
        self.dim = dl.dim
        b = dl.as_batch()
        h = C.c_void_p()
        check(load().osmt_scene_upload(ctx._h, C.byref(b), C.byref(h)))
        self._h = h
        self.labels = None
        if labels is not None:
            self.set_labels(labels)

    def set_labels(self, labels):
        """osmt_scene_set_labels: attach (or, with None, detach) the label pass of every tile."""
        if labels is None:
            check(load().osmt_scene_set_labels(self.ctx._h, self._h, None))
        else:
            assert labels.n_jobs == self.n_jobs
            lb = labels.as_batch()
            check(load().osmt_scene_set_labels(self.ctx._h, self._h, C.byref(lb)))
        self.labels = labels

    def label_status(self):
        """label_generation_statuses of the last render (tile_pixels.rs:160-162)."""
        n = len(self.labels.labels) if self.labels is not None else 0
        out = np.zeros(n, dtype=np.uint8)
        if n:
            check(load().osmt_scene_read_label_status(self.ctx._h, self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def check(self):
        """osmt_scene_check: waits for the scene's launches, raises if a kernel reported an internal error."""
        check(load().osmt_scene_check(self.ctx._h, self._h))

    def free(self):
        if getattr(self, "_h", None):
            load().osmt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Worker:
    """osmt_worker: the per-thread request handle of the reference's server loop (http_server.rs:50-83); concurrent
    render() calls of the workers of one context are gathered into shared launches."""

    def __init__(self, ctx):
        h = C.c_void_p()
        check(load().osmt_worker_create(ctx._h, C.byref(h)))
        self._h = h
        self.ctx = ctx

    def render(self, dl: DisplayList, labels=None, out=None):
        """osmt_worker_render: packed RGB8 [n, W*W*3] of the display list's tiles (usually one)."""
        b = dl.as_batch()
        tight = dl.dim * dl.dim * 3
        if out is None:
            out = np.empty((dl.n_jobs, tight), dtype=np.uint8)
        lb = labels.as_batch() if labels is not None else None
        check(load().osmt_worker_render(self._h, C.byref(b), C.byref(lb) if lb is not None else None,
                                        out.ctypes.data_as(C.POINTER(C.c_uint8)), tight))
        return out

    def close(self):
        if getattr(self, "_h", None):
            load().osmt_worker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PngJob:
    """A begun osmt_render_batch_png job.  Holds the display list (and labels) alive — the uploads are stream-ordered —
    and the native handle exactly once: png_end() takes it (the native call frees the job whatever it returns), a second
    png_end() raises instead of touching freed memory, and a job that is dropped without being ended is ended here with an
    empty buffer, which releases its device buffers (about 0.9 GB per 1024 tiles)."""

    def __init__(self, h, dl, labels, b, lb):
        self._h, self.dl, self.labels, self._b, self._lb = h, dl, labels, b, lb

    def take(self):
        if self._h is None:
            raise RuntimeError("this PNG job has already been ended (osmt_render_batch_png_end frees it whatever it returns)")
        h, self._h = self._h, None
        return h

    def abort(self):
        """Ends the job without reading the files back (frees its device buffers)."""
        if self._h is not None:
            h, self._h = self._h, None
            off = (C.c_uint64 * (self.dl.n_jobs + 1))()
            load().osmt_render_batch_png_end(h, None, 0, off)

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


class Context:
    """One GPU (osmt_ctx): analogue of the reference's Drawer + per-worker TilePixels."""

    def worker(self):
        return Worker(self)

    def __init__(self, device=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the MI355X raster path has no CPU fallback")
        L = load()
        cfg = abi.Config(device=device, flags=0)
        h = C.c_void_p()
        check(L.osmt_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)

    def close(self):
        if getattr(self, "_h", None):
            load().osmt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- icons -------------------------------------------------------------------
    def register_image(self, rgba8):
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        h, w, four = img.shape
        assert four == 4
        out = C.c_uint32()
        check(load().osmt_register_image(self._h, img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, C.byref(out)))
        return out.value

    # -- whole path --------------------------------------------------------------
    def upload(self, dl: DisplayList, labels=None) -> Scene:
        return Scene(self, dl, labels)

    def render(self, scene: Scene, out=None, stream=None):
        """osmt_render_scene: returns a uint8 cuda tensor [n, H, W, 4] (asynchronous)."""
        torch = _torch()
        if out is None:
            out = torch.empty((scene.n_jobs, scene.dim, scene.dim, 4), dtype=torch.uint8, device=self.device)
        stride = scene.dim * scene.dim * 4
        check(load().osmt_render_scene(self._h, scene._h, C.c_void_p(out.data_ptr()), stride, _stream_ptr(stream)))
        return out

    def render_stages(self, scene: Scene, stage_mask, out=None, stream=None):
        stride = scene.dim * scene.dim * 4
        ptr = C.c_void_p(out.data_ptr()) if out is not None else C.c_void_p(0)
        check(load().osmt_render_scene_stages(self._h, scene._h, stage_mask, ptr, stride, _stream_ptr(stream)))
        return out

    def render_f64(self, scene: Scene, stream=None):
        """osmt_render_scene_f64: premultiplied f64 canvas [n, H, W, 4]."""
        torch = _torch()
        out = torch.empty((scene.n_jobs, scene.dim, scene.dim, 4), dtype=torch.float64, device=self.device)
        check(load().osmt_render_scene_f64(self._h, scene._h, C.c_void_p(out.data_ptr()), _stream_ptr(stream)))
        return out

    def read_points(self, scene: Scene):
        out = np.empty((len(scene.dl.coords), 2), dtype=np.int32)
        check(load().osmt_scene_read_points(self._h, scene._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def host_alloc(self, shape, dtype=np.uint8):
        """osmt_host_alloc: a pinned numpy array (free it with host_free)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(load().osmt_host_alloc(self._h, n, C.byref(p)))
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data)
        load().osmt_host_free(self._h, p)

    def render_batch_host(self, dl: DisplayList, labels=None, out=None):
        """osmt_render_batch / osmt_render_batch_labels: host buffers in, host RGBA8 out (`out`: e.g. host_alloc())."""
        b = dl.as_batch()
        if out is None:
            out = np.empty((dl.n_jobs, dl.dim, dl.dim, 4), dtype=np.uint8)
        assert out.shape == (dl.n_jobs, dl.dim, dl.dim, 4) and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
        if labels is None:
            check(load().osmt_render_batch(self._h, C.byref(b), out.ctypes.data_as(C.POINTER(C.c_uint8)), dl.dim * dl.dim * 4))
        else:
            lb = labels.as_batch()
            check(load().osmt_render_batch_labels(self._h, C.byref(b), C.byref(lb), out.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                  dl.dim * dl.dim * 4))
        return out

    def render_batch_rgb(self, dl: DisplayList, labels=None, out=None, stride=None):
        """osmt_render_batch_rgb: host buffers in, packed RGB8 out (the reference's RgbTriples layout)."""
        b = dl.as_batch()
        tight = dl.dim * dl.dim * 3
        stride = tight if stride is None else stride
        if out is None:
            out = np.empty((dl.n_jobs, stride), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size >= dl.n_jobs * stride
        lb = labels.as_batch() if labels is not None else None
        check(load().osmt_render_batch_rgb(self._h, C.byref(b), C.byref(lb) if lb is not None else None,
                                           out.ctypes.data_as(C.POINTER(C.c_uint8)), stride))
        return out

    # -- PNG files from the GPU ----------------------------------------------------
    def encode_png_device(self, rgba, stream=None):
        """osmt_encode_png_device on a uint8 cuda tensor [n, H, W, 4]: (slots uint8 [n, bound], lengths int32 [n])."""
        torch = _torch()
        assert rgba.is_cuda and rgba.dtype == torch.uint8 and rgba.is_contiguous()
        n, H, W, four = rgba.shape
        assert four == 4
        bound = load().osmt_png_device_bound(W, H)
        slots = torch.empty((n, bound), dtype=torch.uint8, device=rgba.device)
        lens = torch.empty((n,), dtype=torch.int32, device=rgba.device)
        check(load().osmt_encode_png_device(self._h, C.c_void_p(rgba.data_ptr()), H * W * 4, n, W, H, C.c_void_p(slots.data_ptr()), bound,
                                            C.c_void_p(lens.data_ptr()), _stream_ptr(stream)))
        return slots, lens

    def render_batch_png(self, dl: DisplayList, labels=None, out=None, as_bytes=True):
        """osmt_render_batch_png: list of PNG files (bytes), one per tile.  `out`: uint8 buffer (e.g. host_alloc) to
        receive the files back to back; as_bytes=False returns (out, offsets) without copying."""
        b = dl.as_batch()
        lb = labels.as_batch() if labels is not None else None
        off = np.zeros(dl.n_jobs + 1, dtype=np.uint64)
        if out is None:
            out = np.empty(dl.n_jobs * load().osmt_png_device_bound(dl.dim, dl.dim), dtype=np.uint8)
        cap = out.size
        check(load().osmt_render_batch_png(self._h, C.byref(b), C.byref(lb) if lb is not None else None,
                                           out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, off.ctypes.data_as(C.POINTER(C.c_uint64))))
        if not as_bytes:
            return out, off
        return [out[int(off[i]) : int(off[i + 1])].tobytes() for i in range(dl.n_jobs)]

    def png_begin(self, dl: DisplayList, labels=None):
        """osmt_render_batch_png_begin: queues the whole batch, returns a job (keeps `dl` / `labels` alive until png_end)."""
        b = dl.as_batch()
        lb = labels.as_batch() if labels is not None else None
        h = C.c_void_p()
        check(load().osmt_render_batch_png_begin(self._h, C.byref(b), C.byref(lb) if lb is not None else None, C.byref(h)))
        return PngJob(h, dl, labels, b, lb)

    def png_end(self, job, out, as_bytes=False):
        """osmt_render_batch_png_end: (out, offsets) — or the list of files with as_bytes=True.

        The native call frees the job WHATEVER it returns (include/osmtile.h): a job can be ended once.  A buffer that
        turns out too small therefore needs a new png_begin — size `out` with osmt_png_device_bound x tiles and it cannot."""
        h, dl = job.take(), job.dl
        off = np.zeros(dl.n_jobs + 1, dtype=np.uint64)
        check(load().osmt_render_batch_png_end(h, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size, off.ctypes.data_as(C.POINTER(C.c_uint64))))
        if not as_bytes:
            return out, off
        return [out[int(off[i]) : int(off[i + 1])].tobytes() for i in range(dl.n_jobs)]

    def hbm_copy_probe(self, nbytes=1 << 30, iters=20):
        """osmt_hbm_copy_probe: (copy GB/s counting read + write, read-only GB/s) of a 16-byte-per-lane device stream."""
        cp, rd = C.c_double(0.0), C.c_double(0.0)
        check(load().osmt_hbm_copy_probe(self._h, int(nbytes), int(iters), C.byref(cp), C.byref(rd)))
        return float(cp.value), float(rd.value)

    # -- stages --------------------------------------------------------------------
    def project(self, latlon, zoom, tx, ty, scale=1.0):
        latlon = np.ascontiguousarray(latlon, dtype=np.float64).reshape(-1, 2)
        out = np.empty((len(latlon), 2), dtype=np.int32)
        check(
            load().osmt_project(
                self._h, latlon.ctypes.data_as(C.POINTER(C.c_double)), len(latlon), zoom, tx, ty, float(scale),
                out.ctypes.data_as(C.POINTER(C.c_int32)),
            )
        )
        return out

    def composite_host(self, planes, canvas_rgba):
        planes = np.ascontiguousarray(planes, dtype=np.float64)
        n, L, H, W, four = planes.shape
        assert four == 4
        cv = np.ascontiguousarray(canvas_rgba, dtype=np.float64)
        out = np.empty((n, H, W, 4), dtype=np.uint8)
        check(
            load().osmt_composite(
                self._h, planes.ctypes.data_as(C.POINTER(C.c_double)), cv.ctypes.data_as(C.POINTER(C.c_double)), n, L,
                W, H, out.ctypes.data_as(C.POINTER(C.c_uint8)),
            )
        )
        return out

    def composite(self, planes, canvas_rgba, out=None, stream=None):
        """osmt_composite_device on a cuda float64 tensor [n, L, H, W, 4]."""
        torch = _torch()
        assert planes.is_cuda and planes.dtype == torch.float64 and planes.is_contiguous()
        n, L, H, W, four = planes.shape
        assert four == 4
        if out is None:
            out = torch.empty((n, H, W, 4), dtype=torch.uint8, device=planes.device)
        cv = (C.c_double * 4)(*[float(v) for v in canvas_rgba])
        check(
            load().osmt_composite_device(
                self._h, C.c_void_p(planes.data_ptr()), cv, n, L, W, H, C.c_void_p(out.data_ptr()), _stream_ptr(stream)
            )
        )
        return out


def encode_png(rgba, level=-1):
    """osmt_encode_png: RGB8 PNG bytes of one RGBA8 tile [H, W, 4] (png_writer.rs:4-21)."""
    img = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, four = img.shape
    assert four == 4
    L = load()
    cap = L.osmt_png_bound(w, h)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t()
    check(L.osmt_encode_png(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, w * 4, level,
                            out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()
