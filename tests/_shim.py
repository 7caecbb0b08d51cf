"""Builds and loads tests/geom_shim.cpp (host build of the kernels' closed forms)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgeomshim.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "geom_shim.cpp")
        hdr = os.path.join(_HERE, "..", "osm_renderer_amd", "csrc", "osmt_geom.h")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO, src])
        L = C.CDLL(_SO)
        ip = C.POINTER(C.c_int32)
        L.shim_fill_row_extent.argtypes = [C.c_int32] * 5 + [ip, ip]
        L.shim_fill_rows.argtypes = [C.c_int32] * 6 + [ip]
        L.shim_stroke_steps.argtypes = [C.c_int32, C.c_int32, ip]
        L.shim_stroke_steps24.argtypes = [C.c_int32, C.c_int32, ip]
        L.shim_extra_events.argtypes = [C.c_int32, C.c_int32, ip, ip]
        L.shim_extra_events.restype = C.c_int32
        L.shim_fmod_pos.argtypes = [C.c_double, C.c_double]
        L.shim_fmod_pos.restype = C.c_double
        dp = C.POINTER(C.c_double)
        L.shim_div_exact_check.argtypes = [dp, dp, C.c_size_t, dp]
        L.shim_div_exact_check.restype = C.c_size_t
        L.shim_div_exact_sweep.argtypes = [C.c_uint64, C.c_size_t, dp]
        L.shim_div_exact_sweep.restype = C.c_size_t
        L.shim_seg_ranges.argtypes = [C.c_int32] * 4 + [C.c_double, C.c_double] + [C.c_int32] * 4 + [ip]
        L.shim_seg_ranges.restype = C.c_uint32
        L.shim_udiv.argtypes = [C.c_int64, C.c_int64]
        L.shim_udiv.restype = C.c_int64
        L.shim_sizeof.argtypes = [C.c_int]
        L.shim_sizeof.restype = C.c_size_t
        _lib = L
    return _lib
