"""Compiles tests/host_mirror_demo.cpp against libosmtile.so."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "host_mirror_demo")


def build_demo():
    src = os.path.join(ROOT, "tests", "host_mirror_demo.cpp")
    hdr = os.path.join(ROOT, "osm_renderer_amd", "host", "osmt_draw.hpp")
    libdir = os.path.join(ROOT, "osm_renderer_amd")
    lib = os.path.join(libdir, "libosmtile.so")
    assert os.path.exists(lib), "build libosmtile.so first (__graft_entry__.build())"
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(lib)):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.check_call(
            ["g++", "-O2", "-std=c++17", "-o", BIN, src, "-L" + libdir, "-losmtile", "-Wl,-rpath," + libdir,
             "-Wl,-rpath-link,/opt/rocm/lib"]
        )
    return BIN
