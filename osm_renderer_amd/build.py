"""Builds libosmtile.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libosmtile.so")
SOURCES = ["osmt_kernels.hip", "osmt_labels.hip", "osmt_pngenc.hip", "osmt_api.cpp", "osmt_png.cpp"]
HEADERS = ["osmt_geom.h", "osmt_internal.h", "osmt_png_table.h", os.path.join("..", "..", "include", "osmtile.h")]
# -ffp-contract=off: the reference never fuses a*b+c; its u8 output truncates, so an FMA flips pixels.
# zlib: PNG encoding of rendered tiles (osmt_png.cpp); --no-undefined: a symbol lost in an edit fails the build, not the first call
LIBS = ["-lz", "-ldl", "-lpthread", "-Wl,--no-undefined"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libosmtile.so cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines):
    """Experimental kernel variant: libosmtile_<name>.so built with extra -D flags (see tools/)."""
    out = os.path.join(HERE, f"libosmtile_{name}.so")
    cmd = [_hipcc()] + FLAGS + [f"-D{d}" for d in defines] + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES] + LIBS
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES] + LIBS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
