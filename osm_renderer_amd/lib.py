"""ctypes binding of libosmtile.so — the C ABI of include/osmtile.h.

There is no fallback: if the HIP library is missing or no GPU is present the
calls raise.  torch is imported first on purpose: it brings its own
libamdhip64.so.7 into the process and the dynamic linker then resolves our
NEEDED entry of the same soname to that copy, so device pointers and streams
are shared between torch tensors and the kernels.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSMT_LIB selects an alternative build of the same library (kernel-variant experiments only)
LIB_PATH = os.environ.get("OSMT_LIB") or os.path.join(_HERE, "libosmtile.so")
_lib = None

EXPORTS = [
    "osmt_create", "osmt_destroy", "osmt_last_error", "osmt_version", "osmt_register_image", "osmt_render_batch",
    "osmt_scene_upload", "osmt_scene_free", "osmt_render_scene", "osmt_render_scene_f64", "osmt_render_scene_stages",
    "osmt_scene_read_points", "osmt_project", "osmt_composite", "osmt_composite_device", "osmt_png_bound",
    "osmt_encode_png", "osmt_render_batch_labels", "osmt_render_batch_rgb", "osmt_scene_set_labels", "osmt_scene_read_label_status",
    "osmt_scene_check", "osmt_worker_create", "osmt_worker_destroy", "osmt_worker_render",
    "osmt_render_batch_png_begin", "osmt_render_batch_png_end",
    "osmt_host_alloc", "osmt_host_free", "osmt_png_device_bound", "osmt_encode_png_device", "osmt_render_batch_png",
    "osmt_validate_batch", "osmt_batch_shard_create", "osmt_batch_shard_get", "osmt_batch_shard_free", "osmt_render_batch_multi",
    "osmt_render_batch_multi_ex",
    "osmt_comm_unique_id", "osmt_comm_init_rank", "osmt_comm_init_local", "osmt_allreduce_tile_count",
    "osmt_allreduce_tile_count_local", "osmt_allreduce_tile_count_enqueue", "osmt_allreduce_tile_count_result", "osmt_hbm_copy_probe",
    "osmt_debug_poison_enabled",
]


class OsmtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"osmtile error {code}: {msg}")
        self.code = code


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m osm_renderer_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback"
        )
    import torch  # noqa: F401  (loads the HIP runtime this library binds to)

    L = C.CDLL(LIB_PATH)
    vp, dp, ip, u8p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    L.osmt_version.restype = C.c_uint32
    L.osmt_last_error.restype = C.c_char_p
    L.osmt_create.argtypes = [C.POINTER(abi.Config), C.POINTER(vp)]
    L.osmt_destroy.argtypes = [vp]
    L.osmt_destroy.restype = None
    L.osmt_register_image.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.osmt_render_batch.argtypes = [vp, C.POINTER(abi.Batch), u8p, C.c_size_t]
    L.osmt_validate_batch.argtypes = [C.POINTER(abi.Batch)]
    L.osmt_batch_shard_create.argtypes = [C.POINTER(abi.Batch), C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.osmt_batch_shard_get.argtypes = [vp]
    L.osmt_batch_shard_get.restype = C.POINTER(abi.Batch)
    L.osmt_batch_shard_free.argtypes = [vp]
    L.osmt_batch_shard_free.restype = None
    L.osmt_render_batch_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(abi.Batch), u8p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.osmt_render_batch_multi_ex.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), C.c_uint32, u8p,
                                             C.c_size_t, C.POINTER(C.c_uint64)]
    L.osmt_comm_unique_id.argtypes = [u8p]
    L.osmt_comm_init_rank.argtypes = [vp, u8p, C.c_uint32, C.c_uint32]
    L.osmt_comm_init_local.argtypes = [C.POINTER(vp), C.c_uint32]
    L.osmt_allreduce_tile_count.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.osmt_allreduce_tile_count_enqueue.argtypes = [vp, C.c_uint64, vp]
    L.osmt_allreduce_tile_count_result.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    L.osmt_allreduce_tile_count_local.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.osmt_hbm_copy_probe.argtypes = [vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.osmt_scene_upload.argtypes = [vp, C.POINTER(abi.Batch), C.POINTER(vp)]
    L.osmt_scene_free.argtypes = [vp]
    L.osmt_scene_free.restype = None
    L.osmt_render_scene.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.osmt_render_scene_f64.argtypes = [vp, vp, vp, vp]
    L.osmt_render_scene_stages.argtypes = [vp, vp, C.c_uint32, vp, C.c_size_t, vp]
    L.osmt_scene_read_points.argtypes = [vp, vp, ip]
    L.osmt_project.argtypes = [vp, dp, C.c_size_t, C.c_uint8, C.c_uint32, C.c_uint32, C.c_double, ip]
    L.osmt_composite.argtypes = [vp, dp, dp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
    L.osmt_composite_device.argtypes = [vp, vp, dp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.osmt_render_batch_labels.argtypes = [vp, C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), u8p, C.c_size_t]
    L.osmt_render_batch_rgb.argtypes = [vp, C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), u8p, C.c_size_t]
    L.osmt_scene_set_labels.argtypes = [vp, vp, C.POINTER(abi.LabelBatch)]
    L.osmt_scene_read_label_status.argtypes = [vp, vp, u8p]
    if hasattr(L, "osmt_scene_check"):  # absent only from older variant builds loaded through OSMT_LIB (A/B timing runs)
        L.osmt_scene_check.argtypes = [vp, vp]
        L.osmt_worker_create.argtypes = [vp, C.POINTER(vp)]
        L.osmt_worker_destroy.argtypes = [vp]
        L.osmt_worker_destroy.restype = None
        L.osmt_worker_render.argtypes = [vp, C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), u8p, C.c_size_t]
    if hasattr(L, "osmt_render_batch_png_begin"):
        L.osmt_render_batch_png_begin.argtypes = [vp, C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), C.POINTER(vp)]
        L.osmt_render_batch_png_end.argtypes = [vp, u8p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.osmt_png_device_bound.argtypes = [C.c_uint32, C.c_uint32]
    L.osmt_png_device_bound.restype = C.c_size_t
    L.osmt_encode_png_device.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp, vp]
    L.osmt_render_batch_png.argtypes = [vp, C.POINTER(abi.Batch), C.POINTER(abi.LabelBatch), u8p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.osmt_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.osmt_host_free.argtypes = [vp, vp]
    L.osmt_host_free.restype = None
    L.osmt_png_bound.argtypes = [C.c_uint32, C.c_uint32]
    L.osmt_png_bound.restype = C.c_size_t
    L.osmt_encode_png.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    _lib = L
    return L


def check(rc):
    if rc != abi.OK:
        raise OsmtError(rc, load().osmt_last_error().decode("utf-8", "replace"))
