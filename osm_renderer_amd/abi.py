"""ctypes mirror of include/osmtile.h (data layout only — no compute).

The struct layouts here must match the C header byte for byte; tests/test_abi.py
checks sizes/offsets against the library's own sizeof probes.
"""
import ctypes as C

TILE_SIZE = 256  # reference: src/tile.rs:6
MAX_ZOOM = 18  # reference: src/tile.rs:5
MAX_DASHES = 16

OK, INVALID_ARG, OOM, HIP_ERROR, UNSUPPORTED, NO_DEVICE, RCCL_ERROR = 0, -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128
MULTI_RGB8 = 1  # osmt_render_batch_multi_ex flags

OP_NONE, OP_FILL_COLOR, OP_FILL_IMAGE, OP_STROKE = 0, 1, 2, 3
CAP_NONE, CAP_BUTT, CAP_ROUND, CAP_SQUARE = 0, 1, 2, 3
COORD_LATLON_F64, COORD_POINT_I32, COORD_NODE_REF = 0, 1, 2

STAGE_PROJECT, STAGE_OPINFO, STAGE_RASTER = 1, 2, 4


class Op(C.Structure):
    _fields_ = [
        ("kind", C.c_uint8),
        ("cap", C.c_uint8),
        ("use_caps_for_dashes", C.c_uint8),
        ("has_dashes", C.c_uint8),
        ("color", C.c_uint8 * 3),
        ("_pad0", C.c_uint8),
        ("opacity", C.c_double),
        ("width", C.c_double),
        ("n_dashes", C.c_uint32),
        ("dashes_off", C.c_uint32),
        ("n_rings", C.c_uint32),
        ("ring_off", C.c_uint32),
        ("image_id", C.c_uint32),
        ("_reserved", C.c_uint32 * 5),
    ]


class Ring(C.Structure):
    _fields_ = [("first_pt", C.c_uint32), ("n_pts", C.c_uint32)]


class TileJob(C.Structure):
    _fields_ = [
        ("x", C.c_uint32),
        ("y", C.c_uint32),
        ("zoom", C.c_uint8),
        ("has_canvas", C.c_uint8),
        ("canvas_rgb", C.c_uint8 * 3),
        ("_pad", C.c_uint8 * 3),
        ("n_ops", C.c_uint32),
        ("op_off", C.c_uint32),
        ("n_pts", C.c_uint32),
        ("pt_off", C.c_uint32),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("jobs", C.POINTER(TileJob)),
        ("n_jobs", C.c_size_t),
        ("ops", C.POINTER(Op)),
        ("n_ops", C.c_size_t),
        ("rings", C.POINTER(Ring)),
        ("n_rings", C.c_size_t),
        ("coord_kind", C.c_uint32),
        ("scale", C.c_uint32),
        ("latlon", C.POINTER(C.c_double)),
        ("points", C.POINTER(C.c_int32)),
        ("n_pts", C.c_size_t),
        ("dashes", C.POINTER(C.c_double)),
        ("n_dashes", C.c_size_t),
        ("nodes", C.POINTER(C.c_double)),
        ("n_nodes", C.c_size_t),
        ("node_refs", C.POINTER(C.c_uint32)),
    ]


class Label(C.Structure):
    _fields_ = [
        ("has_icon", C.c_uint8),
        ("has_text", C.c_uint8),
        ("text_color", C.c_uint8 * 3),
        ("_pad", C.c_uint8 * 3),
        ("image_id", C.c_uint32),
        ("seg_off", C.c_uint32),
        ("n_segs", C.c_uint32),
        ("_reserved", C.c_uint32),
        ("icon_center_x", C.c_double),
        ("icon_center_y", C.c_double),
    ]


class LabelBatch(C.Structure):
    _fields_ = [
        ("labels", C.POINTER(Label)),
        ("n_labels", C.c_size_t),
        ("job_label_off", C.POINTER(C.c_uint32)),
        ("segs", C.POINTER(C.c_double)),
        ("n_segs", C.c_size_t),
    ]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32)]


assert C.sizeof(Op) == 64
assert C.sizeof(Ring) == 8
assert C.sizeof(TileJob) == 32
assert C.sizeof(Label) == 40
