"""Display lists: the flat, pointer-free form of what Drawer::draw_to_pixels walks.

A reference tile is `reset(canvas)` followed by an ordered list of
`draw_one_area` calls (reference: src/draw/drawer.rs:60-131,156-219); here each
call is one 64-byte op (include/osmtile.h `osmt_op`) referencing rings of points
in shared pools.  numpy structured arrays mirror the C structs exactly so a
batch is handed to the C ABI without copying.
"""
import ctypes as C

import numpy as np

from . import abi

OP_DTYPE = np.dtype(
    [
        ("kind", "u1"),
        ("cap", "u1"),
        ("use_caps_for_dashes", "u1"),
        ("has_dashes", "u1"),
        ("color", "u1", (3,)),
        ("_pad0", "u1"),
        ("opacity", "f8"),
        ("width", "f8"),
        ("n_dashes", "u4"),
        ("dashes_off", "u4"),
        ("n_rings", "u4"),
        ("ring_off", "u4"),
        ("image_id", "u4"),
        ("_reserved", "u4", (5,)),
    ]
)
RING_DTYPE = np.dtype([("first_pt", "u4"), ("n_pts", "u4")])
JOB_DTYPE = np.dtype(
    [
        ("x", "u4"),
        ("y", "u4"),
        ("zoom", "u1"),
        ("has_canvas", "u1"),
        ("canvas_rgb", "u1", (3,)),
        ("_pad", "u1", (3,)),
        ("n_ops", "u4"),
        ("op_off", "u4"),
        ("n_pts", "u4"),
        ("pt_off", "u4"),
    ]
)
assert OP_DTYPE.itemsize == 64 and RING_DTYPE.itemsize == 8 and JOB_DTYPE.itemsize == 32


class DisplayList:
    """A batch of tiles (osmt_batch) backed by numpy arrays."""

    def __init__(self, jobs, ops, rings, coords, dashes, coord_kind, scale, nodes=None):
        self.jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        self.ops = np.ascontiguousarray(ops, dtype=OP_DTYPE)
        self.rings = np.ascontiguousarray(rings, dtype=RING_DTYPE)
        self.coord_kind = int(coord_kind)
        self.scale = int(scale)
        self.nodes = None
        if self.coord_kind == abi.COORD_LATLON_F64:
            self.coords = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, 2)
        elif self.coord_kind == abi.COORD_NODE_REF:
            self.coords = np.ascontiguousarray(coords, dtype=np.uint32).reshape(-1)  # node references
            self.nodes = np.ascontiguousarray(nodes, dtype=np.float64).reshape(-1, 2)
        else:
            self.coords = np.ascontiguousarray(coords, dtype=np.int32).reshape(-1, 2)
        self.dashes = np.ascontiguousarray(dashes, dtype=np.float64).reshape(-1)

    @property
    def n_jobs(self):
        return len(self.jobs)

    @property
    def dim(self):
        return abi.TILE_SIZE * self.scale

    def as_batch(self):
        """ctypes osmt_batch pointing into this object's arrays (keep `self` alive)."""
        b = abi.Batch()
        b.jobs = self.jobs.ctypes.data_as(C.POINTER(abi.TileJob))
        b.n_jobs = len(self.jobs)
        b.ops = self.ops.ctypes.data_as(C.POINTER(abi.Op))
        b.n_ops = len(self.ops)
        b.rings = self.rings.ctypes.data_as(C.POINTER(abi.Ring))
        b.n_rings = len(self.rings)
        b.coord_kind = self.coord_kind
        b.scale = self.scale
        if self.coord_kind == abi.COORD_LATLON_F64:
            b.latlon = self.coords.ctypes.data_as(C.POINTER(C.c_double))
            b.points = None
        elif self.coord_kind == abi.COORD_NODE_REF:
            b.latlon = None
            b.points = None
            b.nodes = self.nodes.ctypes.data_as(C.POINTER(C.c_double))
            b.n_nodes = len(self.nodes)
            b.node_refs = self.coords.ctypes.data_as(C.POINTER(C.c_uint32))
        else:
            b.latlon = None
            b.points = self.coords.ctypes.data_as(C.POINTER(C.c_int32))
        b.n_pts = len(self.coords)
        b.dashes = self.dashes.ctypes.data_as(C.POINTER(C.c_double)) if len(self.dashes) else None
        b.n_dashes = len(self.dashes)
        return b

    def algorithmic_bytes(self):
        """B_raster = 16*N_pts + 64*N_ops + 8*N_dashes + 4*W*H per tile, summed."""
        pt_bytes = {abi.COORD_LATLON_F64: 16, abi.COORD_NODE_REF: 4}.get(self.coord_kind, 8)
        return (
            (16 * len(self.nodes) if self.nodes is not None else 0)
            + pt_bytes * len(self.coords)
            + 64 * len(self.ops)
            + 8 * len(self.dashes)
            + 4 * self.dim * self.dim * len(self.jobs)
        )

    def with_node_refs(self):
        """The same tiles with OSMT_COORD_NODE_REF coordinates: identical (lat, lon) pairs — nodes shared by the
        rings of a way's fill/casing/stroke ops and by neighbouring tiles — are stored once."""
        assert self.coord_kind == abi.COORD_LATLON_F64
        nodes, refs = np.unique(self.coords.view([("lat", "f8"), ("lon", "f8")]).reshape(-1), return_inverse=True)
        return DisplayList(self.jobs, self.ops, self.rings, refs.astype(np.uint32), self.dashes, abi.COORD_NODE_REF, self.scale,
                           nodes=nodes.view(np.float64).reshape(-1, 2))

    def subset(self, idx):
        """A new DisplayList holding jobs `idx` (pools are re-packed)."""
        return concat([self._single(i) for i in idx])

    def _single(self, i):
        j = self.jobs[i]
        ops = self.ops[j["op_off"] : j["op_off"] + j["n_ops"]].copy()
        rings_l, dashes_l = [], []
        ring_cursor = dash_cursor = 0
        for o in ops:
            r = self.rings[o["ring_off"] : o["ring_off"] + o["n_rings"]].copy()
            r["first_pt"] -= j["pt_off"]
            rings_l.append(r)
            o["ring_off"] = ring_cursor
            ring_cursor += len(r)
            d = self.dashes[o["dashes_off"] : o["dashes_off"] + o["n_dashes"]]
            dashes_l.append(d)
            o["dashes_off"] = dash_cursor if len(d) else 0
            dash_cursor += len(d)
        job = np.array([j], dtype=JOB_DTYPE)
        job["op_off"] = 0
        job["pt_off"] = 0
        rings = np.concatenate(rings_l) if rings_l else np.zeros(0, RING_DTYPE)
        dashes = np.concatenate(dashes_l) if dashes_l else np.zeros(0)
        coords = self.coords[j["pt_off"] : j["pt_off"] + j["n_pts"]]
        return DisplayList(job, ops, rings, coords, dashes, self.coord_kind, self.scale, nodes=self.nodes)


def concat(lists):
    """Concatenate DisplayLists (same coord_kind and scale), fixing up offsets."""
    lists = list(lists)
    assert lists, "nothing to concatenate"
    ck, sc = lists[0].coord_kind, lists[0].scale
    assert ck != abi.COORD_NODE_REF or all(dl.nodes is lists[0].nodes for dl in lists), "NODE_REF lists must share one node table"
    jobs, ops, rings, coords, dashes = [], [], [], [], []
    o_op = o_ring = o_pt = o_dash = 0
    for dl in lists:
        assert dl.coord_kind == ck and dl.scale == sc
        j = dl.jobs.copy()
        j["op_off"] += o_op
        j["pt_off"] += o_pt
        o = dl.ops.copy()
        o["ring_off"] += o_ring
        o["dashes_off"] += np.where(o["n_dashes"] > 0, o_dash, 0).astype(np.uint32)
        r = dl.rings.copy()
        r["first_pt"] += o_pt
        jobs.append(j)
        ops.append(o)
        rings.append(r)
        coords.append(dl.coords)
        dashes.append(dl.dashes)
        o_op += len(o)
        o_ring += len(r)
        o_pt += len(dl.coords)
        o_dash += len(dl.dashes)
    return DisplayList(
        np.concatenate(jobs), np.concatenate(ops), np.concatenate(rings), np.concatenate(coords), np.concatenate(dashes), ck, sc,
        nodes=lists[0].nodes,
    )


class TileBuilder:
    """Records the draw calls of ONE tile in order (the Python-side mirror of the
    four canvas calls of the reference: reset / fill_contour / draw_lines /
    bump_generation — SURVEY.md §8(b))."""

    def __init__(self, zoom=15, x=0, y=0, scale=1, canvas=(241, 238, 232), coord_kind=abi.COORD_POINT_I32):
        self.zoom, self.x, self.y, self.scale = zoom, x, y, scale
        self.canvas = canvas
        self.coord_kind = coord_kind
        self._ops, self._rings, self._pts, self._dashes = [], [], [], []

    def _add_rings(self, rings):
        off = len(self._rings)
        for ring in rings:
            ring = np.asarray(ring).reshape(-1, 2)
            self._rings.append((len(self._pts), len(ring)))
            self._pts.extend(map(tuple, ring.tolist()))
        return off, len(self._rings) - off

    def _op(self, **kw):
        op = np.zeros(1, OP_DTYPE)[0]
        for k, v in kw.items():
            op[k] = v
        self._ops.append(op)

    def fill(self, rings, color, opacity=1.0):
        """fill_contour(points, Filler::Color(color), opacity)  — fill.rs:16"""
        if len(rings) and np.asarray(rings[0]).ndim == 1:
            rings = [rings]
        off, n = self._add_rings(rings)
        self._op(kind=abi.OP_FILL_COLOR, color=color, opacity=opacity, n_rings=n, ring_off=off)

    def fill_image(self, rings, image_id, opacity=1.0):
        """fill_contour(points, Filler::Image(icon), _) — fill.rs:36-40"""
        if len(rings) and np.asarray(rings[0]).ndim == 1:
            rings = [rings]
        off, n = self._add_rings(rings)
        self._op(kind=abi.OP_FILL_IMAGE, opacity=opacity, n_rings=n, ring_off=off, image_id=image_id)

    def stroke(self, points, width, color, opacity=1.0, dashes=None, cap=abi.CAP_NONE, use_caps_for_dashes=False):
        """draw_lines(points, width, color, opacity, dashes, line_cap, use_caps_for_dashes) — line.rs:9.
        width and dashes are already multiplied by scale (drawer.rs:171-172,191,206)."""
        off, n = self._add_rings([points])
        d_off = len(self._dashes)
        nd = 0
        if dashes is not None:
            self._dashes.extend(float(d) for d in dashes)
            nd = len(dashes)
        self._op(
            kind=abi.OP_STROKE,
            cap=cap,
            use_caps_for_dashes=int(bool(use_caps_for_dashes)),
            has_dashes=int(dashes is not None),
            color=color,
            opacity=opacity,
            width=width,
            n_dashes=nd,
            dashes_off=d_off,
            n_rings=n,
            ring_off=off,
        )

    def stroke_again(self, width, color, opacity=1.0, dashes=None, cap=abi.CAP_NONE, use_caps_for_dashes=False):
        """A second draw_lines call over the rings of the PREVIOUS op (a casing and its stroke are drawn from one way's
        points, drawer.rs:163-216): the two ops share their rings in the pools."""
        prev = self._ops[-1]
        off, n = int(prev["ring_off"]), int(prev["n_rings"])
        d_off = len(self._dashes)
        nd = 0
        if dashes is not None:
            self._dashes.extend(float(d) for d in dashes)
            nd = len(dashes)
        self._op(kind=abi.OP_STROKE, cap=cap, use_caps_for_dashes=int(bool(use_caps_for_dashes)), has_dashes=int(dashes is not None), color=color,
                 opacity=opacity, width=width, n_dashes=nd, dashes_off=d_off, n_rings=n, ring_off=off)

    def nop(self):
        """An area whose style draws nothing still bumps the generation (drawer.rs:218)."""
        self._op(kind=abi.OP_NONE)

    def build(self):
        job = np.zeros(1, JOB_DTYPE)
        job["x"], job["y"], job["zoom"] = self.x, self.y, self.zoom
        if self.canvas is None:
            job["has_canvas"] = 0
        else:
            job["has_canvas"] = 1
            job["canvas_rgb"] = self.canvas
        job["n_ops"] = len(self._ops)
        job["n_pts"] = len(self._pts)
        ops = np.array(self._ops, dtype=OP_DTYPE) if self._ops else np.zeros(0, OP_DTYPE)
        rings = np.array(self._rings, dtype=RING_DTYPE) if self._rings else np.zeros(0, RING_DTYPE)
        if self.coord_kind == abi.COORD_LATLON_F64:
            pts = np.array(self._pts, dtype=np.float64).reshape(-1, 2)
        else:
            pts = np.array(self._pts, dtype=np.int32).reshape(-1, 2)
        return DisplayList(job, ops, rings, pts, np.array(self._dashes, dtype=np.float64), self.coord_kind, self.scale)
