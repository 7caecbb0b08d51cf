"""Test infrastructure for the geodata reader (osm_renderer_amd/host/osmt_geodata.hpp):
  * write_geodata(): a Python restatement of geodata::saver::save_to_internal_format (src/geodata/saver.rs:21-165,
    167-226) so that files in the reference's on-disk format can be produced here (the Rust importer cannot run);
  * Reader: ctypes over tests/geodata_shim.cpp."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "_build", "libgeodata_shim.so")


def build_shim():
    src = os.path.join(ROOT, "tests", "geodata_shim.cpp")
    hdr = os.path.join(ROOT, "osm_renderer_amd", "host", "osmt_geodata.hpp")
    if not os.path.exists(SHIM) or os.path.getmtime(SHIM) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(SHIM), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", SHIM, src])
    return SHIM


def write_geodata(path, nodes, ways=(), polygons=(), multipolygons=(), tile_refs=None, max_zoom_tile=None):
    """nodes: [(global_id, lat, lon, {tags})]; ways: [(global_id, [node idx], {tags})]; polygons: [[node idx]];
    multipolygons: [(global_id, [polygon idx], {tags})].  tile_refs: {(x, y): (node ids, way ids, multipolygon ids)}
    or None = get_tile_references (saver.rs:167-226) with max_zoom_tile(lat, lon) -> (x, y)."""
    ints, strings, str_off = [], bytearray(), {}

    def add_string(s):  # BufferedData::add_string (saver.rs:145-155)
        b = s.encode("utf-8")
        if s not in str_off:
            str_off[s] = len(strings)
            strings.extend(b)
        return str_off[s], len(b)

    def refs(seq):  # save_refs (saver.rs:111-122)
        off = len(ints)
        ints.extend(int(v) for v in seq)
        return struct.pack("<II", off, len(ints) - off)

    def tags(t):  # save_tags (saver.rs:124-136): BTreeMap order = sorted by key
        kv = []
        for k in sorted(t):
            kv.extend(add_string(k))
            kv.extend(add_string(t[k]))
        return refs(kv)

    out = bytearray()
    out += struct.pack("<I", len(nodes))
    for gid, lat, lon, t in nodes:
        out += struct.pack("<Qdd", gid, lat, lon) + tags(t)
    out += struct.pack("<I", len(ways))
    for gid, nids, t in ways:
        out += struct.pack("<Q", gid) + refs(nids) + tags(t)
    out += struct.pack("<I", len(polygons))
    for nids in polygons:
        out += refs(nids)
    out += struct.pack("<I", len(multipolygons))
    for gid, pids, t in multipolygons:
        out += struct.pack("<Q", gid) + refs(pids) + tags(t)
    if tile_refs is None:
        tile_refs = get_tile_references(nodes, ways, polygons, multipolygons, max_zoom_tile)
    out += struct.pack("<I", len(tile_refs))
    for (x, y) in sorted(tile_refs):  # BTreeMap<(u32, u32), _>
        n, w, m = tile_refs[(x, y)]
        out += struct.pack("<II", x, y) + refs(sorted(n)) + refs(sorted(w)) + refs(sorted(m))
    out += struct.pack("<I", len(ints)) + np.asarray(ints, dtype="<u4").tobytes() + bytes(strings)
    with open(path, "wb") as f:
        f.write(out)
    return tile_refs


def get_tile_references(nodes, ways, polygons, multipolygons, max_zoom_tile):
    """saver.rs:167-226: a node belongs to its z18 tile; a way / multipolygon to every z18 tile of the bounding
    range of its nodes' tiles."""
    res = {}

    def ref(x, y):
        return res.setdefault((x, y), (set(), set(), set()))

    tiles = [max_zoom_tile(lat, lon) for _, lat, lon, _ in nodes]
    for i, (x, y) in enumerate(tiles):
        ref(x, y)[0].add(i)

    def spread(node_ids, which, eid):
        node_ids = list(node_ids)
        if not node_ids:
            return
        xs = [tiles[n][0] for n in node_ids]
        ys = [tiles[n][1] for n in node_ids]
        for x in range(min(xs), max(xs) + 1):
            for y in range(min(ys), max(ys) + 1):
                ref(x, y)[which].add(eid)

    for i, (_, nids, _) in enumerate(ways):
        spread(nids, 1, i)
    for i, (_, pids, _) in enumerate(multipolygons):
        spread([n for p in pids for n in polygons[p]], 2, i)
    return res


class Reader:
    def __init__(self, path):
        L = C.CDLL(build_shim())
        u32p = C.POINTER(C.c_uint32)
        L.gd_load.restype = C.c_void_p
        L.gd_load.argtypes = [C.c_char_p]
        L.gd_free.argtypes = [C.c_void_p]
        L.gd_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.gd_tile_range.argtypes = [C.c_uint8, C.c_uint32, C.c_uint32, u32p]
        L.gd_query.restype = C.c_size_t
        L.gd_query.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint8, C.c_uint32, C.c_uint32, u32p, C.c_size_t]
        for f in (L.gd_way_nodes, L.gd_multipolygon_polygons, L.gd_polygon_nodes):
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_size_t, u32p, C.c_size_t]
        L.gd_way_is_closed.argtypes = [C.c_void_p, C.c_size_t]
        L.gd_global_id.restype = C.c_uint64
        L.gd_global_id.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.gd_node_table.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.gd_tag.restype = C.c_long
        L.gd_tag.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t]
        self.L = L
        self.h = L.gd_load(path.encode())
        if not self.h:
            raise RuntimeError("gd_load failed")
        c = (C.c_uint64 * 5)()
        L.gd_counts(self.h, c)
        self.n_nodes, self.n_ways, self.n_polygons, self.n_multipolygons, self.n_tiles = [int(v) for v in c]

    def close(self):
        if self.h:
            self.L.gd_free(self.h)
            self.h = None

    def _list(self, fn, *args):
        cap = 1 << 16
        while True:
            buf = (C.c_uint32 * cap)()
            n = fn(*args, buf, cap)
            if n <= cap:
                return [int(buf[i]) for i in range(n)]
            cap = n

    def query(self, zoom, x, y, neighbours=False):
        return tuple(self._list(self.L.gd_query, self.h, 1 if neighbours else 0, k, zoom, x, y) for k in range(3))

    def way_nodes(self, i):
        return self._list(self.L.gd_way_nodes, self.h, i)

    def multipolygon_polygons(self, i):
        return self._list(self.L.gd_multipolygon_polygons, self.h, i)

    def polygon_nodes(self, i):
        return self._list(self.L.gd_polygon_nodes, self.h, i)

    def way_is_closed(self, i):
        return bool(self.L.gd_way_is_closed(self.h, i))

    def global_id(self, kind, i):
        return int(self.L.gd_global_id(self.h, kind, i))

    def node_table(self):
        out = np.empty((self.n_nodes, 2), dtype=np.float64)
        self.L.gd_node_table(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def tag(self, kind, i, key):
        buf = C.create_string_buffer(4096)
        n = self.L.gd_tag(self.h, kind, i, key.encode(), buf, 4096)
        return None if n < 0 else buf.raw[:n].decode("utf-8")

    @staticmethod
    def tile_range(zoom, x, y):
        L = C.CDLL(build_shim())
        L.gd_tile_range.argtypes = [C.c_uint8, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        out = (C.c_uint32 * 4)()
        L.gd_tile_range(zoom, x, y, out)
        return tuple(int(v) for v in out)
