"""the host-side reader of the reference's geodata file (osm_renderer_amd/host/osmt_geodata.hpp).
Pinned by the reference's own known answers: the doctests of tile_to_max_zoom_tile_range (src/tile.rs:41-63) and the
data + expected result of saver.rs's unit test test_synthetic_data (src/geodata/saver.rs:234-311), written here with
a restatement of the saver (tests/_geodata.py) because the Rust importer cannot run."""
import numpy as np
import pytest

from osm_renderer_amd import abi
from osm_renderer_amd.display_list import JOB_DTYPE, OP_DTYPE, RING_DTYPE, DisplayList
from tests._geodata import Reader, get_tile_references, write_geodata


def test_tile_range_doctests():
    assert Reader.tile_range(0, 0, 0) == (0, 262143, 0, 262143)
    assert Reader.tile_range(15, 19805, 10244) == (158440, 158447, 81952, 81959)
    assert Reader.tile_range(18, 239662, 158582) == (239662, 239662, 158582, 158582)


def test_saver_unit_test_known_answer(tmp_path):
    """saver.rs:234-311: 20 nodes, one per z18 tile; tile {zoom 15, x 0, y 1} covers z18 x 0..7, y 8..15"""
    tiles = [(1, 7, False), (1, 8, True), (1, 9, True), (1, 13, True), (2, 10, True), (2, 11, True), (2, 15, True), (2, 16, False),
             (2, 17, False), (4, 1, False), (4, 4, False), (5, 20, False), (5, 23, False), (5, 200, False), (7, 6, False),
             (7, 11, True), (7, 12, True), (7, 14, True), (7, 16, False), (7, 17, False)]
    good = [i for i, t in enumerate(tiles) if t[2]]
    nodes = [(i, 1.0, 1.0, {}) for i in range(len(tiles))]
    refs = {(x, y): ([i], [], []) for i, (x, y, _) in enumerate(tiles)}
    p = str(tmp_path / "synthetic.bin")
    write_geodata(p, nodes, tile_refs=refs)
    r = Reader(p)
    assert (r.n_nodes, r.n_ways, r.n_polygons, r.n_multipolygons, r.n_tiles) == (20, 0, 0, 0, 20)
    assert r.query(15, 0, 1)[0] == good
    # neighbouring tiles add (1,7), (7,6) [row above], (2,16), (2,17), (7,16), (7,17) [row below], ...: sorted, unique
    n = r.query(15, 0, 1, neighbours=True)[0]
    assert n == sorted(set(n)) and set(good) <= set(n) and 0 in n and 7 in n
    assert r.query(15, 3, 3)[0] == []
    r.close()


def _world(oracle, rng, n_ways=60):
    """a few hundred nodes around (55.75, 37.61) with ways, closed ways and multipolygons, tagged"""
    lat0, lon0 = 55.75, 37.61
    nodes, ways, polygons, multis = [], [], [], []
    def node(lat, lon, tags=None):
        nodes.append((1000 + len(nodes), lat, lon, tags or {}))
        return len(nodes) - 1
    for w in range(n_ways):
        c = (lat0 + rng.uniform(-0.01, 0.01), lon0 + rng.uniform(-0.02, 0.02))
        k = int(rng.integers(2, 9))
        ids = [node(c[0] + rng.uniform(-0.002, 0.002), c[1] + rng.uniform(-0.003, 0.003)) for _ in range(k)]
        closed = rng.random() < 0.5 and k > 2
        if closed:
            ids.append(ids[0])
        tags = {"building": "yes", "name": f"дом {w}"} if closed else {"highway": rng.choice(["service", "residential"]), "ref": str(w)}
        ways.append((5000 + w, ids, tags))
    for m in range(8):
        c = (lat0 + rng.uniform(-0.01, 0.01), lon0 + rng.uniform(-0.02, 0.02))
        pids = []
        for ring in range(int(rng.integers(1, 3))):
            r = 0.002 / (ring + 1)
            ids = [node(c[0] + r * np.cos(a), c[1] + 1.5 * r * np.sin(a)) for a in np.linspace(0, 2 * np.pi, 7)[:-1]]
            ids.append(ids[0])
            polygons.append(ids)
            pids.append(len(polygons) - 1)
        multis.append((9000 + m, pids, {"type": "multipolygon", "landuse": "grass"}))
    multis.append((9999, [], {"type": "multipolygon"}))  # no polygons: dropped by the neighbours query
    node(lat0, lon0, {"amenity": "cafe", "name": "Даблби"})
    return nodes, ways, polygons, multis


def test_reader_against_a_brute_force_model(tmp_path, oracle):
    rng = np.random.default_rng(12)
    nodes, ways, polygons, multis = _world(oracle, rng)
    mzt = lambda lat, lon: oracle.coords_to_max_zoom_tile(lat, lon)
    p = str(tmp_path / "world.bin")
    refs = write_geodata(p, nodes, ways, polygons, multis, max_zoom_tile=mzt)
    r = Reader(p)
    assert (r.n_nodes, r.n_ways, r.n_polygons, r.n_multipolygons, r.n_tiles) == (len(nodes), len(ways), len(polygons), len(multis), len(refs))
    assert np.array_equal(r.node_table(), np.array([[n[1], n[2]] for n in nodes]))
    for i in (0, 7, len(ways) - 1):
        assert r.way_nodes(i) == ways[i][1] and r.global_id(1, i) == ways[i][0]
        assert r.way_is_closed(i) == (len(ways[i][1]) > 2 and ways[i][1][0] == ways[i][1][-1])
        for k, v in ways[i][2].items():
            assert r.tag(1, i, k) == v
        assert r.tag(1, i, "zzz") is None and r.tag(1, i, "a") is None
    assert r.tag(0, len(nodes) - 1, "name") == "Даблби" and r.tag(0, 0, "name") is None
    assert r.multipolygon_polygons(0) == multis[0][1] and r.polygon_nodes(multis[0][1][0]) == polygons[multis[0][1][0]]
    # tile queries at several zooms vs a brute-force scan of the tile index
    cx, cy = mzt(55.75, 37.61)
    for zoom in (18, 17, 15, 13):
        f = 1 << (18 - zoom)
        for (tx, ty) in {(cx // f, cy // f), (cx // f + 1, cy // f), (cx // f - 1, cy // f - 1)}:
            want = [[], [], []]
            for (x, y) in sorted(refs):  # file order: x, then y
                if tx * f <= x < (tx + 1) * f and ty * f <= y < (ty + 1) * f:
                    for k in range(3):
                        want[k].extend(sorted(refs[(x, y)][k]))
            got = r.query(zoom, tx, ty)
            assert list(got) == want, (zoom, tx, ty)
            wn = [set(), set(), set()]
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for (x, y) in refs:
                        if (tx + dx) * f <= x < (tx + dx + 1) * f and (ty + dy) * f <= y < (ty + dy + 1) * f:
                            for k in range(3):
                                wn[k] |= set(refs[(x, y)][k])
            wn[2] = {m for m in wn[2] if multis[m][1]}
            assert [set(v) for v in r.query(zoom, tx, ty, neighbours=True)] == wn
    r.close()


def _display_list(r, oracle, tile, multis_n, ways_n):
    """a toy styler: closed ways and multipolygons become fills, open ways strokes; rings reference NODE INDICES"""
    zoom, tx, ty = tile
    _, way_ids, mp_ids = r.query(zoom, tx, ty, neighbours=True)
    ops, rings, refs = [], [], []
    def ring(ids):
        rings.append((len(refs), len(ids)))
        refs.extend(ids)
    for m in mp_ids:
        op = np.zeros((), OP_DTYPE)
        op["kind"], op["color"], op["opacity"] = abi.OP_FILL_COLOR, (0xAE, 0xD1, 0xA0), 1.0
        op["ring_off"], op["n_rings"] = len(rings), len(r.multipolygon_polygons(m))
        for p in r.multipolygon_polygons(m):
            ring(r.polygon_nodes(p))
        ops.append(op)
    for w in way_ids:
        op = np.zeros((), OP_DTYPE)
        op["ring_off"], op["n_rings"] = len(rings), 1
        if r.way_is_closed(w):
            op["kind"], op["color"], op["opacity"] = abi.OP_FILL_COLOR, (0xBC, 0xA9, 0xA9), 0.9
        else:
            op["kind"], op["color"], op["opacity"], op["width"], op["cap"] = abi.OP_STROKE, (255, 255, 255), 1.0, 4.0, abi.CAP_ROUND
        ring(r.way_nodes(w))
        ops.append(op)
    job = np.zeros(1, JOB_DTYPE)
    job["x"], job["y"], job["zoom"], job["has_canvas"], job["canvas_rgb"] = tx, ty, zoom, 1, (0xF1, 0xEE, 0xE8)
    job["n_ops"], job["n_pts"] = len(ops), len(refs)
    return DisplayList(job, np.array(ops, dtype=OP_DTYPE), np.array(rings, dtype=RING_DTYPE), np.array(refs, dtype=np.uint32),
                       np.zeros(0), abi.COORD_NODE_REF, 1, nodes=r.node_table())


def test_file_to_display_list_in_the_oracle(tmp_path, oracle):
    """the whole feed on the CPU: file -> reader -> NODE_REF display list == the same list with per-point lat/lon"""
    rng = np.random.default_rng(5)
    nodes, ways, polygons, multis = _world(oracle, rng, n_ways=40)
    p = str(tmp_path / "w.bin")
    write_geodata(p, nodes, ways, polygons, multis, max_zoom_tile=lambda a, b: oracle.coords_to_max_zoom_tile(a, b))
    r = Reader(p)
    cx, cy = oracle.coords_to_max_zoom_tile(55.75, 37.61)
    dl = _display_list(r, oracle, (15, cx // 8, cy // 8), multis, ways)
    assert dl.jobs[0]["n_ops"] > 10
    flat = DisplayList(dl.jobs, dl.ops, dl.rings, dl.nodes[dl.coords], dl.dashes, abi.COORD_LATLON_F64, 1)
    a, b = oracle.render_job(dl, 0), oracle.render_job(flat, 0)
    assert np.array_equal(a, b) and len(np.unique(a.reshape(-1, 4), axis=0)) > 10
    r.close()


@pytest.mark.gpu
def test_file_to_gpu_tiles(tmp_path, gpu_ctx, oracle):
    from osm_renderer_amd.display_list import concat

    rng = np.random.default_rng(6)
    nodes, ways, polygons, multis = _world(oracle, rng, n_ways=80)
    p = str(tmp_path / "w.bin")
    write_geodata(p, nodes, ways, polygons, multis, max_zoom_tile=lambda a, b: oracle.coords_to_max_zoom_tile(a, b))
    r = Reader(p)
    cx, cy = oracle.coords_to_max_zoom_tile(55.75, 37.61)
    table = None
    lists = []
    for zoom, dx in ((15, 0), (15, 1), (16, 0), (14, 0)):
        f = 1 << (18 - zoom)
        dl = _display_list(r, oracle, (zoom, cx // f + dx, cy // f), multis, ways)
        if table is None:
            table = dl.nodes
        dl.nodes = table  # ONE node table for every tile of the file
        lists.append(dl)
    batch = concat(lists)
    got = gpu_ctx.render_batch_host(batch)
    assert np.array_equal(got, oracle.render_batch(batch, threads=4))
    r.close()
