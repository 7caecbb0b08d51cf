"""The C-ABI library loads and exports every symbol include/osmtile.h declares; the ctypes /
numpy mirrors have the header's layout.  No compute calls (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from osm_renderer_amd import abi, display_list, lib
from tests import _shim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "osmtile.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(osmt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_lists():
    assert _declared_symbols() == sorted(lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(lib.LIB_PATH):
        pytest.fail("libosmtile.so is not built: run __graft_entry__.build()")
    L = lib.load()
    for name in _declared_symbols():
        assert hasattr(L, name), name
    assert L.osmt_version() >> 16 == 1


def test_struct_layouts_match_the_header():
    s = _shim.lib().shim_sizeof
    assert s(0) == C.sizeof(abi.Op) == display_list.OP_DTYPE.itemsize == 64
    assert s(1) == C.sizeof(abi.Ring) == display_list.RING_DTYPE.itemsize == 8
    assert s(2) == C.sizeof(abi.TileJob) == display_list.JOB_DTYPE.itemsize == 32
    assert s(3) == C.sizeof(abi.Batch)
    assert s(4) == C.sizeof(abi.Config)
    assert s(10) == abi.Op.opacity.offset == display_list.OP_DTYPE.fields["opacity"][1]
    assert s(11) == abi.Op.width.offset == display_list.OP_DTYPE.fields["width"][1]
    assert s(12) == abi.Op.n_dashes.offset == display_list.OP_DTYPE.fields["n_dashes"][1]
    assert s(13) == abi.Op.image_id.offset == display_list.OP_DTYPE.fields["image_id"][1]
    assert s(20) == abi.TileJob.n_ops.offset == display_list.JOB_DTYPE.fields["n_ops"][1]
    assert s(21) == abi.TileJob.pt_off.offset == display_list.JOB_DTYPE.fields["pt_off"][1]
    assert s(30) == abi.Batch.coord_kind.offset
    assert s(31) == abi.Batch.latlon.offset
    assert s(32) == abi.Batch.dashes.offset
    assert s(33) == abi.Batch.nodes.offset
    assert s(34) == abi.Batch.node_refs.offset
    from osm_renderer_amd import labels

    assert s(5) == C.sizeof(abi.Label) == labels.LABEL_DTYPE.itemsize == 40
    assert s(6) == C.sizeof(abi.LabelBatch)
    assert s(40) == abi.Label.image_id.offset == labels.LABEL_DTYPE.fields["image_id"][1]
    assert s(41) == abi.Label.n_segs.offset == labels.LABEL_DTYPE.fields["n_segs"][1]
    assert s(42) == abi.Label.icon_center_x.offset == labels.LABEL_DTYPE.fields["icon_center_x"][1]
    assert s(43) == abi.LabelBatch.job_label_off.offset
    assert s(44) == abi.LabelBatch.n_segs.offset


def test_no_device_is_a_loud_error_not_a_fallback():
    """Without a GPU osmt_create must fail with OSMT_NO_DEVICE (there is no CPU path)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = lib.load()
    h = C.c_void_p()
    rc = L.osmt_create(None, C.byref(h))
    assert rc == abi.NO_DEVICE and not h.value
    assert b"no CPU path" in L.osmt_last_error()


def test_product_does_not_reference_the_oracle():
    """Nothing under osm_renderer_amd/ may import, link or call oracle/."""
    pkg = os.path.join(ROOT, "osm_renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_py" not in text and "liboracle" not in text and "osm_oracle" not in text, f


def _validate(dl):
    L = lib.load()
    b = dl.as_batch()
    rc = L.osmt_validate_batch(C.byref(b))
    return rc, L.osmt_last_error().decode()


def test_validate_batch_rejects_ops_no_job_covers():
    """k_opinfo pre-processes EVERY op of the pool, so an op outside every job's range (or inside two) must not reach
    the device: orphan ops, n_jobs == 0 with ops, overlapping op / point ranges, bad fields of an orphan op."""
    import numpy as np

    from osm_renderer_amd import synth

    good = synth.config2(3)
    assert _validate(good)[0] == abi.OK
    # an orphan op between two jobs' ranges
    dl = synth.config2(3)
    dl.jobs["n_ops"][1] -= 1
    rc, msg = _validate(dl)
    assert rc == abi.INVALID_ARG and "not covered by any job" in msg
    # no jobs at all, ops present
    dl = synth.config2(2)
    dl.jobs = dl.jobs[:0]
    rc, msg = _validate(dl)
    assert rc == abi.INVALID_ARG and "not covered by any job" in msg
    # overlapping op ranges
    dl = synth.config2(3)
    dl.jobs["op_off"][2] -= 1
    rc, msg = _validate(dl)
    assert rc == abi.INVALID_ARG and "overlap" in msg
    # overlapping point ranges (a shared point would be projected against the wrong tile)
    dl = synth.config2(3)
    dl.jobs["pt_off"][2] -= 4
    rc, msg = _validate(dl)
    assert rc == abi.INVALID_ARG and "point ranges" in msg
    # an orphan op with a wild ring range is reported as orphan, a covered one as a ring error
    dl = synth.config2(2)
    dl.ops["ring_off"][5] = 10**7
    rc, msg = _validate(dl)
    assert rc == abi.INVALID_ARG and "ring range" in msg
    # coordinates outside the Web-Mercator square / not finite
    for bad in (np.nan, np.inf, 89.9, -90.0):
        dl = synth.config2(1)
        dl.coords[7, 0] = bad
        rc, msg = _validate(dl)
        assert rc == abi.UNSUPPORTED and "Web-Mercator" in msg, (bad, msg)
    dl = synth.config2(1)
    dl.coords[7, 1] = 181.0
    assert _validate(dl)[0] == abi.UNSUPPORTED
    # empty batches stay valid
    dl = synth.config2(1)
    dl.jobs = dl.jobs[:0]
    dl.ops = dl.ops[:0]
    assert _validate(dl)[0] == abi.OK


def test_validate_batch_partition_is_order_free_and_exact():
    """The partition check sorts the jobs' ranges (O(n_jobs log n_jobs), osmt_render_batch_multi runs it on the calling
    thread while every GPU's thread validates its own jobs): jobs may be listed in any order and empty jobs may sit
    anywhere, but a gap at the end of the op pool, or a node reference beyond the table, is still refused."""
    import numpy as np

    from osm_renderer_amd import synth

    dl = synth.config2(5)
    dl.jobs = dl.jobs[[3, 0, 4, 1, 2]].copy()  # any order
    assert _validate(dl)[0] == abi.OK
    dl = synth.config2(4)
    empty = dl.jobs[:1].copy()
    empty["n_ops"] = 0
    empty["n_pts"] = 0
    empty["op_off"] = dl.jobs["op_off"][2]  # an empty job pointing into the middle of the pool
    dl.jobs = np.concatenate([dl.jobs[:2], empty, dl.jobs[2:]])
    assert _validate(dl)[0] == abi.OK
    dl = synth.config2(3)
    dl.jobs["n_ops"][2] -= 2  # the last two ops of the pool belong to nobody
    rc, msg = _validate(dl)
    assert rc == abi.INVALID_ARG and "not covered by any job" in msg
    dl = synth.config2(2).with_node_refs()
    assert _validate(dl)[0] == abi.OK
    dl.coords[11] = len(dl.nodes)  # one past the node table
    rc, msg = _validate(dl)
    assert rc == abi.INVALID_ARG and "node reference" in msg
    dl = synth.config2(2).with_node_refs()
    dl.nodes[3, 0] = 91.0  # a node outside the Web-Mercator square
    rc, msg = _validate(dl)
    assert rc == abi.UNSUPPORTED and "Web-Mercator" in msg
