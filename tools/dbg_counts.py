"""Diagnostic: per-wave work counters of k_raster from a library built with -DOSMT_ABL=5 (the counters replace the
first 8 pixels of every sub-tile's first row).  OSMT_LIB=.../libosmtile_dbg.so python tools/dbg_counts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import Context

ctx = Context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "config2"
if which == "filter_test":  # the tile of tests/test_gpu_parity_ops.py::test_filter_groups_cut_by_slots_and_by_kept_records
    import types
    import tests.test_gpu_parity_ops as T
    grabbed = []
    class _Grab(Exception):
        pass
    def _upload(dl):
        grabbed.append(dl)
        raise _Grab()
    try:
        T.test_filter_groups_cut_by_slots_and_by_kept_records(types.SimpleNamespace(upload=_upload), None)
    except _Grab:
        pass
    dl = grabbed[0]
else:
    dl = synth.config5(16) if which == "config5" else synth.config2(64)
NT = dl.n_jobs
out = ctx.render(ctx.upload(dl)).cpu().numpy().view(np.uint32).reshape(NT, 256, 256)
c = out[:, ::16, :].reshape(NT, 16, 8, 32)[:, :, :, :8].reshape(-1, 8).astype(np.int64)
names = ["stroke_visits", "passes", "items", "group_filter_passes", "groups_cut_by_kept", "fill_visits", "ops_over_segcap_kept", "ops_over_filtcap_slots"]
print("waves", len(c))
for i, n in enumerate(names):
    print(f"{n:16s} per wave {c[:, i].mean():10.2f}   per tile {c[:, i].sum() / NT:12.1f}")
print("items per pass", c[:, 2].sum() / max(c[:, 1].sum(), 1), " passes per stroke visit", c[:, 1].sum() / max(c[:, 0].sum(), 1))
