"""A one-minute run (each) of the randomised differential fuzzer (tools/fuzz_parity.py): adversarial widths, dashes,
caps, directions, huge coordinates, multi-ring fills, scales 1..3 — GPU vs oracle, bit-exact.
(Round 1, longer runs on the GPU box: 4380 tiles of area ops and 6612 tiles with random label passes, 0 mismatches.)"""
import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_short(gpu_ctx, oracle):
    from tools import fuzz_parity

    # one full cycle of the batch sizes (1, 12, 64, 65, 130) whatever the host's speed: the coverage asserted below is driven
    # by the iteration count, the minute only adds to it
    tiles, bad = fuzz_parity.run(budget=60.0, seed=2026, ctx=gpu_ctx, dump=False, min_iters=len(fuzz_parity.BATCH_SIZES))
    assert tiles >= 60 and bad == 0
    st = fuzz_parity.run.last_stats  # both raster instantiations, the list kernel and empty tiles were all in the run
    assert st["folded_tiles"] > 0 and st["listed_tiles"] > 0 and st["empty_tiles"] > 0 and st["batches"].get(65, 0) > 0


def test_fuzz_short_with_labels(gpu_ctx, oracle):
    """the same with a random label pass per tile: adversarial draw_line calls, icons, collisions, wide windows"""
    from tools import fuzz_parity

    tiles, bad = fuzz_parity.run(budget=60.0, seed=77, ctx=gpu_ctx, dump=False, with_labels=True)
    assert tiles >= 60 and bad == 0
