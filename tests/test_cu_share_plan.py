"""Host-only checks of two round-6 rules: the CU split between a confined BatchNorm backward and the weight gradient beside it
(nbdt.ops.plan_cu_share + WRNEngine's per-stage, batch-scaled time budgets), and the XCD-contiguous block order of the depthwise
kernels (csrc/effnet.hip: xcd_contiguous), restated here in Python.  Numbers: profiles/r06_stage_target_ab.txt,
profiles/r06_c5_traffic_by_kernel_{before,after}.txt."""
import pytest

import nbdt_path

nbdt_path.add()
from nbdt import ops  # noqa: E402

# (output-grid pixels, channels) of WRN-28-10's three stages and the engine's budgets for them (us at 512 images)
STAGES = {1024: 160, 256: 320, 64: 640}
BUDGET_US = {1024: 190.0, 256: 170.0, 64: 190.0}     # WRNEngine.share_stage_us + set_cu_share's split_target_us default
BN2_TENSORS, BN1_TENSORS = 5, 6                      # _Engine.share_bn2_tensors / share_bn1_tensors


def _plan(B, grid, C, tensors, us):
    side = int(grid ** 0.5)
    desc = ops.conv_wgrad_desc(B, side, side, C, C, 3, 1) if hasattr(ops, "conv_wgrad_desc") else None
    if desc is None:
        pytest.skip("no host-side weight-gradient descriptor builder")
    return ops.plan_cu_share(desc, B * grid * C, tensors, 47.0, us, 16, 128)


def test_engine_defaults_are_the_ones_this_file_pins():
    import inspect
    from nbdt import engine as E
    src = inspect.getsource(E.WRNEngine.__init__)
    assert "self.share_stage_us = {1024: 190.0, 256: 170.0}" in src
    assert "split_target_us=190.0" in inspect.getsource(E._Engine.set_cu_share)


@pytest.mark.parametrize("B", [512, 256, 128])
def test_the_cu_split_does_not_depend_on_the_batch(B):
    """The budgets are quoted at 512 images and scaled by batch / 512 (WRNEngine._split_us): a shard of any size gets the CU
    split of the 512-image step -- with unscaled microseconds a 256-image shard gave its passes half the CUs (10.6 vs 9.3 ms)."""
    for grid, C in STAGES.items():
        for tensors in (BN2_TENSORS, BN1_TENSORS):
            ref_budget, ref_n = _plan(512, grid, C, tensors, BUDGET_US[grid])
            budget, n = _plan(B, grid, C, tensors, BUDGET_US[grid] * B / 512.0)
            assert n % 8 == 0 and 16 <= n <= 128
            assert abs(n - ref_n) <= 8, (B, grid, tensors, n, ref_n)          # the XCD rule quantises to 8 CUs
            side = int(grid ** 0.5)
            blocks = ops.conv_wgrad_blocks(ops.conv_wgrad_desc(B, side, side, C, C, 3, 1), budget)
            assert 0 < blocks <= budget and blocks + n <= 256              # the weight gradient's blocks and the pass share the chip
            assert -(-blocks // 8) + n // 8 <= 32                           # ... XCD by XCD (blocks go to the XCDs round-robin)


def test_headline_split_of_the_benched_configuration():
    # what tests/test_engine_gpu.py sees the engine launch at 512 images: 96 / 112 CUs beside the stage-1 weight gradients,
    # 56 / 72 in stage 2, 96 in stage 3 (80 tiles x 2 pixel splits = 160 weight-gradient blocks)
    assert _plan(512, 1024, 160, BN2_TENSORS, 190.0)[1] == 96
    assert _plan(512, 1024, 160, BN1_TENSORS, 190.0)[1] == 112
    assert {_plan(512, 256, 320, BN2_TENSORS, 170.0)[1], _plan(512, 256, 320, BN1_TENSORS, 170.0)[1]} <= {56, 72}
    assert _plan(512, 64, 640, BN2_TENSORS, 190.0)[1] == 96


def xcd_contiguous(lin, total):
    """csrc/effnet.hip, restated: XCD x = lin % 8 owns the logical range [x*q + min(x, r), ...) of length q + (x < r)."""
    q, r, x, k = total >> 3, total & 7, lin & 7, lin >> 3
    return x * q + min(x, r) + k


@pytest.mark.parametrize("total", list(range(1, 70)) + [127, 128, 129, 1000, 1280, 1281, 4096, 12345])
def test_xcd_contiguous_is_a_bijection_with_contiguous_ranges_per_xcd(total):
    image = [xcd_contiguous(lin, total) for lin in range(total)]
    assert sorted(image) == list(range(total))
    for x in range(min(8, total)):
        mine = [image[lin] for lin in range(x, total, 8)]
        assert mine == list(range(mine[0], mine[0] + len(mine)))           # one contiguous, ascending logical range per XCD


@pytest.mark.parametrize("K", [3, 5])
def test_depthwise_weight_gradient_decode_covers_every_kernel_row_of_every_block(K):
    gx, gy = 7, 5                                      # (row blocks x channel blocks, batch chunks)
    total = gx * gy * K
    seen = set()
    for lin in range(total):
        lb = xcd_contiguous(lin, total)
        r, bx, by = lb % K, (lb // K) % gx, lb // (K * gx)
        seen.add((r, bx, by))
    assert len(seen) == total and all(r < K and bx < gx and by < gy for r, bx, by in seen)
