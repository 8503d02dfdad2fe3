"""The launch rules of the dense 3x3 conv, pinned without a GPU: nbdt_conv_plan is host arithmetic over the descriptor and
the process's reserved CUs (csrc/conv_halo.hip: conv_halo_applicable, conv_ksplit_rule).  The numbers behind every rule:
profiles/r05_half_tile_ab.txt."""
import pytest

import nbdt_path

nbdt_path.add()
from nbdt import _C, ops  # noqa: E402

DMA, FOUR_WAVE, PP512, PP512_PAD, HALF = 0, 1, 2, 3, 4
TILED = 0x1000          # any non-zero "address": the plan never dereferences it


def fwd(B, H, C, tiled=True, cout=None):
    d = ops.conv_fwd_desc(B, H, H, C, cout or C, 3, 1)
    d.w_tiled = TILED if tiled else 0
    return d


def dgrad(B, H, C):
    (d,) = ops.conv_dgrad_descs(B, H, H, C, C, 3, 1)
    d.w_tiled = TILED
    return d


def test_headline_grids_take_512_pixel_tiles():
    # WRN-28-10 at 512 images per GPU: every dense 3x3 launch has >= 192 tiles of 512 pixels
    for H, C in ((32, 160), (16, 320), (8, 640)):
        assert ops.conv_plan(fwd(512, H, C)) == (PP512, 1)
        assert ops.conv_plan(dgrad(512, H, C)) == (PP512, 1)


def test_small_grids_take_half_tiles_when_they_fit_the_cus_in_one_round():
    assert ops.conv_plan(fwd(256, 8, 640)) == (HALF, 1)          # config 3's shard: 256 half tiles, one per CU
    assert ops.conv_plan(fwd(128, 16, 256)) == (HALF, 1)         # ResNet18 / 64x64, stage 3
    assert ops.conv_plan(fwd(320, 8, 640)) == (PP512, 1)         # 320 half tiles = two rounds: one round of 160 full tiles
    assert ops.conv_plan(fwd(160, 16, 320)) == (PP512, 1)
    assert ops.conv_plan(fwd(256, 8, 640, tiled=False)) == (FOUR_WAVE, 1)     # no DMA-ordered weights: 4-wave kernels


def test_split_k_needs_a_long_k_loop_and_free_cus():
    assert ops.conv_plan(fwd(128, 4, 512)) == (HALF, 4)          # ResNet18 / CIFAR stage 4: 32 tiles x 144 K steps
    assert ops.conv_plan(dgrad(128, 4, 512)) == (HALF, 4)        # 32 x 4 = 128 blocks: half the CUs, fine beside a weight gradient
    assert ops.conv_plan(fwd(128, 8, 512)) == (HALF, 2)          # 128 tiles: two blocks per tile fill the chip
    assert ops.conv_plan(dgrad(128, 8, 512)) == (HALF, 1)        # ... which a data gradient may not (weight gradient beside it)
    assert ops.conv_plan(fwd(128, 8, 256)) == (HALF, 1)          # 72 K steps: the exchange costs more than it saves
    assert ops.conv_plan(fwd(64, 8, 640)) == (HALF, 4)           # 64 tiles x 180 steps


def test_forced_forms_and_their_errors():
    d = fwd(256, 8, 640)
    d.wide_tile = 3
    assert ops.conv_plan(d) == (FOUR_WAVE, 1)
    d.wide_tile = 2
    assert ops.conv_plan(d) == (PP512, 1)
    d.wide_tile = 4
    assert ops.conv_plan(d) == (PP512_PAD, 1)
    d.wide_tile = 5
    assert ops.conv_plan(d) == (HALF, 1)                         # a forced form never splits on its own
    d.ksplit = 3
    assert ops.conv_plan(d) == (HALF, 3)
    d.ksplit = 64
    assert ops.conv_plan(d) == (HALF, 20)                        # at most one 32-channel slice per block
    e = fwd(512, 8, 640)
    e.ksplit = 1
    assert ops.conv_plan(e) == (PP512, 1)
    wide = fwd(4, 32, 160)
    wide.wide_tile = 4                                           # a 32-wide image has no padded form
    with pytest.raises(_C.NBDTHipError, match="wide_tile"):
        ops.conv_plan(wide)
    strided = ops.conv_fwd_desc(64, 32, 32, 160, 320, 3, 2)
    strided.w_tiled = TILED
    assert ops.conv_plan(strided) == (DMA, 1)
    strided.wide_tile = 5
    with pytest.raises(_C.NBDTHipError, match="wide_tile"):
        ops.conv_plan(strided)
    assert ops.conv_plan(ops.conv_fwd_desc(64, 16, 16, 160, 320, 1, 1)) == (DMA, 1)


def test_reserved_cus_move_the_rules():
    # data-parallel training keeps CUs free for RCCL (nbdt_set_reserved_cus): 256 half tiles no longer fit one round
    try:
        ops.set_reserved_cus(16)
        assert ops.conv_plan(fwd(256, 8, 640)) == (PP512, 1)
        assert ops.conv_plan(fwd(128, 8, 512)) == (HALF, 1)      # 128 tiles x 2 > 240 CUs
        assert ops.conv_plan(fwd(128, 4, 512)) == (HALF, 4)
    finally:
        ops.set_reserved_cus(0)
    assert ops.conv_plan(fwd(256, 8, 640)) == (HALF, 1)
