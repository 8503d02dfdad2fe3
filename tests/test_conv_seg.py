"""Host-only checks of the slice-list convolution plans (csrc/conv_seg.hip, nbdt_conv_seg_*): the tile / buffer choice
and the LDS-DMA schedule of every K step, re-derived here from the ordering rules in the kernel's header -- no GPU.

Reference ops these launches replace: nbdt/models/resnet.py:56-67 (strided conv1 + 1x1 shortcut), pytorchcv PreResUnit
with stride 2 behind nbdt/models/wideresnet.py:1-5."""
import numpy as np
import pytest

import nbdt_path  # noqa: F401
from nbdt import ops


def halo_pieces(gh, gw, tile):
    if gw * gh >= tile:
        rb, ib = tile // gw, 1
    else:
        rb, ib = gh, tile // (gw * gh)
    hp = ib * (rb + 2) * (gw + 2)
    return (hp * 4 + 63) // 64


def check_schedule(plan, gh, gw):
    """Every slice's pieces go out exactly once, into the buffer the slice's steps read, not before the buffer's previous
    tenant was read for the last time and early enough to be readable (two steps ahead if strict, three otherwise)."""
    a_instr = halo_pieces(gh, gw, plan.tile)
    a_bytes = a_instr * 1024
    rounds = (a_instr + 7) // 8
    hw2 = gw + 2
    worst = 0
    for c, k in enumerate(plan.classes):
        rec, npro = plan.steps(c)
        slices = k["slices"]
        first, t = [], 0
        for s in slices:
            first.append(t)
            t += len(s[2])
        assert t == len(rec) == plan.nsteps[c]
        last = [f + len(s[2]) - 1 for f, s in zip(first, slices)]
        # what each step reads
        for si, s in enumerate(slices):
            for j, tap in enumerate(s[2]):
                r = rec[first[si] + j]
                assert r[0] == (tap // 3) * hw2 + tap % 3
                assert r[1] == (si % plan.nbuf) * a_bytes
        # what each step issues
        got = {si: [] for si in range(len(slices))}
        for tt, r in enumerate(rec):
            n = int(r[2])
            if n == 0:
                continue
            worst = max(worst, n)
            buf, off = divmod(int(r[3]), a_bytes)
            id0 = off // 1024
            assert off % 1024 == 0 and r[4] == id0 * 16 and id0 % 8 == 0
            # the slice being fetched: the first one after this step's that lives in that buffer with this tensor/channel
            cand = [si for si in range(len(slices)) if si % plan.nbuf == buf and first[si] > tt
                    and slices[si][0] == r[6] and slices[si][1] == r[5]]
            assert cand, (c, tt, r)
            si = cand[0]
            lo = last[si - plan.nbuf] + 1 if si >= plan.nbuf else 0
            hi = first[si] - 2
            assert lo <= tt <= hi, (c, tt, si, lo, hi)
            assert int(r[7]) & 1 == (1 if tt == hi else 0)
            # bits 8..: how many waves issue the step's last round (the slice's last round can be short of waves)
            assert int(r[7]) >> 8 == min(8, a_instr - (id0 + 8 * (n - 1)))
            got[si].extend(range(id0 // 8, id0 // 8 + n))
        for si in range(len(slices)):
            if si < npro:
                assert got[si] == []
            else:
                assert got[si] == list(range(rounds)), (c, si, got[si])
        assert npro in (1, 2)
    assert worst == plan.max_rounds <= 7
    return worst


WRN = [  # (what, builder, pixel grid) at 512 images: the launches of WRN-28-10's three shape-changing units
    ("s2u1 conv1", lambda: ops.seg_fwd_s2(512, 32, 32, 160, 320), (16, 16)),
    ("s3u1 conv1", lambda: ops.seg_fwd_s2(512, 16, 16, 320, 640), (8, 8)),
    ("s1u1 conv2+sc", lambda: ops.seg_conv3x3_plus_1x1(512, 32, 32, 160, 160, 32, 32), (32, 32)),
    ("s2u1 conv2+sc", lambda: ops.seg_conv3x3_plus_1x1(512, 16, 16, 320, 320, 160, 640), (16, 16)),
    ("s3u1 conv2+sc", lambda: ops.seg_conv3x3_plus_1x1(512, 8, 8, 640, 640, 320, 1280), (8, 8)),
    ("s2u1 dgrad", lambda: ops.seg_dgrad_s2(512, 32, 32, 160, 320, shortcut=True), (16, 16)),
    ("s3u1 dgrad", lambda: ops.seg_dgrad_s2(512, 16, 16, 320, 640, shortcut=True), (8, 8)),
    ("s1u1 dgrad", lambda: ops.seg_dgrad3x3_plus_1x1(512, 32, 32, 32, 160), (32, 32)),
]


@pytest.mark.parametrize("what,build,grid", WRN, ids=[w[0] for w in WRN])
def test_wrn_plans_schedule(what, build, grid):
    plan = build()
    check_schedule(plan, *grid)
    # 8x8 grids cannot hold three 50-piece halo buffers of a 512-pixel tile: half tiles; 16x16 and 32x32 take full tiles
    assert plan.tile == (256 if grid == (8, 8) else 512), (what, plan.tile)
    # (one 1-tap slice at the very end of a class does not need a third buffer)
    assert plan.nbuf == (2 if what == "s1u1 conv2+sc" else 3)
    assert plan.w_tile_elems > 0


def test_dense_slices_need_two_buffers_and_one_round_per_step():
    """A plain 3x3 stride-1 conv as a slice list: nine taps per slice, two halo buffers, the schedule of the dense
    kernel (one piece per wave and step)."""
    cin = cout = 160
    sl = [(0, kc * 32, list(range(9)), 0, [t * cin + kc * 32 for t in range(9)]) for kc in range(cin // 32)]
    plan = ops.ConvSeg(512, 32, 32, cout, [cin], [9 * cin], [{"slices": sl, "out": ops._plain_out(32, 32, cout)}])
    assert (plan.tile, plan.nbuf) == (512, 2)
    assert check_schedule(plan, 32, 32) == 1
    assert plan.nsteps == [45]


def test_bad_descriptions_are_refused():
    ok = [(0, 0, [4], 0, [0])]
    with pytest.raises(Exception):       # channels beyond the pixel
        ops.ConvSeg(8, 8, 8, 32, [32], [32], [{"slices": [(0, 8, [4], 0, [0])], "out": ops._plain_out(8, 8, 32)}])
    with pytest.raises(Exception):       # tap outside the 3x3 halo
        ops.ConvSeg(8, 8, 8, 32, [32], [32], [{"slices": [(0, 0, [9], 0, [0])], "out": ops._plain_out(8, 8, 32)}])
    with pytest.raises(Exception):       # a 24-wide image: 256 / 512 pixels are not whole rows
        ops.ConvSeg(8, 24, 24, 32, [32], [32], [{"slices": ok, "out": ops._plain_out(24, 24, 32)}])
    with pytest.raises(Exception):       # two 1-tap slices in a row with two buffers
        ops.ConvSeg(8, 8, 8, 32, [64], [64], [{"slices": [(0, 0, [4], 0, [0]), (0, 32, [4], 0, [32]), (0, 0, [4], 0, [0])],
                                                "out": ops._plain_out(8, 8, 32)}], nbuf=2)
    plan = ops.ConvSeg(8, 8, 8, 32, [32], [32], [{"slices": ok, "out": ops._plain_out(8, 8, 32)}])
    assert plan.nsteps == [1] and plan.tile in (256, 512)
