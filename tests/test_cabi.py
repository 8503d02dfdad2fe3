"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/nbdt_hip.h
declares (no compute calls without a GPU); the product path refuses CPU tensors loudly."""
import os
import re

import pytest
import torch

import nbdt_path
from nbdt import _C


def _header_symbols():
    text = open(os.path.join(nbdt_path.ROOT, "include", "nbdt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbdt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_C.libpath()):
        from nbdt import _build
        _build.build()
    declared = _header_symbols()
    assert len(declared) >= 26
    assert sorted(_C.SIGNATURES) == declared, "ctypes table and header disagree"
    exported = set(_C.exported_symbols())
    assert not (set(declared) - exported), f"missing from .so: {sorted(set(declared) - exported)}"
    lib = _C.lib()
    assert lib.nbdt_version() >= 100
    assert isinstance(lib.nbdt_last_error(), bytes)
    assert lib.nbdt_device_count() >= 0


def test_argument_validation_without_gpu():
    lib = _C.lib()
    # null handle / null pointers are rejected before any HIP call
    assert lib.nbdt_soft_forward(None, None, 0, 4, 10, None, None) == -1
    assert b"null tree handle" in lib.nbdt_last_error()
    d = _C.ConvDesc()
    assert lib.nbdt_conv_igemm(d, None, None, None, None, None) == -1


def test_weight_gradient_block_count_query_is_host_only():
    """nbdt_conv_wgrad_blocks (what engine.set_cu_share plans the CU split with) does no device work: the pixel
    split of the 8-wave kernel for the three WRN-28-10 stage shapes at 512 images, with and without a CU budget;
    0 for launches that take another kernel; blocks never exceed a budget of at least one block per tile; and the ctypes mirror of
    nbdt_wgrad_desc has the header's layout (35 int32)."""
    import ctypes
    from nbdt import ops
    assert ctypes.sizeof(_C.WgradDesc) == 35 * 4
    expect = {(512, 32, 32, 160): {0: 255, 208: 205, 202: 200, 185: 185},
              (512, 16, 16, 320): {0: 240, 229: 220, 220: 220, 216: 200},
              (512, 8, 8, 640): {0: 240, 240: 240, 238: 160}}
    for (B, H, W, C), table in expect.items():
        d = ops.conv_wgrad_desc(B, H, W, C, C, 3, 1)
        for budget, blocks in table.items():
            assert ops.conv_wgrad_blocks(d, budget) == blocks, (B, H, W, C, budget)
        for budget in range(96, 257, 7):     # (below one block per (cout, cin) tile -- 80 in stage 3 -- there is no split left)
            assert 0 < ops.conv_wgrad_blocks(d, budget) <= budget
        assert ops.conv_wgrad_blocks(d, 8) == 0                     # not a valid budget
    assert ops.conv_wgrad_blocks(ops.conv_wgrad_desc(16, 8, 8, 640, 640, 3, 1), 0) == 0    # too few pixels: 4-wave kernel
    assert ops.conv_wgrad_blocks(ops.conv_wgrad_desc(512, 32, 32, 160, 320, 3, 2), 208) == 0  # strided: other kernel
    assert ops.conv_wgrad_blocks(ops.conv_wgrad_desc(512, 32, 32, 160, 320, 1, 1), 208) == 0  # 1x1


def test_cu_share_plan_respects_the_xcd_rule():
    """ops.plan_cu_share (engine.set_cu_share's arithmetic): the CUs of the confined BatchNorm pass plus the weight
    gradient's blocks never exceed 32 per XCD, for the shapes and settings bench.py runs (fused sums: 3 / 4 tensor
    passes at 200 us; split form: 5 / 6 at 190 us) and over a sweep."""
    from nbdt import ops
    shapes = [(512, 32, 32, 160), (512, 16, 16, 320), (512, 8, 8, 640), (256, 32, 32, 160), (1024, 16, 16, 320)]
    seen = {}
    for (B, H, W, C) in shapes:
        d = ops.conv_wgrad_desc(B, H, W, C, C, 3, 1)
        for tensors, us, hi in [(3, 200.0, 96), (4, 200.0, 96), (5, 190.0, 128), (6, 190.0, 128)]:
            budget, n = ops.plan_cu_share(d, B * H * W * C, tensors, 47.0, us, 16, hi)
            blocks = ops.conv_wgrad_blocks(d, budget)
            assert 0 < blocks <= budget and n % 8 == 0 and n >= 8
            assert (blocks + 7) // 8 + n // 8 <= 32, (B, H, W, C, tensors, budget, blocks, n)
            seen[(B, C, tensors)] = (budget, blocks, n)
        for us in range(100, 400, 37):
            for tensors in (3, 4, 5, 6):
                budget, n = ops.plan_cu_share(d, B * H * W * C, tensors, 47.0, float(us), 16, 160)
                blocks = ops.conv_wgrad_blocks(d, budget)
                assert (blocks + 7) // 8 + n // 8 <= 32
    # what the benched configuration runs
    assert seen[(512, 160, 3)] == (202, 200, 56) and seen[(512, 160, 4)] == (185, 185, 64)
    assert seen[(512, 160, 5)] == (162, 160, 96) and seen[(512, 160, 6)] == (143, 140, 112)
    assert seen[(512, 320, 5)][1:] == (200, 56) and seen[(512, 640, 5)][1:] == (160, 96)


def test_product_path_refuses_cpu_tensors():
    from nbdt.loss import SoftTreeSupLoss
    from nbdt.model import HardEmbeddedDecisionRules, SoftEmbeddedDecisionRules
    z = torch.randn(4, 10)
    with pytest.raises(_C.NBDTHipError):
        SoftEmbeddedDecisionRules(dataset="CIFAR10", hierarchy="induced")(z)
    with pytest.raises(_C.NBDTHipError):
        HardEmbeddedDecisionRules(dataset="CIFAR10", hierarchy="induced")(z)
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=torch.nn.CrossEntropyLoss(), hierarchy="induced")
    with pytest.raises(_C.NBDTHipError):
        crit(z, torch.zeros(4, dtype=torch.long))
    if not torch.cuda.is_available():
        from nbdt import engine
        with pytest.raises(RuntimeError):
            engine.WRNEngine(device="cpu")


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(nbdt_path.PKG_DIR, "nbdt")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "nbdt_oracle" not in src and "torch_models" not in src, f


def test_timing_experiment_switches_do_not_compile_into_a_product_build(tmp_path):
    """-DNBDT_WPP_NO_EPI, -DNBDT_PP_KFRAC5=3, -DNBDT_PP_ABLATE=4 ... skip or fake part of a kernel (wrong
    gradients by design).  csrc/common.h turns any of them into a compile error unless -DNBDT_TIMING_BUILD is given too,
    and a timing build exports nbdt_timing_build, which nbdt._C.lib() refuses without NBDT_ALLOW_TIMING_BUILD=1."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    src = tmp_path / "probe.hip"
    src.write_text('#include "common.h"\nint main() { return 0; }\n')
    base = [hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-I",
            os.path.join(nbdt_path.PKG_DIR, "csrc"), str(src)]
    for switch in ("-DNBDT_WPP_NO_EPI", "-DNBDT_PP_KFRAC5=3", "-DNBDT_PP_ABLATE=4", "-DNBDT_HEAD_SKIP=1"):
        bad = subprocess.run(base + [switch], capture_output=True, text=True)
        assert bad.returncode != 0 and "NBDT_TIMING_BUILD" in bad.stderr, (switch, bad.stderr[-500:])
        ok = subprocess.run(base + [switch, "-DNBDT_TIMING_BUILD"], capture_output=True, text=True)
        assert ok.returncode == 0, (switch, ok.stderr[-500:])
    assert subprocess.run(base + ["-DNBDT_PP_ABLATE=0"], capture_output=True, text=True).returncode == 0
    # the shipped library is not a timing build
    import ctypes
    assert not hasattr(ctypes.CDLL(_C.libpath()), "nbdt_timing_build")


def test_process_wide_switches_are_host_only():
    """Setters / getters that take no device: callable on a box without a GPU, defaults as the header states."""
    from nbdt import ops
    assert ops.wgrad_store_epilogue() is True            # include/nbdt_hip.h: default 1
    ops.set_wgrad_store_epilogue(False)
    try:
        assert ops.wgrad_store_epilogue() is False
    finally:
        ops.set_wgrad_store_epilogue(True)
    assert ops.is_deterministic() is False and ops.reserved_cus() == 0
    # the descriptor builders carry the A/B default of nbdt_conv_desc.ksplit (0 = the launch rule decides)
    assert ops.conv_fwd_desc(8, 8, 8, 64, 64, 3, 1).ksplit == 0
    (d,) = ops.conv_dgrad_descs(8, 8, 8, 64, 64, 3, 1)
    assert d.ksplit == 0 and d.wide_tile == 0 and ops.conv_fwd_desc(8, 8, 8, 64, 64, 3, 1).wide_tile == 1
