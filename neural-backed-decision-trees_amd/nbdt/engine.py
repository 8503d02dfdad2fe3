"""Backbone execution engine: explicit forward/backward/SGD over hand-written HIP kernels.

This replaces the autograd graph of stock aten/cuDNN ops the reference builds per step
(main.py:233-239 around nbdt/models/resnet.py:136-149 / pytorchcv CIFARWRN.forward) with a fixed,
pre-planned launch sequence:

  * every parameter lives in ONE flat fp32 buffer (``ParamStore``) with a flat gradient buffer, a
    flat momentum buffer and a flat bf16 mirror -- SGD is one kernel, gradient all-reduce is a few
    large buckets, and the bf16 weights the convs read are refreshed by the SGD kernel itself;
  * activations are padded NHWC bf16 buffers allocated once per batch size and kept resident
    (288 GB of HBM: nothing is recomputed or freed inside a step);
  * forward records nothing: backward is the hand-written mirror sequence.

Python here only sequences C-ABI launches on the current HIP stream (capturable in a hipGraph via
torch.cuda.CUDAGraph); there is no CPU or PyTorch-op fallback for any tensor op on the path.
"""
import math

import torch

from nbdt import _C, ops

SIDE_STREAM_PRIORITY = 0       # HIP stream priority of the weight-gradient stream (A/B: scratch/ab_stream_priority.py)

ALIGN = 8  # elements: keeps every parameter 16-byte aligned in the bf16 mirror

_SIDE_STREAMS = {}


def side_stream(device):
    """THE second HIP stream of this process on `device` (weight gradients, derived-weight builds): created once and
    shared by every engine.  Streams created later in a process land on the main stream's hardware queue in about one
    of four tries (measured in round 3; seen again in round 6 as one 20 ms engine in ten in scripts that build several
    engines) -- there the two-stream schedule serialises and a WRN-28-10 step takes 19-20 ms instead of 16.5.  One engine
    steps at a time, so sharing the stream costs nothing."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=SIDE_STREAM_PRIORITY)
    return _SIDE_STREAMS[key]


def _pad32(c):
    return (c + 31) // 32 * 32


class ParamStore:
    """Flat parameter / gradient / momentum / bf16-mirror buffers + named views."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.entries = {}   # name -> (offset, internal_shape)
        self.inits = []     # (name, fn(view))
        self.n = 0
        self.flat = self.grad = self.mom = self.bf16 = None

    def add(self, name, shape, init):
        assert self.flat is None and name not in self.entries
        numel = 1
        for s in shape:
            numel *= s
        self.entries[name] = (self.n, tuple(shape))
        self.inits.append((name, init))
        self.n += (numel + ALIGN - 1) // ALIGN * ALIGN

    def finalize(self):
        n = (self.n + 3) // 4 * 4
        host = torch.zeros(n, dtype=torch.float32)
        for name, init in self.inits:
            off, shape = self.entries[name]
            numel = math.prod(shape)
            init(host[off:off + numel].view(shape))
        self.flat = host.to(self.device)
        self.grad = torch.zeros_like(self.flat)
        self.mom = torch.zeros_like(self.flat)
        self.bf16 = self.flat.to(torch.bfloat16)

    def _view(self, buf, name):
        off, shape = self.entries[name]
        return buf[off:off + math.prod(shape)].view(shape)

    def _cached(self, which, buf, name):
        # the flat buffers are allocated once (finalize) and only ever written in place: a named view is made once.  (ResNet18
        # on CIFAR10 at 128 images is bound by the host's launch path: 9,000 slice + view calls per 50 steps were 10 % of it.)
        cache = self.__dict__.setdefault("_views", {})
        hit = cache.get((which, name))
        if hit is not None and hit[1] is buf:          # (a replaced flat buffer -- nobody does that today -- is a miss)
            return hit[0]
        v = self._view(buf, name)
        cache[(which, name)] = (v, buf)
        return v

    def p(self, name):
        return self._cached(0, self.flat, name)

    def g(self, name):
        return self._cached(1, self.grad, name)

    def pb(self, name):
        return self._cached(2, self.bf16, name)

    def refresh_bf16(self):
        self.bf16.copy_(self.flat)   # plumbing: only after load_state_dict, not on the step path

    def zero_grad(self):
        self.grad.zero_()            # one memset node


class Conv:
    """3x3 / 1x1 convolution (bias-free) with forward, data-gradient and weight-gradient launches."""

    def __init__(self, store, name, cin_real, cout_real, k, stride, gen, init="kaiming_a0"):
        self.store, self.name = store, name
        self.cin_real, self.cout_real, self.k, self.stride = cin_real, cout_real, k, stride
        self.cin, self.cout = _pad32(cin_real), _pad32(cout_real)
        self.taps = k * k
        fan_in = cin_real * k * k
        if init == "kaiming_a0":       # pytorchcv: init.kaiming_uniform_(w)  (a = 0)
            bound = math.sqrt(2.0) * math.sqrt(3.0 / fan_in)
        else:                          # nn.Conv2d default: kaiming_uniform_(w, a=sqrt(5))
            bound = 1.0 / math.sqrt(fan_in)

        def init(v):
            v.zero_()
            v[:cout_real, :, :cin_real].uniform_(-bound, bound, generator=gen)

        store.add(name, (self.cout, self.taps, self.cin), init)
        self._plans = {}
        self.wd = None
        self.merge_parity_classes = True   # (A/B: False launches the parity classes of a strided data gradient one by one)

    def logical(self, buf):
        """[cout_real, cin_real, k, k] view (OIHW semantics, channels_last memory) of a flat buffer."""
        v = self.store._view(buf, self.name).view(self.cout, self.k, self.k, self.cin)
        return v[:self.cout_real, :, :, :self.cin_real].permute(0, 3, 1, 2)

    def plan(self, B, Hi, Wi):
        key = (B, Hi, Wi)
        if key not in self._plans:
            self._plans[key] = self._make_plan(B, Hi, Wi)
        return self._plans[key]

    def _make_plan(self, B, Hi, Wi):
        p = self._raw_plan(B, Hi, Wi)
        if getattr(self, "wt_fwd", None) is not None:      # dense 3x3 / stride 1: DMA-ordered weight tiles
            p[0].w_tiled = self.wt_fwd.data_ptr()
            for descs in (p[1], p[2]):
                if descs is not None and len(descs) == 1:
                    descs[0].w_tiled = self.wt_dgrad.data_ptr()
        return p

    def _raw_plan(self, B, Hi, Wi):
        """(forward desc, dgrad descs, accumulating dgrad descs, wgrad desc) for one input geometry."""
        args = (B, Hi, Wi, self.cin, self.cout, self.k, self.stride)
        plain_dgrad = None if (self.k == 1 and self.stride == 2) else ops.conv_dgrad_descs(*args, accumulate=False)
        plan = (ops.conv_fwd_desc(*args), plain_dgrad, ops.conv_dgrad_descs(*args, accumulate=True),
                ops.conv_wgrad_desc(*args))
        # what the launch timers count as a launch's flops: the layer's REAL channels, not the 32-padded ones the kernels
        # multiply (16 -> 32 for WRN's first unit): fwd / wgrad (cin, cout), dgrad (cout, cin) -- ops.desc_flops
        for d in (plan[0], plan[3]):
            d.flop_channels = (self.cin_real, self.cout_real)
        for descs in (plan[1], plan[2]):
            for d in descs or ():
                d.flop_channels = (self.cout_real, self.cin_real)
        return plan

    def numel(self):
        return self.cout * self.taps * self.cin

    def _w_for(self, x):
        """Forward weights in the storage type of the activations: the bf16 mirror, or (verification-only fp32
        reference mode) the fp32 master itself."""
        return self.store.p(self.name) if x.dtype == torch.float32 else self.store.pb(self.name)

    def _wd_for(self, g):
        """Data-gradient weights [cin][taps][cout], tap order reversed: the bf16 copy weight_prep built, or an fp32
        one (reference mode; plumbing on parameters, rebuilt by refresh_derived_weights)."""
        return self.wd32 if g.dtype == torch.float32 else self.wd

    def forward(self, x, out, residual=None, bn_scratch=None):
        B, Hp, Wp, _ = x.shape
        ops.conv_igemm(self.plan(B, Hp - 2, Wp - 2)[0], x, self._w_for(x), out, residual, bn_scratch)

    def forward_affine(self, x, out, bn, act=1, residual=None):
        """Inference: conv + eval-mode BatchNorm `bn` (folded to scale/shift) + activation [+ residual] in one
        launch (act: 0 none, 1 ReLU, 2 swish)."""
        B, Hp, Wp, _ = x.shape
        scale, shift = bn.eval_affine()
        ops.conv_igemm_affine(self.plan(B, Hp - 2, Wp - 2)[0], x, self.store.pb(self.name), out, scale, shift, act,
                              residual)

    def backward_data(self, gout, gin, accumulate=False, bn=None, bn_x=None, partials=None):
        """bn/bn_x/partials: `gin` is dL/d(relu(bn(bn_x))) -- also emit that BatchNorm's backward sums
        (single-launch stride-1 dgrads only)."""
        B, Hp, Wp, _ = gin.shape
        plan = self.plan(B, Hp - 2, Wp - 2)
        descs = plan[2] if accumulate else plan[1]
        if bn is not None:
            assert len(descs) == 1 and not accumulate
            ops.conv_igemm_bnbwd(descs[0], gout, self.wd, gin, bn_x, bn.mean, bn.rstd, bn.gamma, bn.beta, partials)
            return
        if len(descs) > 1 and self.merge_parity_classes:      # strided 3x3: four parity classes, one grid
            ops.conv_igemm_multi(descs, gout, self._wd_for(gout), gin)
            return
        for d in descs:
            ops.conv_igemm(d, gout, self._wd_for(gout), gin)

    def s2d_plan(self, B, Hi, Wi):
        """(forward desc [1x1 only], wgrad desc) of this stride-2 conv over the space-to-depth copy of its input."""
        key = ("s2d", B, Hi, Wi)
        if key not in self._plans:
            fwd = ops.conv_fwd_desc_s2d_1x1(B, Hi, Wi, self.cin, self.cout) if self.k == 1 else None
            wg = ops.conv_wgrad_desc_s2d(B, Hi, Wi, self.cin, self.cout, self.k)
            for d in (fwd, wg):
                if d is not None:
                    d.flop_channels = (self.cin_real, self.cout_real)
            self._plans[key] = (fwd, wg)
        return self._plans[key]

    def forward_s2d(self, xs, out, Hi, Wi):
        ops.conv_igemm(self.s2d_plan(xs.shape[0], Hi, Wi)[0], xs, self._w_for(xs), out)

    def backward_weight_s2d(self, xs, gout, Hi, Wi, cu_budget=0):
        desc = self.s2d_plan(xs.shape[0], Hi, Wi)[1]
        side = getattr(self, "side_stream", None)
        if side is None:
            ops.conv_wgrad(desc, xs, gout, self.store.g(self.name), cu_budget)
            return
        main = torch.cuda.current_stream(xs.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.conv_wgrad(desc, xs, gout, self.store.g(self.name), cu_budget)

    def backward_weight(self, x, gout, cu_budget=0):
        """cu_budget: CUs this launch is sized for when it runs on the second stream (0 = all of them): the caller
        is about to launch an HBM-bound pass on the main stream that should get the remaining CUs."""
        B, Hp, Wp, _ = x.shape
        side = getattr(self, "side_stream", None)
        if side is None:
            ops.conv_wgrad(self.plan(B, Hp - 2, Wp - 2)[3], x, gout, self.store.g(self.name), cu_budget)
            return
        # the weight gradient only feeds the optimizer, so it runs on a second stream next to the
        # data-gradient chain (see WRNEngine.backward for the buffer-reuse ordering)
        main = torch.cuda.current_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.conv_wgrad(self.plan(B, Hp - 2, Wp - 2)[3], x, gout, self.store.g(self.name), cu_budget)


class SegOp:
    """A slice-list launch (ops.ConvSeg, csrc/conv_seg.hip) together with its DMA-ordered weight tiles.  `sources` returns
    the bf16 weight matrices the tiles are built from (views of the engine's flat bf16 mirror / data-gradient copies) and
    `sources32` their fp32 twins for the verification-only reference mode; `on_side`: the sources are the data-gradient
    copies the engine builds on its second stream, so the tiles are built there too."""

    def __init__(self, plan, sources, sources32, on_side):
        self.plan, self.sources, self.sources32, self.on_side = plan, sources, sources32, on_side
        self.tiles = None

    def retile(self):
        self.tiles = self.plan.tile_weights(self.sources(), out=self.tiles)

    def __call__(self, ins, out, bn_scratch=None):
        if ins[0].dtype == torch.float32:
            self.plan(ins, self.sources32(), out, bn_scratch=bn_scratch)
        else:
            self.plan(ins, self.tiles, out, bn_scratch=bn_scratch)


class BatchNorm:
    def __init__(self, store, name, c_real, scratch_owner):
        self.store, self.name, self.c_real = store, name, c_real
        self.C = _pad32(c_real)
        store.add(name + ".weight", (self.C,), lambda v: v.fill_(1.0))
        store.add(name + ".bias", (self.C,), lambda v: v.zero_())
        dev = store.device
        self.running_mean = torch.zeros(self.C, device=dev)
        self.running_var = torch.ones(self.C, device=dev)
        self.num_batches_tracked = 0   # host counter (state_dict compat); no launch on the step path
        self.mean = torch.empty(self.C, device=dev)
        self.rstd = torch.empty(self.C, device=dev)
        self.dsum = torch.empty(2 * self.C, device=dev)
        self.owner = scratch_owner
        self._affine = None            # cached (scale, shift) of the eval-mode transform

    def eval_affine(self):
        """scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale: two [C] vectors computed
        once per parameter / running-statistics version (plumbing on parameters, not on the data path)."""
        if self._affine is None:
            scale = self.gamma * torch.rsqrt(self.running_var + ops.BN_EPS)
            self._affine = (scale.contiguous(), (self.beta - self.running_mean * scale).contiguous())
        return self._affine

    @property
    def gamma(self):
        return self.store.p(self.name + ".weight")

    @property
    def beta(self):
        return self.store.p(self.name + ".bias")

    def stats(self, x, training, fused=False):
        """fused=True: the producing conv's epilogue already wrote per-tile partial sums (fold only)."""
        if training:
            self._affine = None        # running statistics are about to change
            if fused:
                ops.bn_finalize(x, self.owner.partials(x), self.mean, self.rstd, self.running_mean,
                                self.running_var)
            else:
                ops.bn_stats(x, self.owner.scratch(self.C), self.mean, self.rstd, self.running_mean,
                             self.running_var)
            self.num_batches_tracked += 1
        else:  # eval: running statistics (two [C]-sized plumbing ops, not on the training path)
            self.mean.copy_(self.running_mean)
            torch.rsqrt(self.running_var + ops.BN_EPS, out=self.rstd)

    def apply(self, x, y, relu=True, residual=None):
        ops.bn_apply(x, self.mean, self.rstd, self.gamma, self.beta, y, relu=relu, residual=residual)

    def backward_fused(self, gy, x, gx, partials, gx_add=None, cus=0):
        """Backward of relu(bn(x)) when the dgrad that produced gy already wrote the reduction partials.
        cus > 0: the elementwise pass is confined to that many CUs (a weight gradient runs on the others)."""
        ops.bn_bwd_fused(gy, x, self.mean, self.rstd, self.gamma, self.beta, partials, self.dsum,
                         self.store.g(self.name + ".weight"), self.store.g(self.name + ".bias"), gx, gx_add=gx_add,
                         cus=cus)

    def backward_cus(self, gy, x, gx, cus, gx_add=None):
        """Backward of relu(bn(x)) entirely on `cus` CUs: sums and elementwise pass, the fold of the sums in the latter's
        prologue (no dgrad-epilogue partials; owner.fuse_bn_fold = False: the three-launch form, A/B and tests)."""
        scratch = self.owner.slot_pair(self.C) if self.owner.fuse_bn_fold else self.owner.scratch(self.C)
        ops.bn_bwd_cus(gy, x, self.mean, self.rstd, self.gamma, self.beta, scratch, self.dsum,
                       self.store.g(self.name + ".weight"), self.store.g(self.name + ".bias"), gx, cus, gx_add=gx_add)

    def backward(self, gy, y, x, gx, relu=True, gx_add=None, g_resid=None):
        """y=None: recompute the ReLU mask from x (valid when apply() had no residual)."""
        ops.bn_bwd(gy, y, x, self.mean, self.rstd, self.gamma, self.owner.scratch(self.C), self.dsum,
                   self.store.g(self.name + ".weight"), self.store.g(self.name + ".bias"), gx, relu=relu,
                   gx_add=gx_add, g_resid=g_resid, beta=self.beta)


class _Engine:
    """Shared machinery: parameter store, scratch, activation buffers, optimizer."""

    def __init__(self, device, seed):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the NBDT backbone engine runs on MI355X only (no CPU fallback)")
        self.store = ParamStore(self.device)
        self.gen = torch.Generator().manual_seed(seed)
        self._scratch = None
        self._bufs = {}
        self.convs, self.bns = [], []
        self.training = True
        self.act_dtype = torch.bfloat16   # storage of activations / activation gradients (fp32: set_reference_fp32)
        self.debug_keep = False   # tests: give every unit its own gradient buffers (no reuse)
        self.debug_share_serial = False   # tests: the CU-sharing schedule's exact launches (CU counts, budgets) on ONE stream
        self.debug_join_each_unit = False # A/B: join the side stream at the top of every unit (the schedule before round 3)
        self.fuse_stats = True    # BN sums come out of the producing conv's epilogue (no stats pass)
        self.fuse_eval = True     # inference: eval-mode BN + activation folded into the conv epilogue
        self.share_bn2_tensors, self.share_bn1_tensors = 5, 6   # tensor passes of the confined BatchNorm backward (CU plan)
        self.fuse_bn_fold = True  # CU-confined BatchNorm backward: fold of the sums inside the elementwise pass (2 launches)
        self.fuse_dw_bn_bwd = True  # MBConv: BatchNorm-backward sums in the depthwise data gradient's epilogue (A/B)
        self.fuse_bn1_bwd = True  # ResNet basic block: bn1's backward sums in conv2's data-gradient epilogue (A/B)
        self._side = None         # second stream for weight gradients (WRNEngine turns it on)
        self._cu_share = None     # set_cu_share(): BatchNorm-backward passes beside weight gradients on disjoint CUs
        self._share_join = False
        self._share_split = (False, 250.0)
        self._share_calibrated = True
        self.cu_share_report = None
        self._overlap = True
        self.use_seg = True       # shape-changing units on the slice-list kernel (conv_seg.hip); False: rounds 1-5's launches
        self._seg_ops = {}
        self.seg_share = True     # ... with bn1's backward beside conv1's weight gradient on disjoint CUs (strided units; "all": every one; False: none)
        self.seg_join = None      # wait for conv2's weight gradient before the slice-list data gradient: "small" (8x8 grids), "all", None
        # CU sharing, per stage: dense units whose output grid has at least this many pixels take their BatchNorm-backward
        # sums from the data gradient's epilogue again (the round-1 fused form) and confine only the elementwise pass
        # (0 = every stage uses the split form).  Where the confined reduce + apply outlast the weight gradient beside them
        # (32x32: 290-310 us against 250-300), the sums cost less at MFMA price than on the critical chain.
        self.share_fused_hw = 0
        self.share_stage_us = {}  # output-grid pixels (ho * wo) -> split_target_us of that stage (absent: set_cu_share's)
        self.share_scale_batch = True   # ... scaled by batch / 512 (False: rounds 3-5, the same microseconds at every batch)
        self.share_fused_target_us = 200.0

    def seg_op(self, key, build, sources, sources32, on_side=False):
        """The SegOp `key` (created, and its weights tiled, at first use: a new batch size or image size)."""
        op = self._seg_ops.get(key)
        if op is None:
            op = SegOp(build(), sources, sources32, on_side)
            if self.act_dtype != torch.float32:
                if on_side and self._side is not None and self._overlap:
                    self._side.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(self._side):
                        op.retile()
                    torch.cuda.current_stream(self.device).wait_stream(self._side)
                else:
                    op.retile()
            self._seg_ops[key] = op
        return op

    def _retile_seg(self, on_side):
        if self.act_dtype == torch.float32:
            return
        for op in self._seg_ops.values():
            if op.on_side == on_side:
                op.retile()

    def s2d_buf(self, key, B, H, W, C):
        k = (key, "s2d", B, H, W, C)
        if k not in self._bufs:
            self._bufs[k] = ops.s2d_buffer(B, H, W, C, self.device, self.act_dtype)
        return self._bufs[k]

    def scratch(self, C):
        need = ops.BN_SLOTS * 2 * C
        if self._scratch is None or self._scratch.numel() < need:
            # zero on entry is the C-ABI contract; the fold kernels re-zero what they read
            self._scratch = torch.zeros(max(need, ops.BN_SLOTS * 2 * 2048), device=self.device)
        return self._scratch

    def slot_pair(self, C):
        """(slots, slots_other) for ops.bn_bwd_cus' two-launch form: two 32-slot buffers PER CHANNEL COUNT that swap roles
        at every call (the pass leaves the one it summed into dirty and zeroes the other one -- for ITS channel count, hence
        one pair per C).  Every user is a launch on the caller's stream, in program order.  The kernel's contract is
        "`slots` is zero on entry": a step makes an ODD number of calls per channel count (2n-1 for a WRN stage), so the
        pair does not return to its starting state by itself -- backward() calls reset_slot_pairs() first, which makes
        every step (and every captured graph of one) start from (zero, zero) whatever ran before it."""
        pairs = self.__dict__.setdefault("_slot_pairs", {})
        if C not in pairs:
            n = ops.BN_SLOTS * 2 * C
            # [buffer 0, buffer 1, index of the buffer the LAST call summed into (dirty), or -1: both are zero]
            pairs[C] = [torch.zeros(n, device=self.device), torch.zeros(n, device=self.device), -1]
        pair = pairs[C]
        use = 0 if pair[2] != 0 else 1          # sum into the buffer the last call zeroed
        pair[2] = use
        return pair[use], pair[use ^ 1]

    def reset_slot_pairs(self):
        """Zero the one dirty buffer of every slot pair (a stream-ordered fill on the caller's stream) so that the next
        slot_pair() call finds (zero, zero).  Called at the top of backward(): a hipGraph of a step bakes the pointers
        in, and without this replay k+1's first call per channel count summed into what replay k's last call left."""
        pairs = self.__dict__.get("_slot_pairs", {})
        if not pairs:
            return
        if self.__dict__.get("_slot_arena_pairs") != len(pairs):
            # (re)pack every pair into ONE zeroed arena, so that a step pays one fill launch for all channel counts instead of
            # one each (three per WRN-28-10 step: 15 us + their launch gaps between forward and backward).  Only when a new
            # channel count appeared -- the first backward of a shape; GraphedStep's warm-up steps come before its capture.
            arena = torch.zeros(sum(2 * p[0].numel() for p in pairs.values()), device=self.device)
            off = 0
            for p in pairs.values():
                n = p[0].numel()
                p[0], p[1], p[2] = arena[off:off + n], arena[off + n:off + 2 * n], -1
                off += 2 * n
            self._slot_arena, self._slot_arena_pairs = arena, len(pairs)
            return
        if any(p[2] >= 0 for p in pairs.values()):
            self._slot_arena.zero_()
            for p in pairs.values():
                p[2] = -1

    def partials(self, out):
        """Workspace for the conv-epilogue BN partial sums of a padded [B,H+2,W+2,C] output."""
        B, Hp, Wp, C = out.shape
        need = ((B * (Hp - 2) * (Wp - 2) + 255) // 256) * 2 * C
        if getattr(self, "_partials", None) is None or self._partials.numel() < need:
            self._partials = torch.empty(need, device=self.device)
        return self._partials

    def buf(self, key, B, H, W, C):
        k = (key, B, H, W, C)
        if k not in self._bufs:
            self._bufs[k] = ops.padded(B, H, W, C, self.device, self.act_dtype)
        return self._bufs[k]

    def conv(self, name, cin, cout, k, stride, init="kaiming_a0"):
        c = Conv(self.store, name, cin, cout, k, stride, self.gen, init)
        if self._side is not None:
            c.side_stream = self._side
        self.convs.append(c)
        return c

    def set_reference_fp32(self, on=True):
        """VERIFICATION ONLY.  Switch the storage of every activation / activation-gradient buffer to fp32: nbdt.ops then
        routes each launch on such a buffer to the plain fp32 kernel of the same meaning (csrc/ref_fp32.hip) while this
        engine keeps doing exactly what it does in production -- the same forward() / backward() code, launch order,
        two streams, events, rotating buffers, fused-statistics and CU-sharing protocol.  With fp32 storage a whole
        training step must agree with the fp32 oracle to ~1e-5; with bf16 storage the same path agrees to a gradient
        cosine of ~0.9 (mask flips from 1-ulp roundings).  The difference between the two is storage precision and
        nothing else -- that is the statement tests/test_reference_fp32_gpu.py makes.  Slow (one thread per output);
        small models and batches only.  Inference fusion (conv + folded BatchNorm) has no fp32 twin: eval-mode
        forwards run unfused."""
        self.join_side_stream()
        self.act_dtype = torch.float32 if on else torch.bfloat16
        self.fuse_eval = not on
        self._bufs = {}
        self._seg_ops = {}
        self.__dict__.pop("_seg_skip", None)
        for c in self.convs:
            c._plans = {}
        self._share_calibrated = True        # (a timing decision between two schedules of the SLOW kernels means nothing)
        self.refresh_derived_weights()

    def set_overlap(self, on):
        """Turn the second (weight-gradient) stream on/off at run time; off = every launch on the caller's stream,
        so a kernel's event-bracketed duration is its own (bench.py's roofline pass)."""
        self.join_side_stream()
        for c in self.convs + list(getattr(self, "dws", [])):
            c.side_stream = self._side if on else None
        self._overlap = bool(on)

    def set_cu_share(self, gbps_per_cu=47.0, target_us=200.0, min_cus=16, max_cus=128, join=False, calibrate=True,
                     split_reduce=True, split_target_us=190.0):
        """Run the BatchNorm-backward pass of every fused WRN unit BESIDE the weight gradient of the same conv, on
        disjoint CUs (gbps_per_cu=None: off -- weight gradients next to the data gradients, every pass on all CUs).
        WRNEngine turns this on at construction (calibrated, see below), so main.py / HipBackbone users and bench.py
        run the same schedule.

        Why it pays (probes/cu_share_probe.hip, profiles/r02_cu_share_probe.txt): the pass is HBM-bound and HBM needs
        few CUs -- one CU streams ~47 GB/s, 64 CUs 3.0 TB/s, all 256 5.6 TB/s -- while the MFMA-bound weight gradient
        loses LESS than its share of CUs when it gives some up, because the chip is power-limited (192 CUs deliver
        83 % of the 256-CU matrix rate, and a streaming kernel on the other 64 does not slow them).  The two kernels
        can never share a CU (a weight-gradient block takes its whole register file), so the split is by CU: the pass
        runs as n persistent one-per-CU blocks (nbdt_bn_bwd_apply_cus), the weight gradient is sized for the rest
        (nbdt_wgrad_desc.cu_budget).  n = bytes of the pass / (gbps_per_cu x target_us), clamped to
        [min_cus, max_cus] and then set by the weight gradient's actual block count per XCD (ops.plan_cu_share).
        split_reduce: the BatchNorm-backward SUMS move there too -- the data gradients run with their plain epilogue
        (the fused one reads the BatchNorm input in an HBM burst while the matrix pipes wait: 240 instead of 190 us
        per stage-1 launch) and the confined work becomes reduce + fold + apply (nbdt_bn_bwd_reduce_cus +
        nbdt_bn_bwd_apply_cus: 5-6 tensor passes instead of 3-4, so n is larger: split_target_us).  Same-box A/B at
        512 images: 19.25 ms per step without sharing, 18.45 with the fused sums, 17.55 with the split
        (profiles/r03_split_cu_share_ab.txt).
        join: wait for the weight gradient before the next data gradient (bounds the cost of an unbalanced pair to
        max(pass, weight gradient); measured 1 % slower when the pairs are balanced, so off by default).
        calibrate: calibrate_cu_share() runs before the first backward() at a new setting (or when the caller invokes
        it): it times the conv2 half of one stage-1 unit in the default order and in the order that would be used, on
        the engine's own buffers, and keeps the sharing only if it is faster -- e.g. not when the two streams were
        mapped to one hardware queue and cannot overlap at all, where a pass confined to 50 CUs costs 35 % of a step.
        Only the second stream the engine was created with is used: streams created later wrapped onto the main
        stream's hardware queue in one of four tries (measured), never the first one.
        The CU arithmetic is MI355X's (256 CUs = 8 XCDs x 32): on any other device the sharing stays off."""
        self.join_side_stream()
        self.cu_share_report = None
        self._share_join = bool(join)
        self._share_split = bool(split_reduce), float(split_target_us)
        if gbps_per_cu is not None and not ops.cu_topology_is_mi355x(self.device):
            self._cu_share = None
            self._share_calibrated = True
            self.cu_share_report = {"enabled": False, "reason": "not a 256-CU / 8-XCD device: the CU split is MI355X's"}
            return
        self._cu_share = None if gbps_per_cu is None else (float(gbps_per_cu), float(target_us),
                                                           max(8, int(min_cus)), int(max_cus))
        self._share_calibrated = not calibrate

    def _split_us(self, grid_pixels, B):
        """Time budget of the confined reduce + apply of a stage whose output grid has `grid_pixels` pixels: share_stage_us
        (or set_cu_share's split_target_us), quoted at 512 images -- the weight gradient beside the pass lasts in proportion
        to the batch, so the pass's budget does too and the CU split does not depend on the batch (a 256-image shard with
        512-image microseconds gave its passes half the CUs: 10.6 instead of 9.3 ms per step)."""
        return self.share_stage_us.get(grid_pixels, self._share_split[1]) * (B / 512.0 if self.share_scale_batch else 1.0)

    def _share_plan(self, conv, x, elements, tensors, us=None):
        """(weight-gradient descriptor, its CU budget, CUs for the elementwise pass of `tensors` tensors of `elements`
        bf16 that runs beside it)."""
        gbps, us0, lo, hi = self._cu_share
        B, Hp, Wp, _ = x.shape
        desc = conv.plan(B, Hp - 2, Wp - 2)[3]
        budget, n = ops.plan_cu_share(desc, elements, tensors, gbps, us0 if us is None else us, lo, hi)
        return desc, budget, n

    def _share_pair(self, conv, x, gout, elements, tensors, us=None):
        """Issue conv's weight gradient on the second stream next to the pass that follows; returns the pass's CUs."""
        _, budget, n = self._share_plan(conv, x, elements, tensors, us)
        conv.backward_weight(x, gout, cu_budget=budget)
        return n

    def _calibration_unit(self):
        """The unit whose conv2 pair the calibration times (subclasses with CU sharing override)."""
        return None

    def calibrate_cu_share(self, comm=None):
        """Decide whether the CU-sharing schedule set by set_cu_share() is kept on THIS box: time the conv2 half of one
        widest-tensor unit -- data gradient, BatchNorm backward, weight gradient, exactly the launches backward()
        would make -- in the default order and in the sharing order that is configured (split or fused sums), each
        4 times on the engine's own buffers (about 3 ms, once), and keep the sharing only if it is at least 6 %
        faster (the pair-to-step fit in the code below).  Needs a training-mode forward() at the batch size that will be trained (it uses that forward's
        activations and BatchNorm statistics); backward() calls it before its first launch when a new setting has not
        been calibrated yet.  It synchronises with the host, so it must not run inside a hipGraph capture (GraphedStep
        warms up, and thereby calibrates, before it captures).
        comm (a GradComm of more than one rank): every rank measures, rank 0's decision is broadcast and adopted by
        all, so the ranks of a data-parallel job never run different schedules (the slowest would set the pace)."""
        self._share_calibrated = True
        if self._cu_share is None:
            return self.cu_share_report
        if ops.is_deterministic():
            # The keep/discard decision is a TIMING measurement, and the two schedules sum the BatchNorm-backward
            # reductions in different orders: a run-to-run different decision would break "same launches + same
            # inputs => same bits".  Deterministic mode therefore pins the configured schedule and measures nothing.
            self.cu_share_report = {"enabled": True, "decided_by": "deterministic mode (schedule pinned, not timed)"}
            return self.cu_share_report
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("calibrate_cu_share() synchronises with the host: call it (or run one eager step) "
                               "before capturing the step in a hipGraph")
        u = self._calibration_unit()
        if u is None or self._side is None or getattr(self, "_B", None) is None:
            self._cu_share = None
            self.cu_share_report = {"enabled": False, "reason": "no unit / second stream / forward to calibrate on"}
            return self.cu_share_report
        B = self._B
        k, cout = u["key"], _pad32(u["cout"])
        x_out = u["x_out"]
        h, w = x_out.shape[1] - 2, x_out.shape[2] - 2
        conv, bn = u["conv2"], u["bn2"]
        a2, t = self.buf(k + ".a2", B, h, w, cout), self.buf(k + ".t", B, h, w, cout)
        ga2, gt = self.buf(f"ga2_{cout}", B, h, w, cout), self.buf(f"gt_{cout}_0", B, h, w, cout)
        g = self.buf(f"g_in{cout}_{h}_0", B, h, w, cout)      # backward() rewrites all three before it reads them
        # (plumbing: a one-off fill so the MFMAs see real data -- from a private generator: the caller's global
        # RNG stream is not advanced by a calibration)
        ops.interior(g).normal_(0.0, 1e-3, generator=torch.Generator(device=self.device).manual_seed(0x5eed))
        split, split_us = self._share_split[0], self._split_us(h * w, B)
        elements = B * h * w * cout
        desc, budget, n = self._share_plan(conv, a2, elements, 5 if split else 3, split_us if split else None)
        main = torch.cuda.current_stream(self.device)
        dw = torch.zeros_like(self.store.g(conv.name))
        dsum = torch.empty(2 * bn.C, device=self.device)
        dg, db = torch.zeros(bn.C, device=self.device), torch.zeros(bn.C, device=self.device)
        partials, scratch = self.partials(t), self.scratch(bn.C)
        plan = conv.plan(B, h, w)
        d_plain = plan[1][0]

        def dgrad(fused):
            if fused:
                ops.conv_igemm_bnbwd(d_plain, g, conv.wd, ga2, t, bn.mean, bn.rstd, bn.gamma, bn.beta, partials)
            else:
                ops.conv_igemm(d_plain, g, conv.wd, ga2)

        def wgrad_side(cu_budget):
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                ops.conv_wgrad(desc, a2, g, dw, cu_budget)

        def default_order():        # what backward() does with set_cu_share(None)
            wgrad_side(0)
            dgrad(True)
            ops.bn_bwd_fused(ga2, t, bn.mean, bn.rstd, bn.gamma, bn.beta, partials, dsum, dg, db, gt)
            main.wait_stream(self._side)

        def sharing_order():
            dgrad(not split)
            wgrad_side(budget)
            if split:
                ops.bn_bwd_cus(ga2, t, bn.mean, bn.rstd, bn.gamma, bn.beta,
                               self.slot_pair(bn.C) if self.fuse_bn_fold else scratch, dsum, dg, db, gt, n)
            else:
                ops.bn_bwd_fused(ga2, t, bn.mean, bn.rstd, bn.gamma, bn.beta, partials, dsum, dg, db, gt, cus=n)
            main.wait_stream(self._side)

        def best_us(fn):
            best = float("inf")
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                fn()
                e1.record(main)
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3)
            return best

        t_default, t_share = best_us(default_order), best_us(sharing_order)
        # The isolated pair overstates what the whole step gains: in the default order a weight gradient already hides
        # under the NEXT data gradient, which one pair cannot show.  Measured on WRN-28-10 (scratch/share_by_batch.py;
        # round 4, after the epilogue statistics stopped costing the default order's data gradients 40 us each:
        # profiles/r04_share_by_batch.txt): pair 4.7 % SLOWER -> step 10 % slower (128 images); pair 8-10 % faster ->
        # step +2.7 ... +3.7 % (256); 18 % -> +5.7 % (384); 29 % -> +12.7 % (512) -- break-even near 5 %, so the
        # sharing is kept from 6 % up (round 3's data put the break-even at 8 %).
        keep = t_share < 0.94 * t_default
        decided_by = "this rank"
        if comm is not None and comm.world_size > 1:
            keep = comm.broadcast_flag(keep, self.device)
            decided_by = "rank 0 (broadcast)"
        self.cu_share_report = {"pass_cus": n, "wgrad_cu_budget": budget, "default_order_us": round(t_default, 1),
                                "sharing_order_us": round(t_share, 1), "enabled": bool(keep), "decided_by": decided_by,
                                "timed": f"conv2 half of unit {k} (data gradient + BatchNorm backward + weight "
                                         f"gradient) at batch {B}",
                                "bn_sums": "beside the weight gradient (nbdt_bn_bwd_reduce_cus)" if split
                                           else "data-gradient epilogue"}
        if not keep:
            self._cu_share = None
        return self.cu_share_report

    def _reserve_for(self, comm):
        """While gradient buckets are in flight the collective's kernels hold `comm.reserved_cus` CUs (one RCCL block
        per channel): the one-block-per-CU MFMA launches that follow are sized for the rest (ops.set_reserved_cus), so
        none of their persistent blocks waits for a CU an all-reduce block holds.  comm=None: back to the whole chip."""
        n = int(getattr(comm, "reserved_cus", 0) or 0) if comm is not None else 0
        if n != getattr(self, "_reserved_now", 0):
            ops.set_reserved_cus(n)
            self._reserved_now = n

    def join_side_stream(self):
        """Order every weight-gradient launch issued on the side stream before what follows on the main one."""
        if self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)

    def bn(self, name, c):
        b = BatchNorm(self.store, name, c, self)
        self.bns.append(b)
        return b

    def finalize(self):
        self.store.finalize()
        # one flat buffer for every conv's data-gradient weight copy + the layer table of the batched
        # transpose kernel (one launch per optimizer step instead of one per layer)
        total = sum(c.numel() for c in self.convs)
        self._wd_flat = torch.empty(total, dtype=torch.bfloat16, device=self.device)
        rows, off, tiles = [], 0, 0
        for c in self.convs:
            rows.append([self.store.entries[c.name][0], off, c.cout, c.taps, c.cin, tiles])
            c.wd = self._wd_flat[off:off + c.numel()].view(c.cin, c.taps, c.cout)
            off += c.numel()
            tiles += c.taps * ((c.cout + 63) // 64) * (c.cin // 32)
        self._wd_table = torch.tensor(rows, dtype=torch.int64, device=self.device)
        self._wd_total = tiles
        # DMA-ordered weight tiles for the dense 3x3 / stride-1 kernel (forward weights from the bf16 mirror, data-
        # gradient weights from the transposed copy): one 1-KiB LDS-DMA instruction then reads one contiguous KiB
        dense = [c for c in self.convs if c.k == 3 and c.stride == 1]
        self._wt_n = len(dense)
        if dense:
            n_el = sum(c.numel() for c in dense)
            self._wt_fwd = torch.empty(n_el, dtype=torch.bfloat16, device=self.device)
            self._wt_dgrad = torch.empty(n_el, dtype=torch.bfloat16, device=self.device)
            frows, drows, off, tf, td = [], [], 0, 0, 0

            def ntile(r):
                r32 = r // 32
                return 5 if r32 % 5 == 0 else 4 if r32 % 4 == 0 else 2 if r32 % 2 == 0 else 1

            wd_off = {}
            o = 0
            for c in self.convs:
                wd_off[c.name] = o
                o += c.numel()
            for c in dense:
                frows.append([self.store.entries[c.name][0], off, c.cout, c.cin, tf])
                drows.append([wd_off[c.name], off, c.cin, c.cout, td])
                c.wt_fwd = self._wt_fwd[off:off + c.numel()]
                c.wt_dgrad = self._wt_dgrad[off:off + c.numel()]
                off += c.numel()
                tf += (c.cout // (32 * ntile(c.cout))) * (c.cin // 32) * 9
                td += (c.cin // (32 * ntile(c.cin))) * (c.cout // 32) * 9
            self._wt_ftable = torch.tensor(frows, dtype=torch.int64, device=self.device)
            self._wt_dtable = torch.tensor(drows, dtype=torch.int64, device=self.device)
            self._wt_ftotal, self._wt_dtotal = tf, td
        self.refresh_derived_weights()

    def refresh_derived_weights(self):
        for b in self.bns:             # gamma / beta may have changed: drop the folded eval transforms
            b._affine = None
        if self.act_dtype == torch.float32:      # reference mode: fp32 data-gradient weights (plumbing on parameters)
            for c in self.convs:
                c.wd32 = self.store.p(c.name).flip(1).permute(2, 1, 0).contiguous()
        if self._wt_n:     # forward tiles are needed by the very next forward: caller's stream
            ops.weight_tile_batched(self.store.bf16, self._wt_ftable, self._wt_n, self._wt_ftotal, self._wt_fwd)
        self._retile_seg(False)
        if self._side is None or not getattr(self, "_overlap", True):
            ops.weight_prep_batched(self.store.flat, self._wd_table, len(self.convs), self._wd_total, self._wd_flat)
            if self._wt_n:
                ops.weight_tile_batched(self._wd_flat, self._wt_dtable, self._wt_n, self._wt_dtotal, self._wt_dgrad)
            self._retile_seg(True)
            return
        # the transposed copies are only read by data-gradient launches: build them on the second stream while
        # the next forward runs (backward() joins the stream before its first dgrad)
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side):
            ops.weight_prep_batched(self.store.flat, self._wd_table, len(self.convs), self._wd_total, self._wd_flat)
            if self._wt_n:
                ops.weight_tile_batched(self._wd_flat, self._wt_dtable, self._wt_n, self._wt_dtotal, self._wt_dgrad)
            self._retile_seg(True)

    def sgd_step(self, lr, momentum=0.9, weight_decay=5e-4, grad_scale=1.0, zero_grad=False):
        """optim.SGD step over every parameter (main.py:207) + refresh of bf16 / dgrad weights.
        zero_grad: the same pass leaves the gradient buffer zeroed, and the next zero_grad() is then free (the separate
        146 MB fill took 108 us per WRN-28-10 step; train_step asks for it)."""
        self.join_side_stream()
        s = self.store
        ops.sgd_step(s.flat, s.grad, s.mom, lr, momentum, weight_decay, grad_scale, s.bf16, zero_grad=zero_grad)
        self._grad_is_zero = bool(zero_grad)
        self.refresh_derived_weights()

    def zero_grad(self):
        if not getattr(self, "_grad_is_zero", False):
            self.store.zero_grad()
        self._grad_is_zero = False       # whatever runs next may accumulate into it

    # ---- logical (reference-named) parameter / buffer views for state_dict compatibility
    def extra_param_views(self, buf):
        """name -> view for parameters that are not Conv/BatchNorm objects (stem, classifier)."""
        return {}

    def named_params(self, which="flat"):
        buf = {"flat": self.store.flat, "grad": self.store.grad, "bf16": self.store.bf16}[which]
        out = dict(self.extra_param_views(buf))
        for c in self.convs:
            out[c.name] = c.logical(buf)
        for b in self.bns:
            out[b.name + ".weight"] = self.store._view(buf, b.name + ".weight")[:b.c_real]
            out[b.name + ".bias"] = self.store._view(buf, b.name + ".bias")[:b.c_real]
        return out

    def named_buffers(self):
        out = {}
        for b in self.bns:
            out[b.name + ".running_mean"] = b.running_mean[:b.c_real]
            out[b.name + ".running_var"] = b.running_var[:b.c_real]
            out[b.name + ".num_batches_tracked"] = torch.tensor(b.num_batches_tracked, dtype=torch.long)
        return out

    def load_state_dict(self, sd):
        """Copy a reference-named state dict (fp32 tensors, any device) into the engine."""
        params, bufs = self.named_params("flat"), self.named_buffers()
        missing = {m for m in (set(params) | set(bufs)) - set(sd) if not m.endswith("num_batches_tracked")}
        if missing:
            raise KeyError(f"state dict is missing {sorted(missing)[:5]} ...")
        for name, view in list(params.items()) + list(bufs.items()):
            if name in sd and not name.endswith("num_batches_tracked"):
                view.copy_(sd[name].to(self.device))
        for b in self.bns:
            if b.name + ".num_batches_tracked" in sd:
                b.num_batches_tracked = int(sd[b.name + ".num_batches_tracked"])
        self.store.refresh_bf16()
        self.refresh_derived_weights()

    def state_dict(self):
        out = {k: v.detach().clone().contiguous() for k, v in self.named_params("flat").items()}
        out.update({k: v.detach().clone() for k, v in self.named_buffers().items()})
        return out

    def _tensor(self, key, shape):
        k = (key,) + tuple(shape)
        if k not in self._bufs:
            self._bufs[k] = torch.empty(shape, dtype=torch.float32, device=self.device)
        return self._bufs[k]

    def _zeroed(self, key, shape):
        """fp32 accumulator under the zero-on-entry / re-zeroed-by-the-last-reader contract (like the BatchNorm slots)."""
        k = (key, "zeroed") + tuple(shape)
        if k not in self._bufs:
            self._bufs[k] = torch.zeros(shape, dtype=torch.float32, device=self.device)
        return self._bufs[k]


class WRNEngine(_Engine):
    """Pre-activation WideResNet (pytorchcv ``CIFARWRN``: wrn28_10_cifar10/100, reference
    nbdt/models/wideresnet.py:1-5).  State-dict names follow pytorchcv."""

    def __init__(self, num_classes=10, blocks=28, width_factor=10, in_size=(32, 32), device="cuda", seed=0):
        super().__init__(device, seed)
        assert (blocks - 4) % 6 == 0
        self.num_classes = num_classes
        self.in_size = in_size
        layers = [(blocks - 4) // 6] * 3
        widths = [16 * width_factor, 32 * width_factor, 64 * width_factor]
        self.stem_c = 16
        self.stem_cpad = _pad32(16)
        gen = self.gen
        bound = math.sqrt(2.0) * math.sqrt(3.0 / 27)
        self.store.add("features.init_block.weight", (self.stem_c, 3, 3, 3),
                       lambda v: v.uniform_(-bound, bound, generator=gen))
        self.units = []
        cin = 16
        for i, (n, cout) in enumerate(zip(layers, widths)):
            for j in range(n):
                stride = 2 if (j == 0 and i != 0) else 1
                pre = f"features.stage{i + 1}.unit{j + 1}."
                u = {
                    "bn1": self.bn(pre + "body.conv1.bn", cin),
                    "conv1": self.conv(pre + "body.conv1.conv.weight", cin, cout, 3, stride),
                    "bn2": self.bn(pre + "body.conv2.bn", cout),
                    "conv2": self.conv(pre + "body.conv2.conv.weight", cout, cout, 3, 1),
                    "idconv": (self.conv(pre + "identity_conv.weight", cin, cout, 1, stride)
                               if (cin != cout or stride != 1) else None),
                    "cin": cin, "cout": cout, "stride": stride, "key": f"s{i + 1}u{j + 1}",
                }
                self.units.append(u)
                cin = cout
        self.feat_c = cin
        self.post_bn = self.bn("features.post_activ.bn", cin)
        kb = 1.0 / math.sqrt(cin)   # nn.Linear default init
        self.store.add("output.weight", (num_classes, cin), lambda v: v.uniform_(-kb, kb, generator=gen))
        self.store.add("output.bias", (num_classes,), lambda v: v.uniform_(-kb, kb, generator=gen))
        self.finalize()
        self._img = None
        # Weight gradients on a second HIP stream: a wgrad block (55 KB LDS, 4 waves) and a 256-pixel igemm block
        # (76 KB, 4 waves) fit on one CU together, and the two kernels stall on different things, so running
        # conv.wgrad next to the dgrad / BatchNorm-backward chain instead of in front of it is worth 3.9 % of the
        # step (21.62 -> 20.80 ms, same-box A/B; engine.set_overlap(False) restores the single-stream order).
        self._side = side_stream(self.device)     # engine.set_overlap(False) puts everything back
        for c in self.convs:                                   # on the caller's stream (profiling passes)
            c.side_stream = self._side
        # BatchNorm-backward passes beside the weight gradients on disjoint CUs: on by default, kept only if the
        # calibration before the first backward() finds it faster on this box (set_cu_share(None) turns it off)
        self.set_cu_share(47.0)
        # per stage (output-grid pixels -> split_target_us, quoted at 512 images): the 32x32 and 16x16 stages are bound by the
        # main stream's chain (data gradient, then the confined reduce + apply), the 8x8 stage by its MFMA work.  With plain
        # loads in the passes the chain-bound stages wanted more CUs (170 / 150 us: 104-120 / 72); with the nontemporal loads
        # of round 6 the passes are faster and 190 / 170 us (96-112 / 56-72 CUs) is best again -- the value of the per-stage
        # table is now the batch scaling (_split_us) and a place to retune (profiles/r06_stage_target_ab.txt, sections 1-5)
        self.share_stage_us = {1024: 190.0, 256: 170.0}

    # ---- the shape-changing units on the slice-list kernel (csrc/conv_seg.hip)
    def _seg_conv1(self, u, B, h, w):
        """stride-2 conv1 forward over the space-to-depth a1."""
        c = u["conv1"]
        return self.seg_op((u["key"], "conv1", B, h, w), lambda: ops.seg_fwd_s2(B, h, w, c.cin, c.cout),
                           lambda: [self.store.pb(c.name).view(c.cout, 9 * c.cin)],
                           lambda: [self.store.p(c.name).view(c.cout, 9 * c.cin)])

    def _seg_conv2_sc(self, u, B, ho, wo):
        """conv2 + shortcut in one launch (the residual add disappears into the K loop); None where the plan would fall
        back to 256-pixel half tiles (8x8 grids cannot hold three halo buffers of a 512-pixel tile: the dense kernel + a
        separate 1x1 launch are faster there; engine.seg_fuse_half_tiles = True fuses anyway, A/B)."""
        c2, ci = u["conv2"], u["idconv"]
        key = (u["key"], "conv2sc", B, ho, wo)
        if key in self._seg_ops:
            return self._seg_ops[key]
        skip = self.__dict__.setdefault("_seg_skip", set())
        if key in skip:
            return None
        try:
            plan = ops.seg_conv3x3_plus_1x1(B, ho, wo, c2.cin, c2.cout, ci.cin, (4 if u["stride"] == 2 else 1) * ci.cin)
        except _C.NBDTHipError:
            plan = None
        if plan is None or (plan.tile != 512 and not getattr(self, "seg_fuse_half_tiles", False)):
            skip.add(key)
            return None
        return self.seg_op(key, lambda: plan,
                           lambda: [self.store.pb(c2.name).view(c2.cout, 9 * c2.cin), self.store.pb(ci.name).view(ci.cout, ci.cin)],
                           lambda: [self.store.p(c2.name).view(c2.cout, 9 * c2.cin), self.store.p(ci.name).view(ci.cout, ci.cin)])

    def _seg_dgrad(self, u, B, hi, wi):
        """dL/d(a1) = conv1's data gradient (four parity classes when strided) + the shortcut's, one launch."""
        c1, ci = u["conv1"], u["idconv"]
        if u["stride"] == 2:
            build = lambda: ops.seg_dgrad_s2(B, hi, wi, c1.cin, c1.cout, shortcut=True)
        else:
            build = lambda: ops.seg_dgrad3x3_plus_1x1(B, hi, wi, c1.cin, c1.cout)
        return self.seg_op((u["key"], "dgrad", B, hi, wi), build,
                           lambda: [c1.wd.view(c1.cin, 9 * c1.cout), ci.wd.view(ci.cin, ci.cout)],
                           lambda: [c1.wd32.view(c1.cin, 9 * c1.cout), ci.wd32.view(ci.cin, ci.cout)], on_side=True)

    def _calibration_unit(self):
        for u in self.units:      # a widest-tensor unit without a shape change: both of its convs are dense 3x3
            if u["idconv"] is None and u["cout"] == self.units[0]["cout"]:
                return u
        return None

    # ------------------------------------------------------------------ forward / backward
    classifier_names = ("output.weight", "output.bias")

    def forward(self, img, training=None, head=True):
        """head=False: stop at the pooled features [B, feat_c] (fp32) -- the caller runs the classifier inside the
        fused head + loss kernel (train_step) and hands dL/dpooled to backward(gpooled=...)."""
        training = self.training if training is None else training
        if img.dtype != torch.float32 or not img.is_contiguous():
            img = img.float().contiguous()
        B, _, H, W = img.shape
        self._img, self._B = img, B
        x = self.buf("x0", B, H, W, self.stem_cpad)
        ops.stem_conv(img, self.store.p("features.init_block.weight"), x, self.stem_c)
        h, w = H, W
        x_has_stats = False   # the stem kernel does not produce BN sums
        for u in self.units:
            k, s = u["key"], u["stride"]
            cin, cout = _pad32(u["cin"]), _pad32(u["cout"])
            ho, wo = h // s, w // s
            t = self.buf(k + ".t", B, ho, wo, cout)
            a2 = self.buf(k + ".a2", B, ho, wo, cout)
            out = self.buf(k + ".out", B, ho, wo, cout)
            fuse = training and self.fuse_stats
            u["bn1"].stats(x, training, fused=fuse and x_has_stats)
            # Shape-changing units in training mode (round 6, csrc/conv_seg.hip): the pre-activation of a stride-2 unit is
            # written as its space-to-depth copy -- its only readers are conv1, the shortcut and their weight gradients,
            # all stride 2 -- conv1 runs as a slice list over it, and conv2 takes the shortcut as one more term of its K
            # loop (no shortcut launch, no residual read in the epilogue).
            seg = training and self.use_seg and u["idconv"] is not None
            u["seg"] = seg
            if seg and s == 2:
                a1 = self.s2d_buf(k + ".a1", B, h, w, cin)
                ops.bn_apply_s2d(x, u["bn1"].mean, u["bn1"].rstd, u["bn1"].gamma, u["bn1"].beta, a1, relu=True)
            else:
                a1 = self.buf(k + ".a1", B, h, w, cin)
                u["bn1"].apply(x, a1, relu=True)
            if not training and self.fuse_eval:   # conv1 + bn2 + ReLU in one launch (t is never materialised)
                u["conv1"].forward_affine(a1, a2, u["bn2"], act=1)
            else:
                if seg and s == 2:
                    self._seg_conv1(u, B, h, w)([a1], t, bn_scratch=self.partials(t) if fuse else None)
                else:
                    u["conv1"].forward(a1, t, bn_scratch=self.partials(t) if fuse else None)
                u["bn2"].stats(t, training, fused=fuse)
                u["bn2"].apply(t, a2, relu=True)
            both = self._seg_conv2_sc(u, B, ho, wo) if seg else None
            if both is not None:
                both([a2, a1], out, bn_scratch=self.partials(out) if fuse else None)
            else:
                if u["idconv"] is not None:
                    idn = self.buf(f"idn{cout}", B, ho, wo, cout)
                    if seg and s == 2:
                        u["idconv"].forward_s2d(a1, idn, h, w)
                    else:
                        u["idconv"].forward(a1, idn)
                    res = idn
                else:
                    res = x
                u["conv2"].forward(a2, out, residual=res, bn_scratch=self.partials(out) if fuse else None)
            x_has_stats = True
            u["x_in"], u["x_out"] = x, out
            x, h, w = out, ho, wo
        self._x_last, self._hw = x, (h, w)
        self.post_bn.stats(x, training, fused=training and self.fuse_stats)
        self._pooled = self._tensor("pooled", (B, self.feat_c))
        ops.bn_relu_pool(x, self.post_bn.mean, self.post_bn.rstd, self.post_bn.gamma, self.post_bn.beta,
                         self._pooled)
        if not head:
            return self._pooled
        z = self._tensor("z", (B, self.num_classes))
        ops.linear_fwd(self._pooled, self.store.p("output.weight"), self.store.p("output.bias"), z)
        return z

    def grad_buckets(self):
        """(lo, hi) ranges of the flat gradient buffer in the order backward completes them:
        [stage3 .. classifier], [stage2], [stem .. stage1]."""
        ent = self.store.entries
        s2 = ent["features.stage2.unit1.body.conv1.bn.weight"][0]
        s3 = ent["features.stage3.unit1.body.conv1.bn.weight"][0]
        return [(s3, self.store.grad.numel()), (s2, s3), (0, s2)]

    def backward(self, gz, comm=None, gpooled=None):
        """Accumulates d(loss)/d(params) into the flat gradient buffer given gz = dloss/dz [B, classes] -- or, after
        forward(head=False), given gpooled = dloss/dpooled [B, feat_c] (the fused head kernel already accumulated the
        classifier's gradients).  With a GradComm, each stage's gradient bucket is all-reduced as soon as it is
        complete."""
        self._grad_is_zero = False   # this call accumulates into the gradient buffer
        B = self._B
        self.join_side_stream()      # dgrad weight copies (built on the second stream after the last update)
        if self._cu_share is not None and not self._share_calibrated:
            self.calibrate_cu_share(comm)      # once per set_cu_share(): keep the sharing only if it is faster here
        self.reset_slot_pairs()      # the confined BatchNorm backward's slot buffers start every step from (zero, zero)
        two_streams = self._side is not None and self._overlap
        buckets = self.grad_buckets() if comm is not None else None
        st = self.store
        if gpooled is not None:
            gpool = gpooled
        else:
            gz = gz.contiguous()
            gpool = self._tensor("gpool", (B, self.feat_c))
            ops.linear_bwd(self._pooled, st.p("output.weight"), gz, gpool, st.g("output.weight"), st.g("output.bias"))
        h, w = self._hw
        C = _pad32(self.feat_c)
        g = self.buf(f"g_out{C}", B, h, w, C)
        pb = self.post_bn
        ops.pool_bn_bwd(gpool, self._x_last, pb.mean, pb.rstd, pb.gamma, pb.beta, self.scratch(C), pb.dsum,
                        st.g(pb.name + ".weight"), st.g(pb.name + ".bias"), g)
        toggle, n_unit, side_mark = 0, 0, None
        if two_streams:      # (also forks the side stream into a hipGraph capture before its first event is recorded)
            self._side.wait_stream(torch.cuda.current_stream(self.device))
        for u in reversed(self.units):
            # Gradient buffers are shared between units, and a weight gradient still running on the side stream may be
            # reading what a later unit is about to overwrite: conv1's reads `gt`, conv2's the unit's output gradient
            # (the previous unit's g_in).  Joining the side stream here would order that, but it makes the main
            # stream wait for the weight gradient issued LAST (the previous unit's conv1, ~70 us of a 1 ms unit with
            # 96 CUs idle: 0.95 ms of launch gaps per backward in profiles/r03_split_*two_streams*).  Instead the
            # main stream waits for everything the side stream had been given ONE UNIT AGO (an event recorded at the
            # top of the previous unit), and the buffers a weight gradient of the previous unit can still be reading
            # are not the ones this unit writes: `gt` alternates between two buffers, g_in rotates through three.
            # (tests/test_engine_gpu.py::test_deterministic_mode_makes_the_schedules_bit_comparable: this schedule
            # gives the same bits as one stream with private buffers.)
            if two_streams and self.debug_join_each_unit:
                self.join_side_stream()
            elif two_streams:
                main = torch.cuda.current_stream(self.device)
                if side_mark is not None:
                    main.wait_event(side_mark)
                side_mark = torch.cuda.Event()
                side_mark.record(self._side)
            k, s = u["key"], u["stride"]
            cin, cout = _pad32(u["cin"]), _pad32(u["cout"])
            ho, wo = h, w
            hi, wi = ho * s, wo * s
            seg = bool(u.get("seg"))
            a1 = self.s2d_buf(k + ".a1", B, hi, wi, cin) if (seg and s == 2) else self.buf(k + ".a1", B, hi, wi, cin)
            t = self.buf(k + ".t", B, ho, wo, cout)
            a2 = self.buf(k + ".a2", B, ho, wo, cout)
            tag = ("@" + k) if self.debug_keep else ""
            ga2 = self.buf(f"ga2_{cout}{tag}", B, ho, wo, cout)
            gt = self.buf(f"gt_{cout}_{n_unit & 1}{tag}", B, ho, wo, cout)
            ga1 = self.buf(f"ga1_{cin}_{hi}{tag}", B, hi, wi, cin)
            # the unit's input gradient must not alias its output gradient `g` (nor, see above, the one before that)
            toggle = (toggle + 1) % 3
            n_unit += 1
            g_in = self.buf(f"g_in{cin}_{hi}_{toggle}{tag}", B, hi, wi, cin)
            u["dbg"] = {"g_out": g, "ga2": ga2, "gt": gt, "ga1": ga1, "g_in": g_in}
            fuse = self.fuse_stats
            # CU sharing (set_cu_share): each weight gradient is issued AFTER the data gradient of its conv and sized
            # for 256 - n CUs, and the BatchNorm-backward pass that follows on this stream is confined to n CUs, so
            # the HBM-bound pass and the MFMA-bound kernel run at the same time on disjoint CUs.  In its split form
            # the BatchNorm-backward sums move out of the data gradient's epilogue (which reads the BatchNorm input in
            # an HBM burst while the matrix pipes wait) into the CU-confined pass beside the weight gradient.
            share = self._cu_share is not None and fuse and (two_streams or self.debug_share_serial)
            split = share and self._share_split[0]
            fused_here = bool(split and self.share_fused_hw and ho * wo >= self.share_fused_hw and not seg)
            if fused_here:
                split = False
            split_us = self._split_us(ho * wo, B)
            if self._cu_share is not None and self._share_split[0] and fuse and not share:
                # one-stream mode (profiling passes, bench.py's roofline pass) of the split schedule: the same MFMA
                # kernels as the timed step -- data gradients with their plain epilogue -- and the BatchNorm sums in a
                # pass of their own, back to back on all CUs
                fuse = False
            if split:
                u["conv2"].backward_data(g, ga2)
                n2 = self._share_pair(u["conv2"], a2, g, B * ho * wo * cout, self.share_bn2_tensors, split_us)
                u["bn2"].backward_cus(ga2, t, gt, n2)
                if self._share_join:
                    self.join_side_stream()
            elif fuse:   # the dgrad epilogue also produces bn2's backward sums (ga2 is not re-read for them)
                if not share:
                    u["conv2"].backward_weight(a2, g)
                u["conv2"].backward_data(g, ga2, bn=u["bn2"], bn_x=t, partials=self.partials(t))
                n2 = self._share_pair(u["conv2"], a2, g, B * ho * wo * cout, 3,
                                      self.share_fused_target_us if fused_here else None) if share else 0
                u["bn2"].backward_fused(ga2, t, gt, self.partials(t), cus=n2)
                if share and self._share_join:
                    # The next data gradient is one persistent block per CU with a fixed share of the tiles: blocks
                    # that found their CU still held by the weight gradient would start late and finish late, and
                    # the delay would push the next weight gradient under the next data gradient, and so on (measured:
                    # 19.1 -> 26 ms per step with the passes slightly too fast for the weight gradients).  Waiting
                    # here makes an unbalanced pair cost max(pass, weight gradient), never more.
                    self.join_side_stream()
            else:
                u["conv2"].backward_weight(a2, g)
                u["conv2"].backward_data(g, ga2)
                u["bn2"].backward(ga2, None, t, gt, relu=True)   # mask recomputed from t: a2 not re-read
            if split and u["idconv"] is None:
                x_in = u["x_in"]
                u["conv1"].backward_data(gt, ga1)
                n1 = self._share_pair(u["conv1"], a1, gt, B * hi * wi * cin, self.share_bn1_tensors, split_us)
                u["bn1"].backward_cus(ga1, x_in, g_in, n1, gx_add=g)
                g, h, w = g_in, hi, wi
                continue
            if fuse and u["idconv"] is None:
                x_in = u["x_in"]
                if not share:
                    u["conv1"].backward_weight(a1, gt)
                u["conv1"].backward_data(gt, ga1, bn=u["bn1"], bn_x=x_in, partials=self.partials(x_in))
                n1 = self._share_pair(u["conv1"], a1, gt, B * hi * wi * cin, 4,
                                      self.share_fused_target_us if fused_here else None) if share else 0
                u["bn1"].backward_fused(ga1, x_in, g_in, self.partials(x_in), gx_add=g, cus=n1)
                g, h, w = g_in, hi, wi
                continue
            if seg:
                # conv1's and the shortcut's data gradients in one launch (strided: the four parity classes, the
                # shortcut's gradient one more term of the even/even class); weight gradients read the same a1
                x_in = u["x_in"]
                if split and self.seg_share and (s == 2 or self.seg_share == "all"):
                    # the CU-sharing order of the dense units: data gradient on the whole chip, then conv1's weight
                    # gradient sized for 256 - n CUs beside bn1's backward on n (the unit's input gradient is bn1's alone:
                    # conv1 and the shortcut both read relu(bn1(x)))
                    if self.seg_join == "all" or (self.seg_join == "small" and ho * wo <= 64):
                        # conv2's weight gradient outlasts the short BatchNorm pass it was paired with: the one-block-per-CU
                        # data gradient that follows would find most CUs held and take as long as both
                        self.join_side_stream()
                    self._seg_dgrad(u, B, hi, wi)([gt, g], ga1)
                    desc = u["conv1"].s2d_plan(B, hi, wi)[1] if s == 2 else u["conv1"].plan(B, hi, wi)[3]
                    gbps, _, lo, hi_cus = self._cu_share
                    budget, n1 = ops.plan_cu_share(desc, B * hi * wi * cin, self.share_bn2_tensors, gbps, split_us, lo, hi_cus)
                    if s == 2:
                        u["conv1"].backward_weight_s2d(a1, gt, hi, wi, cu_budget=budget)
                        u["idconv"].backward_weight_s2d(a1, g, hi, wi)
                    else:
                        u["conv1"].backward_weight(a1, gt, cu_budget=budget)
                        u["idconv"].backward_weight(a1, g)
                    u["bn1"].backward_cus(ga1, x_in, g_in, n1)
                else:
                    if s == 2:
                        u["conv1"].backward_weight_s2d(a1, gt, hi, wi)
                        u["idconv"].backward_weight_s2d(a1, g, hi, wi)
                    else:
                        u["conv1"].backward_weight(a1, gt)
                        u["idconv"].backward_weight(a1, g)
                    self._seg_dgrad(u, B, hi, wi)([gt, g], ga1)
                    u["bn1"].backward(ga1, None, x_in, g_in, relu=True)
                g, h, w = g_in, hi, wi
                if comm is not None and u["key"] in ("s3u1", "s2u1"):
                    self.join_side_stream()
                    comm.reduce_range(st.grad, *buckets[0 if u["key"] == "s3u1" else 1])
                    self._reserve_for(comm)
                continue
            u["conv1"].backward_weight(a1, gt)
            u["conv1"].backward_data(gt, ga1)
            if u["idconv"] is not None:
                u["idconv"].backward_weight(a1, g)
                u["idconv"].backward_data(g, ga1, accumulate=True)
                u["bn1"].backward(ga1, None, u["x_in"], g_in, relu=True)
            else:
                u["bn1"].backward(ga1, None, u["x_in"], g_in, relu=True, gx_add=g)
            g, h, w = g_in, hi, wi
            if comm is not None and u["key"] in ("s3u1", "s2u1"):
                self.join_side_stream()
                comm.reduce_range(st.grad, *buckets[0 if u["key"] == "s3u1" else 1])
                self._reserve_for(comm)
        ops.stem_wgrad(self._img, g, st.g("features.init_block.weight"), self.stem_c)
        self.join_side_stream()      # every gradient is complete on the caller's stream when backward returns
        if comm is not None:
            comm.reduce_range(st.grad, *buckets[2])
            comm.finish(st.grad)
            self._reserve_for(None)

    # ------------------------------------------------------------------ reference-named views
    def extra_param_views(self, buf):
        return {
            "features.init_block.weight": self.store._view(buf, "features.init_block.weight").permute(0, 3, 1, 2),
            "output.weight": self.store._view(buf, "output.weight"),
            "output.bias": self.store._view(buf, "output.bias"),
        }


class ResNetEngine(_Engine):
    """CIFAR-style ResNet of the reference (nbdt/models/resnet.py:42-74 BasicBlock, :115-149 ResNet,
    :171-179 ResNet18): 3x3 stem, 4 stages of post-activation BasicBlocks, global average pool,
    ``linear``.  State-dict names are the reference's (conv1, bn1, layerN.M.*, shortcut.0/1, linear)."""

    def __init__(self, num_classes=10, num_blocks=(2, 2, 2, 2), device="cuda", seed=0):
        super().__init__(device, seed)
        self.num_classes = num_classes
        gen = self.gen
        self.stem_c = 64
        b0 = 1.0 / math.sqrt(27)
        self.store.add("conv1.weight", (64, 3, 3, 3), lambda v: v.uniform_(-b0, b0, generator=gen))
        self.bn0 = self.bn("bn1", 64)
        self.blocks = []
        cin = 64
        for i, (cout, stride0, n) in enumerate(zip((64, 128, 256, 512), (1, 2, 2, 2), num_blocks)):
            for j in range(n):
                stride = stride0 if j == 0 else 1
                pre = f"layer{i + 1}.{j}."
                short = stride != 1 or cin != cout
                blk = {
                    "conv1": self.conv(pre + "conv1.weight", cin, cout, 3, stride, init="torch_default"),
                    "bn1": self.bn(pre + "bn1", cout),
                    "conv2": self.conv(pre + "conv2.weight", cout, cout, 3, 1, init="torch_default"),
                    "bn2": self.bn(pre + "bn2", cout),
                    "sconv": self.conv(pre + "shortcut.0.weight", cin, cout, 1, stride, init="torch_default") if short else None,
                    "sbn": self.bn(pre + "shortcut.1", cout) if short else None,
                    "cin": cin, "cout": cout, "stride": stride, "key": f"l{i + 1}b{j}",
                }
                self.blocks.append(blk)
                cin = cout
        self.feat_c = cin
        kb = 1.0 / math.sqrt(cin)
        self.store.add("linear.weight", (num_classes, cin), lambda v: v.uniform_(-kb, kb, generator=gen))
        self.store.add("linear.bias", (num_classes,), lambda v: v.uniform_(-kb, kb, generator=gen))
        self.finalize()
        self._side = side_stream(self.device)     # weight gradients on a second stream (see WRNEngine)
        for c in self.convs:
            c.side_stream = self._side
        # (GB/s per CU, time budget of bn1's confined pass at 128 images in us, min CUs, max CUs) or None: see backward()
        self.res_share = (47.0, 15.0, 16, 128)
        dev = self.device
        # identity "BN" for the plain average-pool head (features are already post-ReLU)
        self._id_mean = torch.zeros(cin, device=dev)
        self._id_rstd = torch.ones(cin, device=dev)
        self._id_gamma = torch.ones(cin, device=dev)
        self._id_beta = torch.zeros(cin, device=dev)
        self._id_dsum = torch.zeros(2 * cin, device=dev)

    def extra_param_views(self, buf):
        return {
            "conv1.weight": self.store._view(buf, "conv1.weight").permute(0, 3, 1, 2),
            "linear.weight": self.store._view(buf, "linear.weight"),
            "linear.bias": self.store._view(buf, "linear.bias"),
        }

    def grad_buckets(self):
        ent = self.store.entries
        l3 = ent["layer3.0.conv1.weight"][0]
        l2 = ent["layer2.0.conv1.weight"][0]
        return [(l3, self.store.grad.numel()), (l2, l3), (0, l2)]

    classifier_names = ("linear.weight", "linear.bias")

    def forward(self, img, training=None, head=True):
        training = self.training if training is None else training
        if img.dtype != torch.float32 or not img.is_contiguous():
            img = img.float().contiguous()
        B, _, H, W = img.shape
        self._img, self._B = img, B
        t0 = self.buf("t0", B, H, W, 64)
        x = self.buf("a0", B, H, W, 64)
        ops.stem_conv(img, self.store.p("conv1.weight"), t0, 64)
        self.bn0.stats(t0, training)
        self.bn0.apply(t0, x, relu=True)
        h, w = H, W
        for blk in self.blocks:
            k, s, cin, cout = blk["key"], blk["stride"], blk["cin"], blk["cout"]
            ho, wo = h // s, w // s
            a1 = self.buf(k + ".a1", B, ho, wo, cout)
            out = self.buf(k + ".out", B, ho, wo, cout)
            fuse = training and self.fuse_stats
            if not training and self.fuse_eval:
                # inference: every Conv-BN(-ReLU)(+shortcut) group of the BasicBlock is ONE launch
                blk["conv1"].forward_affine(x, a1, blk["bn1"], act=1)
                if blk["sconv"] is not None:
                    sc = self.buf(f"sc{cout}", B, ho, wo, cout)
                    blk["sconv"].forward_affine(x, sc, blk["sbn"], act=0)
                    res = sc
                else:
                    res = x
                blk["conv2"].forward_affine(a1, out, blk["bn2"], act=1, residual=res)
                blk["x_in"] = x
                x, h, w = out, ho, wo
                continue
            t1 = self.buf(k + ".t1", B, ho, wo, cout)     # raw conv outputs: only the training path keeps them
            t2 = self.buf(k + ".t2", B, ho, wo, cout)
            scr = self.partials(t1) if fuse else None
            blk["conv1"].forward(x, t1, bn_scratch=scr)
            blk["bn1"].stats(t1, training, fused=fuse)
            blk["bn1"].apply(t1, a1, relu=True)
            blk["conv2"].forward(a1, t2, bn_scratch=scr)
            blk["bn2"].stats(t2, training, fused=fuse)
            if blk["sconv"] is not None:
                ts = self.buf(k + ".ts", B, ho, wo, cout)
                sc = self.buf(f"sc{cout}", B, ho, wo, cout)
                blk["sconv"].forward(x, ts, bn_scratch=scr)
                blk["sbn"].stats(ts, training, fused=fuse)
                blk["sbn"].apply(ts, sc, relu=False)
                res = sc
            else:
                res = x
            blk["bn2"].apply(t2, out, relu=True, residual=res)
            blk["x_in"] = x
            x, h, w = out, ho, wo
        self._x_last, self._hw = x, (h, w)
        self._pooled = self._tensor("pooled", (B, self.feat_c))
        ops.bn_relu_pool(x, self._id_mean, self._id_rstd, self._id_gamma, self._id_beta, self._pooled)
        if not head:
            return self._pooled
        z = self._tensor("z", (B, self.num_classes))
        ops.linear_fwd(self._pooled, self.store.p("linear.weight"), self.store.p("linear.bias"), z)
        return z

    def backward(self, gz, comm=None, gpooled=None):
        self._grad_is_zero = False   # this call accumulates into the gradient buffer
        B = self._B
        self.join_side_stream()      # dgrad weight copies (built on the second stream after the last update)
        st = self.store
        buckets = self.grad_buckets() if comm is not None else None
        if gpooled is not None:      # after forward(head=False): the fused head kernel did the classifier's backward
            gpool = gpooled
        else:
            gz = gz.contiguous()
            gpool = self._tensor("gpool", (B, self.feat_c))
            ops.linear_bwd(self._pooled, st.p("linear.weight"), gz, gpool, st.g("linear.weight"), st.g("linear.bias"))
        h, w = self._hw
        g = self.buf(f"g_out{self.feat_c}", B, h, w, self.feat_c)
        # plain avg-pool backward: identity BN with zero batch sums (the x>0 mask it applies is the
        # same mask the following ReLU backward applies anyway)
        ops.pool_bn_bwd_apply(gpool, self._x_last, self._id_mean, self._id_rstd, self._id_gamma, self._id_beta,
                              self._id_dsum, g)
        toggle, n_blk, side_mark = 0, 0, None
        two_streams = self._side is not None and self._overlap
        if two_streams:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
        for blk in reversed(self.blocks):
            # Shared gradient buffers vs weight gradients still running on the second stream: as in WRNEngine.backward,
            # wait for what the side stream had been given one block ago, and alternate the buffers a weight gradient
            # reads (gt2 / gt1 / gts) between consecutive blocks, so that the block in flight never overwrites them.
            if two_streams:
                main = torch.cuda.current_stream(self.device)
                if side_mark is not None:
                    main.wait_event(side_mark)
                side_mark = torch.cuda.Event()
                side_mark.record(self._side)
            k, s, cin, cout = blk["key"], blk["stride"], blk["cin"], blk["cout"]
            ho, wo = h, w
            hi, wi = ho * s, wo * s
            tag = ("@" + k) if self.debug_keep else ""
            par = n_blk & 1
            n_blk += 1
            t1 = self.buf(k + ".t1", B, ho, wo, cout)
            a1 = self.buf(k + ".a1", B, ho, wo, cout)
            t2 = self.buf(k + ".t2", B, ho, wo, cout)
            out = self.buf(k + ".out", B, ho, wo, cout)
            gt2 = self.buf(f"gt2_{cout}_{par}{tag}", B, ho, wo, cout)
            ga1 = self.buf(f"ga1_{cout}{tag}", B, ho, wo, cout)
            gt1 = self.buf(f"gt1_{cout}_{par}{tag}", B, ho, wo, cout)
            toggle ^= 1
            g_in = self.buf(f"g_in{cin}_{hi}_{toggle}{tag}", B, hi, wi, cin)
            x_in = blk["x_in"]
            if blk["sconv"] is not None:
                gsc = self.buf(f"gsc_{cout}{tag}", B, ho, wo, cout)
                gts = self.buf(f"gts_{cout}_{par}{tag}", B, ho, wo, cout)
                blk["bn2"].backward(g, out, t2, gt2, relu=True, g_resid=gsc)
            else:
                # identity shortcut: the masked gradient IS part of the block-input gradient
                blk["bn2"].backward(g, out, t2, gt2, relu=True, g_resid=g_in)
            share = (self.res_share is not None and (two_streams or self.debug_share_serial) and self.fuse_bn1_bwd
                     and self.act_dtype == torch.bfloat16 and ops.cu_topology_is_mi355x(self.device))
            if share:
                # CU sharing as in WRNEngine (fused-sums form): conv2's weight gradient is issued AFTER its data gradient,
                # sized for 256 - n CUs, and bn1's elementwise pass that follows on this stream is confined to n CUs --
                # in the plain order the pass's blocks waited for the weight gradient's to leave their CUs
                # (bn_bwd_fold_partials: 42 us in the trace for 5 us of work, profiles/r06_c4_step_dump_before.txt)
                gbps, us, lo, hi_cus = self.res_share
                blk["conv2"].backward_data(gt2, ga1, bn=blk["bn1"], bn_x=t1, partials=self.partials(t1))
                budget, n1 = ops.plan_cu_share(blk["conv2"].plan(B, ho, wo)[3], B * ho * wo * cout, 3, gbps,
                                               us * B / 128.0, lo, hi_cus)
                blk["conv2"].backward_weight(a1, gt2, cu_budget=budget)
                blk["bn1"].backward_fused(ga1, t1, gt1, self.partials(t1), cus=n1)
            elif self.fuse_bn1_bwd and self.act_dtype == torch.bfloat16:
                # conv2's data gradient is dL/d(relu(bn1(t1))): its epilogue also emits bn1's backward sums (as in
                # WRNEngine's fused order), so ga1 is not re-read for them -- one HBM-bound pass fewer per block
                blk["conv2"].backward_weight(a1, gt2)
                blk["conv2"].backward_data(gt2, ga1, bn=blk["bn1"], bn_x=t1, partials=self.partials(t1))
                blk["bn1"].backward_fused(ga1, t1, gt1, self.partials(t1))
            else:
                blk["conv2"].backward_weight(a1, gt2)
                blk["conv2"].backward_data(gt2, ga1)
                blk["bn1"].backward(ga1, None, t1, gt1, relu=True)
            blk["conv1"].backward_weight(x_in, gt1)
            if blk["sconv"] is not None:
                blk["conv1"].backward_data(gt1, g_in)
                ts = self.buf(k + ".ts", B, ho, wo, cout)
                blk["sbn"].backward(gsc, None, ts, gts, relu=False)
                blk["sconv"].backward_weight(x_in, gts)
                blk["sconv"].backward_data(gts, g_in, accumulate=True)
            else:
                blk["conv1"].backward_data(gt1, g_in, accumulate=True)
            blk["dbg"] = {"g_out": g, "g_in": g_in}
            g, h, w = g_in, hi, wi
            if comm is not None and k in ("l3b0", "l2b0"):
                self.join_side_stream()
                comm.reduce_range(st.grad, *buckets[0 if k == "l3b0" else 1])
                self._reserve_for(comm)
        gt0 = self.buf("gt0", B, h, w, 64)
        self.bn0.backward(g, None, self.buf("t0", B, h, w, 64), gt0, relu=True)
        ops.stem_wgrad(self._img, gt0, st.g("conv1.weight"), 64)
        self.join_side_stream()
        if comm is not None:
            comm.reduce_range(st.grad, *buckets[2])
            comm.finish(st.grad)
            self._reserve_for(None)


def train_step(engine, criterion, img, targets, lr, momentum=0.9, weight_decay=5e-4, comm=None, fused_head=True,
               zero_grad=True):
    """One full training step (main.py:233-239): zero_grad, forward, SoftTreeSupLoss fwd+bwd (one fused
    kernel -- with the classifier's forward and backward inside it when the criterion and the head's width allow,
    fused_head=False keeps linear -> loss -> linear-backward as three launches), backward, [gradient all-reduce],
    SGD.  Returns the loss tensor (device scalar).  zero_grad=False leaves the step's gradient in the flat gradient
    buffer instead of having the SGD kernel clear it for the next step (tests that compare gradients)."""
    engine.zero_grad()
    names = getattr(engine, "classifier_names", None)
    if (fused_head and names is not None and hasattr(criterion, "can_fuse_head")
            and criterion.can_fuse_head(engine.num_classes)):
        # classifier + rules + loss + their backward in ONE launch (nbdt_head_soft_tree_loss): logits stay on chip
        pooled = engine.forward(img, training=True, head=False)
        st = engine.store
        loss, gpool, _ = criterion.head_loss_and_grad(pooled, st.p(names[0]), st.p(names[1]), targets,
                                                      grad_weight=st.g(names[0]), grad_bias=st.g(names[1]))
        engine.backward(None, comm=comm, gpooled=gpool)
    else:
        z = engine.forward(img, training=True)
        loss, gz = criterion.loss_and_grad(z, targets)
        engine.backward(gz, comm=comm)
    scale = 1.0 / comm.world_size if comm is not None else 1.0
    engine.sgd_step(lr, momentum, weight_decay, grad_scale=scale, zero_grad=zero_grad)
    return loss


class GraphedStep:
    """One training step captured in a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the
    stream every C-ABI launch already uses) and replayed with one host call.

    The step is a fixed sequence of ~130 (ResNet18) to ~700 (EfficientNet-B0) launches with no host
    synchronisation, so it captures as is.  Measured on MI355X (scratch/bench_graph.py, profiles/r04_graph.txt):
    replay is 4-8 % SLOWER than eager launching for every configuration tried (ResNet18 B=128: 3.35 vs 3.13 ms;
    WRN-28-10 B=512: 18.42 vs 17.40 ms; in round 2, before the two-stream schedule, the two were equal at 21.8 ms) --
    the ctypes launch path costs ~5 us per kernel and the CPU stays ahead of the GPU (2.2 ms of enqueue for a 17 ms
    step), while the graph serialises the cross-stream edges the eager schedule overlaps.  So this is latency insurance
    for slower hosts, not a speed-up, and bench.py does not use it.  The learning rate and the dropout
    seed are kernel arguments baked into the graph: build one GraphedStep per learning rate, and do not use it
    for models with dropout.  So are the loss weights and the device pointers of the hierarchy: a replay after
    ``criterion.set_epoch`` changed the weights (--tswe / --xwe schedules) or after the hierarchy was re-induced
    (SoftTreeLoss) raises instead of silently using the captured ones; the captured tree handle is kept alive
    by this object.  Gradient all-reduce (RCCL) is not captured: single-GPU steps only."""

    def __init__(self, engine, criterion, img, targets, lr, momentum=0.9, weight_decay=5e-4, warmup=2):
        if getattr(engine, "dropout_rate", 0.0) > 0.0:
            raise ValueError("the dropout seed is a kernel argument: a captured step would repeat one mask")
        self.engine, self.criterion = engine, criterion
        self.img = img.clone()
        self.targets = targets.clone()
        # Warm up and capture on ONE explicit stream: the library's per-(device, stream) workspaces -- K-split
        # weight-gradient copies, stem weight-gradient rows, split-K partials and tickets -- cannot be allocated inside a
        # capture, so a launch that meets its stream for the first time there silently takes its fallback form (atomics, no
        # split).  torch.cuda.graph() without stream= captures on a private stream the warm-up never ran on.
        self._stream = torch.cuda.Stream(device=self.img.device)
        self._stream.wait_stream(torch.cuda.current_stream(self.img.device))
        with torch.cuda.stream(self._stream):
            for _ in range(warmup):     # allocate every buffer and workspace, set kernel attributes, warm the allocator
                train_step(engine, criterion, self.img, self.targets, lr, momentum, weight_decay)
        torch.cuda.synchronize()
        self._captured = self._criterion_state()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self._stream):
            self.loss = train_step(engine, criterion, self.img, self.targets, lr, momentum, weight_decay)
            engine.join_side_stream()     # a capture must end with every forked stream joined

    def _criterion_state(self):
        """(loss weights, tree handle) the captured launches were built with; holding the handle keeps its device
        memory alive even if the Tree drops it."""
        c = self.criterion
        weights = tuple(float(w) for w in c.current_weights()) if hasattr(c, "current_weights") else None
        tree = getattr(c, "tree", None)
        handle = tree.device_handle(self.img.device.index) if tree is not None else None
        return weights, handle

    def __call__(self, img, targets):
        weights, handle = self._criterion_state()
        if weights != self._captured[0] or handle is not self._captured[1]:
            raise RuntimeError("GraphedStep: the criterion's weights or hierarchy changed since capture; "
                               "build a new GraphedStep (they are baked into the captured launches)")
        self.img.copy_(img, non_blocking=True)
        self.targets.copy_(targets, non_blocking=True)
        # the captured step holds no gradient fill when it was captured after a step whose SGD pass left the buffer
        # zeroed (zero_grad() was free then): anything eager that accumulated since must be cleared here, not summed in
        if not getattr(self.engine, "_grad_is_zero", False):
            self.engine.store.zero_grad()
        self.graph.replay()
        self.engine._grad_is_zero = True     # the replayed SGD pass zeroed it again
        return self.loss


def smoke():
    """Tiny forward+backward+step of the flagship backbone on cuda:0 (called by __graft_entry__)."""
    import torch.nn as nn
    from nbdt.loss import SoftTreeSupLoss
    eng = WRNEngine(num_classes=10, blocks=10, width_factor=2, device="cuda:0", seed=0)
    crit = SoftTreeSupLoss(dataset="CIFAR10", criterion=nn.CrossEntropyLoss(), hierarchy="induced-wrn28_10_cifar10")
    g = torch.Generator().manual_seed(0)
    img = torch.randn(8, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (8,), generator=g).cuda()
    l0 = train_step(eng, crit, img, y, lr=0.05).item()
    for _ in range(5):
        l1 = train_step(eng, crit, img, y, lr=0.05).item()
    assert math.isfinite(l0) and math.isfinite(l1) and l1 < l0, (l0, l1)
    print(f"smoke ok: WRN-10-2 SoftTreeSupLoss train steps on cuda:0, loss {l0:.4f} -> {l1:.4f}")
