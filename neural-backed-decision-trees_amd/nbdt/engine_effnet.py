"""EfficientNet-B0 execution engine (SURVEY.md row A4 / BASELINE config 5).

Restates pytorchcv's ``efficientnet_b0`` (the model the reference re-exports at
nbdt/models/__init__.py:3 and names in README.md:141-151) as a fixed launch sequence over the HIP
kernels: 1x1 expand / project convolutions are the implicit-GEMM kernels (csrc/conv_dma.hip) with the
following BatchNorm's statistics fused into their epilogue; depthwise convolutions, BatchNorm + swish,
squeeze-and-excitation and dropout are the streaming kernels of csrc/effnet.hip.  State-dict names and
logical shapes follow pytorchcv (``features.stageS.unitU.{conv1,conv2,se,conv3}``,
``features.final_block``, ``output.fc``).

Per MBConv unit the forward keeps only raw conv outputs + the tensors the next conv must read
(e_raw, e_act, d_raw, d_se, p_raw, out); the swish / SE-scaled activations needed by the backward are
recomputed from the raw outputs inside the backward kernels.
"""
import math

import torch

from nbdt import ops
from nbdt.engine import _Engine, _pad32, side_stream

# (stage stride, [(out_channels, kernel, expansion), ...]) -- pytorchcv get_efficientnet(version="b0")
B0_STAGES = [
    (1, [(16, 3, 1)]),
    (2, [(24, 3, 6), (24, 3, 6)]),
    (2, [(40, 5, 6), (40, 5, 6)]),
    (2, [(80, 3, 6)] * 3 + [(112, 5, 6)] * 3),
    (2, [(192, 5, 6)] * 4 + [(320, 3, 6)]),
]
ACT = ops.ACT_SWISH


class DepthwiseConv:
    """groups=C Conv2d(k, stride, padding=k//2), bias-free; master weights fp32 [k*k][Cpad]."""

    def __init__(self, store, name, c_real, k, stride, gen):
        self.store, self.name, self.c_real, self.k, self.stride = store, name, c_real, k, stride
        self.C = _pad32(c_real)
        bound = math.sqrt(2.0) * math.sqrt(3.0 / (k * k))   # kaiming_uniform_(a=0), fan_in = k*k

        def init(v):
            v.zero_()
            v[:, :c_real].uniform_(-bound, bound, generator=gen)

        store.add(name, (k * k, self.C), init)

    def logical(self, buf):
        v = self.store._view(buf, self.name).view(self.k, self.k, self.C)
        return v[:, :, :self.c_real].permute(2, 0, 1).unsqueeze(1)      # [C, 1, k, k]

    def forward(self, x, y, bn_scratch=None):
        ops.dwconv_fwd(x, self.store.p(self.name), y, self.k, self.stride, bn_scratch)

    def backward_data(self, gy, gx):
        ops.dwconv_bwd_data(gy, self.store.p(self.name), gx, self.k, self.stride)

    def backward_weight(self, x, gy, also=None):
        """also: a further launch that only feeds the optimizer (the SE block's parameter gradients), issued behind the
        weight gradient on the same stream -- one cross-stream dependency for both."""
        side = getattr(self, "side_stream", None)
        if side is None:
            ops.dwconv_bwd_weight(x, gy, self.store.g(self.name), self.k, self.stride)
            if also is not None:
                also()
            return
        side.wait_stream(torch.cuda.current_stream(x.device))     # second stream: see WRNEngine
        with torch.cuda.stream(side):
            ops.dwconv_bwd_weight(x, gy, self.store.g(self.name), self.k, self.stride)
            if also is not None:
                also()


class SqueezeExcite:
    """pytorchcv SEBlock: two biased 1x1 convs on the pooled vector (swish, sigmoid)."""

    def __init__(self, store, name, c_real, mid, gen):
        self.store, self.name, self.c_real, self.mid = store, name, c_real, mid
        b1 = math.sqrt(2.0) * math.sqrt(3.0 / c_real)
        b2 = math.sqrt(2.0) * math.sqrt(3.0 / mid)
        store.add(name + ".conv1.weight", (mid, c_real), lambda v: v.uniform_(-b1, b1, generator=gen))
        store.add(name + ".conv1.bias", (mid,), lambda v: v.zero_())
        store.add(name + ".conv2.weight", (c_real, mid), lambda v: v.uniform_(-b2, b2, generator=gen))
        store.add(name + ".conv2.bias", (c_real,), lambda v: v.zero_())

    def views(self, buf):
        s, n = self.store, self.name
        return {
            n + ".conv1.weight": s._view(buf, n + ".conv1.weight").view(self.mid, self.c_real, 1, 1),
            n + ".conv1.bias": s._view(buf, n + ".conv1.bias"),
            n + ".conv2.weight": s._view(buf, n + ".conv2.weight").view(self.c_real, self.mid, 1, 1),
            n + ".conv2.bias": s._view(buf, n + ".conv2.bias"),
        }

    def forward(self, pooled, pre1, gate):
        p = self.store.p
        ops.se_gate_fwd(pooled, p(self.name + ".conv1.weight"), p(self.name + ".conv1.bias"),
                        p(self.name + ".conv2.weight"), p(self.name + ".conv2.bias"), pre1, gate, self.c_real)

    def backward(self, dgate, gate, pre1, pooled, dpre2, dpre1, gpool, defer_params=False):
        """defer_params: the data part only; returns the launch of the parameter gradients for the caller to place (they
        read dpre2 / dpre1 / pre1 / pooled and feed nothing but the optimizer)."""
        p, g = self.store.p, self.store.g
        grads = (g(self.name + ".conv1.weight"), g(self.name + ".conv1.bias"), g(self.name + ".conv2.weight"),
                 g(self.name + ".conv2.bias"))
        ops.se_gate_bwd(dgate, gate, pre1, pooled, p(self.name + ".conv1.weight"), p(self.name + ".conv2.weight"),
                        dpre2, dpre1, gpool, *((None,) * 4 if defer_params else grads), self.c_real)
        if defer_params:
            return lambda: ops.se_param_grad(dpre2, dpre1, pre1, pooled, *grads, self.c_real)
        return None


class EfficientNetEngine(_Engine):
    def __init__(self, num_classes=1000, dropout_rate=0.2, stages=B0_STAGES, init_channels=32,
                 final_channels=1280, device="cuda", seed=0):
        super().__init__(device, seed)
        self.num_classes, self.dropout_rate = num_classes, float(dropout_rate)
        gen = self.gen
        self.stem_c = init_channels
        b0 = math.sqrt(2.0) * math.sqrt(3.0 / 27)
        self.store.add("features.init_block.conv.conv.weight", (init_channels, 3, 3, 3),
                       lambda v: v.uniform_(-b0, b0, generator=gen))
        self.bn0 = self.bn("features.init_block.conv.bn", init_channels)
        self.units, self.dws, self.ses = [], [], []
        cin = init_channels
        for i, (stage_stride, specs) in enumerate(stages):
            for j, (cout, k, exp) in enumerate(specs):
                stride = stage_stride if j == 0 else 1
                pre = f"features.stage{i + 1}.unit{j + 1}."
                mid = cin * exp
                u = {"cin": cin, "cout": cout, "mid": mid, "k": k, "stride": stride, "exp": exp,
                     "key": f"s{i + 1}u{j + 1}", "stage": i + 1, "residual": cin == cout and stride == 1}
                if i == 0:   # EffiDwsConvUnit: depthwise -> SE -> pointwise
                    u["conv1"] = u["bn1"] = None
                    u["dw"] = DepthwiseConv(self.store, pre + "dw_conv.conv.weight", mid, k, stride, gen)
                    u["bn2"] = self.bn(pre + "dw_conv.bn", mid)
                    u["se"] = SqueezeExcite(self.store, pre + "se", mid, mid // 4, gen)
                    u["conv3"] = self.conv(pre + "pw_conv.conv.weight", mid, cout, 1, 1)
                    u["bn3"] = self.bn(pre + "pw_conv.bn", cout)
                else:        # EffiInvResUnit: expand -> depthwise -> SE -> project
                    u["conv1"] = self.conv(pre + "conv1.conv.weight", cin, mid, 1, 1)
                    u["bn1"] = self.bn(pre + "conv1.bn", mid)
                    u["dw"] = DepthwiseConv(self.store, pre + "conv2.conv.weight", mid, k, stride, gen)
                    u["bn2"] = self.bn(pre + "conv2.bn", mid)
                    u["se"] = SqueezeExcite(self.store, pre + "se", mid, mid // (exp * 4), gen)
                    u["conv3"] = self.conv(pre + "conv3.conv.weight", mid, cout, 1, 1)
                    u["bn3"] = self.bn(pre + "conv3.bn", cout)
                self.dws.append(u["dw"])
                self.ses.append(u["se"])
                self.units.append(u)
                cin = cout
        self.final_conv = self.conv("features.final_block.conv.weight", cin, final_channels, 1, 1)
        self.final_bn = self.bn("features.final_block.bn", final_channels)
        self.feat_c = final_channels
        kb = 1.0 / math.sqrt(final_channels)
        self.store.add("output.fc.weight", (num_classes, final_channels), lambda v: v.uniform_(-kb, kb, generator=gen))
        self.store.add("output.fc.bias", (num_classes,), lambda v: v.uniform_(-kb, kb, generator=gen))
        self.finalize()
        self._step = 0
        self.dropout_seed = seed
        self.defer_se_params = True   # SE parameter gradients behind the depthwise weight gradient on the second stream
        self.fuse_se_bwd = True   # SE backward: dL/dgate and bn2's backward sums from ONE pass over (gd, d_raw)
        self._side = side_stream(self.device)     # weight gradients on the process's second stream (see WRNEngine)
        for c in self.convs + self.dws:
            c.side_stream = self._side

    # ------------------------------------------------------------------ reference-named views
    def extra_param_views(self, buf):
        out = {
            "features.init_block.conv.conv.weight":
                self.store._view(buf, "features.init_block.conv.conv.weight").permute(0, 3, 1, 2),
            "output.fc.weight": self.store._view(buf, "output.fc.weight"),
            "output.fc.bias": self.store._view(buf, "output.fc.bias"),
        }
        for d in self.dws:
            out[d.name] = d.logical(buf)
        for s in self.ses:
            out.update(s.views(buf))
        return out

    def grad_buckets(self):
        """Flat-gradient ranges in the order backward completes them: [stage5..classifier], [stage3..4],
        [stem..stage2]."""
        ent = self.store.entries
        s3 = ent["features.stage3.unit1.conv1.conv.weight"][0]
        s5 = ent["features.stage5.unit1.conv1.conv.weight"][0]
        return [(s5, self.store.grad.numel()), (s3, s5), (0, s3)]

    def _vec(self, key, B, n):
        return self._tensor(key, (B, n))

    def algorithmic_bytes(self, B, size):
        """HBM bytes one training step of THIS schedule has to move if every pass reads each of its input tensors once and
        writes each output once (bf16 activations / gradients, padded channel counts; weights, statistics and the [B, C]
        vectors of the SE branch are noise and left out).  What bench.py prices EfficientNet-B0 against: round 4 used
        2 x the resident buffers (12.6 GB at 128 x 224 x 224), which ignores that backward re-reads every stored
        activation and that several passes read two tensors -- the PMC counters said 36 GB.  Per unit, with X / E / D / O
        the unit's input, expanded (before the depthwise conv), depthwise-output and output tensors:
          forward   expand conv X + E | BatchNorm + swish 2E | depthwise E + D | squeeze D | scale 2D | project D + O |
                    BatchNorm (+ skip) 2O (+ X)
          backward  bn3 5O | project weight gradient D + O, data gradient O + D | dL/dgate + bn2's sums 2D (one pass since round
                    6; fuse_se_bwd off: 2D + 2D) | bn2 + SE elementwise 3D | depthwise
                    weight gradient E + D, data gradient + bn1 sums D + 2E | bn1 elementwise 3E | expand weight gradient
                    X + E, data gradient E + X (+ X when it accumulates onto the skip gradient)"""
        h = w = size // 2
        e = lambda hh, ww, c: 2 * B * hh * ww * _pad32(c)       # bytes of one bf16 tensor
        stem = e(h, w, self.stem_c)
        total = 12 * B * size * size + 2 * stem + 2 * stem      # image in, stem conv out, BatchNorm + swish (read + write)
        total += 5 * stem + 12 * B * size * size + stem         # backward of that BatchNorm, stem weight gradient
        se = 2 if self.fuse_se_bwd else 4                      # tensor reads of the reduction pass(es) over (gd, d_raw)
        for u in self.units:
            s = u["stride"]
            ho, wo = h // s, w // s
            X, O = e(h, w, u["cin"]), e(ho, wo, u["cout"])
            E, D = e(h, w, u["mid"]), e(ho, wo, u["mid"])
            res = X if u["residual"] else 0
            if u["conv1"] is not None:
                fwd = (X + E) + 2 * E + (E + D) + D + 2 * D + (D + O) + 2 * O + res
                bwd = 5 * O + (D + O) + (O + D) + se * D + 3 * D + (E + D) + (D + 2 * E) + 3 * E + (X + E) + (E + X) + res
            else:       # stage 1: depthwise on the unit's input, no expand conv
                fwd = (X + D) + D + 2 * D + (D + O) + 2 * O + res
                bwd = 5 * O + (D + O) + (O + D) + se * D + 3 * D + (X + D) + (D + X)
            total += fwd + bwd
            h, w = ho, wo
        F, L = e(h, w, self.feat_c), e(h, w, self.units[-1]["cout"])
        total += (L + F) + F                                    # final conv, BatchNorm + swish + pool (reads only)
        total += 4 * F + F + (L + F) + (F + L)                  # its backward (pool form: no gradient tensor read), conv grads
        return int(total)

    # ------------------------------------------------------------------ forward
    def forward(self, img, training=None):
        training = self.training if training is None else training
        if img.dtype != torch.float32 or not img.is_contiguous():
            img = img.float().contiguous()
        B, _, H, W = img.shape
        if H % 32 or W % 32:
            raise ValueError("EfficientNet input size must be a multiple of 32")
        self._img, self._B = img, B
        fuse = training and self.fuse_stats
        h, w = H // 2, W // 2
        c0 = _pad32(self.stem_c)
        t0 = self.buf("t0", B, h, w, c0)
        x = self.buf("a0", B, h, w, c0)
        ops.stem_conv(img, self.store.p("features.init_block.conv.conv.weight"), t0, self.stem_c, stride=2)
        self.bn0.stats(t0, training)
        bn = self.bn0
        ops.bn_act_apply(t0, bn.mean, bn.rstd, bn.gamma, bn.beta, x, act=ACT)
        for u in self.units:
            k, s = u["key"], u["stride"]
            cin, mid, cout = _pad32(u["cin"]), _pad32(u["mid"]), _pad32(u["cout"])
            ho, wo = h // s, w // s
            if u["conv1"] is not None:
                e_raw = self.buf(k + ".e_raw", B, h, w, mid)
                e_act = self.buf(k + ".e_act", B, h, w, mid)
                if not training and self.fuse_eval:   # expand conv + BN + swish in one launch
                    u["conv1"].forward_affine(x, e_act, u["bn1"], act=2)
                else:
                    u["conv1"].forward(x, e_raw, bn_scratch=self.partials(e_raw) if fuse else None)
                    bn = u["bn1"]
                    bn.stats(e_raw, training, fused=fuse)
                    ops.bn_act_apply(e_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, e_act, act=ACT)
            else:
                e_act = x
            d_raw = self.buf(k + ".d_raw", B, ho, wo, mid)
            d_se = self.buf(k + ".d_se", B, ho, wo, mid)
            bn = u["bn2"]
            if fuse:   # the depthwise kernel leaves sum / sum-of-squares in the BN slots: fold only
                u["dw"].forward(e_act, d_raw, bn_scratch=self.scratch(bn.C))
                ops.bn_stats(d_raw, self.scratch(bn.C), bn.mean, bn.rstd, bn.running_mean, bn.running_var,
                             slots_filled=True)
                bn.num_batches_tracked += 1
            else:
                u["dw"].forward(e_act, d_raw)
                bn.stats(d_raw, training)
            pooled, gate = self._vec(k + ".pooled", B, mid), self._vec(k + ".gate", B, mid)
            pre1 = self._vec(k + ".pre1", B, u["se"].mid)
            ops.bn_act_pool(d_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, pooled, act=ACT)
            u["se"].forward(pooled, pre1, gate)
            ops.bn_act_apply(d_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, d_se, act=ACT, gate=gate)
            p_raw = self.buf(k + ".p_raw", B, ho, wo, cout)
            out = self.buf(k + ".out", B, ho, wo, cout)
            if not training and self.fuse_eval:       # project conv + BN (+ skip) in one launch
                u["conv3"].forward_affine(d_se, out, u["bn3"], act=0, residual=x if u["residual"] else None)
            else:
                u["conv3"].forward(d_se, p_raw, bn_scratch=self.partials(p_raw) if fuse else None)
                bn = u["bn3"]
                bn.stats(p_raw, training, fused=fuse)
                ops.bn_act_apply(p_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, out, act=ops.ACT_NONE,
                                 residual=x if u["residual"] else None)
            u["x_in"], u["hw_in"] = x, (h, w)
            x, h, w = out, ho, wo
        self._x_last, self._hw = x, (h, w)
        f_raw = self.buf("f_raw", B, h, w, self.feat_c)
        self.final_conv.forward(x, f_raw, bn_scratch=self.partials(f_raw) if fuse else None)
        bn = self.final_bn
        bn.stats(f_raw, training, fused=fuse)
        self._pooled = self._vec("pooled", B, self.feat_c)
        ops.bn_act_pool(f_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, self._pooled, act=ACT)
        feat = self._pooled
        self._dropped = training and self.dropout_rate > 0.0
        if self._dropped:
            if ("mask", B) not in self._bufs:
                self._bufs[("mask", B)] = torch.empty((B, self.feat_c), dtype=torch.uint8, device=self.device)
            self._mask = self._bufs[("mask", B)]
            feat = self._vec("dropped", B, self.feat_c)
            self._step += 1
            ops.dropout_fwd(self._pooled, self.dropout_rate, self.dropout_seed * 1000003 + self._step, self._mask,
                            feat)
        self._feat = feat
        z = self._vec("z", B, self.num_classes)
        ops.linear_fwd(feat, self.store.p("output.fc.weight"), self.store.p("output.fc.bias"), z)
        return z

    # ------------------------------------------------------------------ backward
    def backward(self, gz, comm=None):
        self._grad_is_zero = False   # this call accumulates into the gradient buffer
        B = self._B
        self.join_side_stream()      # dgrad weight copies (built on the second stream after the last update)
        st = self.store
        buckets = self.grad_buckets() if comm is not None else None
        gz = gz.contiguous()
        gfeat = self._vec("gfeat", B, self.feat_c)
        ops.linear_bwd(self._feat, st.p("output.fc.weight"), gz, gfeat, st.g("output.fc.weight"),
                       st.g("output.fc.bias"))
        if self._dropped:
            gpooled = self._vec("gpooled", B, self.feat_c)
            ops.dropout_bwd(gfeat, self.dropout_rate, self._mask, gpooled)
        else:
            gpooled = gfeat
        h, w = self._hw
        f_raw = self.buf("f_raw", B, h, w, self.feat_c)
        gf = self.buf("g_f", B, h, w, self.feat_c)
        bn = self.final_bn
        ops.bn_act_bwd(None, f_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, self.scratch(bn.C), bn.dsum,
                       st.g(bn.name + ".weight"), st.g(bn.name + ".bias"), gf, act=ACT, gpool=gpooled)
        self.final_conv.backward_weight(self._x_last, gf)
        g = self.buf(f"g_{self._x_last.shape[3]}_{h}", B, h, w, self._x_last.shape[3])
        self.final_conv.backward_data(gf, g)
        two_streams = self._side is not None and self._overlap
        n_unit, side_mark = 0, None
        if two_streams:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
        for u in reversed(self.units):
            # Gradient buffers are shared between units and the weight gradients run on the second stream: as in
            # WRNEngine.backward the main stream waits for what the side stream had been given ONE UNIT AGO, and the
            # three buffers a weight gradient reads (gp, gd, ge) alternate between consecutive units, so that a unit
            # never overwrites what the previous unit's weight gradients may still be reading (checked bit for bit
            # against one stream with private buffers: tests/test_effnet_gpu.py).
            if two_streams:
                main = torch.cuda.current_stream(self.device)
                if side_mark is not None:
                    main.wait_event(side_mark)
                side_mark = torch.cuda.Event()
                side_mark.record(self._side)
            par = n_unit & 1
            n_unit += 1
            k, s = u["key"], u["stride"]
            cin, mid, cout = _pad32(u["cin"]), _pad32(u["mid"]), _pad32(u["cout"])
            ho, wo = h, w
            hi, wi = u["hw_in"]
            tag = ("@" + k) if self.debug_keep else ""
            p_raw = self.buf(k + ".p_raw", B, ho, wo, cout)
            d_raw = self.buf(k + ".d_raw", B, ho, wo, mid)
            d_se = self.buf(k + ".d_se", B, ho, wo, mid)
            gp = self.buf(f"gp_{cout}_{ho}_{par}{tag}", B, ho, wo, cout)
            gd = self.buf(f"gd_{mid}_{ho}_{par}{tag}", B, ho, wo, mid)
            bn = u["bn3"]
            ops.bn_act_bwd(g, p_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, self.scratch(bn.C), bn.dsum,
                           st.g(bn.name + ".weight"), st.g(bn.name + ".bias"), gp, act=ops.ACT_NONE)
            u["conv3"].backward_weight(d_se, gp)
            u["conv3"].backward_data(gp, gd)
            # squeeze-and-excitation + BatchNorm/swish of the depthwise output (gd rewritten in place)
            bn = u["bn2"]
            gpool = self._vec(f"gpool{tag}", B, mid)
            gate = self._vec(k + ".gate", B, mid)
            one_pass = self.fuse_se_bwd and gd.dtype == torch.bfloat16 and not ops.is_deterministic()
            if one_pass:
                # dL/dgate AND what bn2's backward sums are linear in, in one pass over (gd, d_raw): the reduction pass
                # of bn_act_bwd (a second read of both tensors) becomes a [B, C]-sized fold once gpool exists
                sums = self._zeroed(f"se_sums{tag}", (5, B, mid))     # zero on entry, re-zeroed by its last reader
                dirty = self.__dict__.setdefault("_se_sums_dirty", set())
                if id(sums) in dirty:        # a backward that stopped between the two calls (an exception, an interrupt) left
                    sums.zero_()             # its sums behind: never accumulate on top of them
                dirty.add(id(sums))
                ops.bn_act_se_sums(gd, d_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, sums, act=ACT)
                dgate = sums[0]
            else:
                dgate = self._vec(f"dgate{tag}", B, mid)
                ops.bn_act_pool(d_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, dgate, act=ACT, mul=gd, scale=1.0)
            # the SE block's parameter gradients go behind the depthwise weight gradient on the second stream (their
            # workspaces alternate between consecutive units like the gradient buffers)
            se_params = u["se"].backward(dgate, gate, self._vec(k + ".pre1", B, u["se"].mid),
                                         self._vec(k + ".pooled", B, mid), self._vec(f"dpre2_{par}{tag}", B, mid),
                                         self._vec(f"dpre1_{par}{tag}", B, u["se"].mid), gpool,
                                         defer_params=self.defer_se_params)
            if one_pass:
                ops.bn_act_se_bwd_apply(gd, gate, gpool, sums, d_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, bn.dsum,
                                        st.g(bn.name + ".weight"), st.g(bn.name + ".bias"), gd, act=ACT)
                dirty.discard(id(sums))
            else:
                ops.bn_act_bwd(gd, d_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, self.scratch(bn.C), bn.dsum,
                               st.g(bn.name + ".weight"), st.g(bn.name + ".bias"), gd, act=ACT, gate=gate, gpool=gpool)
            x_in = u["x_in"]
            if u["conv1"] is not None:
                e_raw = self.buf(k + ".e_raw", B, hi, wi, mid)
                e_act = self.buf(k + ".e_act", B, hi, wi, mid)
                ge = self.buf(f"ge_{mid}_{hi}_{par}{tag}", B, hi, wi, mid)
                u["dw"].backward_weight(e_act, gd, also=se_params)
                bn = u["bn1"]
                if u["dw"].stride == 1 and self.fuse_dw_bn_bwd:
                    # the depthwise data gradient's epilogue leaves bn1's backward sums in the slots: no reduction
                    # pass over ge and e_raw (5.5 % of a step), e_raw is read once there
                    ops.dwconv_bwd_data_bn(gd, st.p(u["dw"].name), ge, u["dw"].k, e_raw, bn.mean, bn.rstd, bn.gamma,
                                           bn.beta, self.scratch(bn.C))
                    ops.bn_act_bwd_apply(ge, e_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, self.scratch(bn.C), bn.dsum,
                                         st.g(bn.name + ".weight"), st.g(bn.name + ".bias"), ge, act=ACT)
                else:
                    u["dw"].backward_data(gd, ge)
                    ops.bn_act_bwd(ge, e_raw, bn.mean, bn.rstd, bn.gamma, bn.beta, self.scratch(bn.C), bn.dsum,
                                   st.g(bn.name + ".weight"), st.g(bn.name + ".bias"), ge, act=ACT)
                u["conv1"].backward_weight(x_in, ge)
                if u["residual"]:
                    # out = bn3(...) + x_in: the unit-output gradient g is also the skip gradient ->
                    # accumulate the expand conv's data gradient into it in place
                    u["conv1"].backward_data(ge, g, accumulate=True)
                    g_in = g
                else:
                    g_in = self.buf(f"g_{cin}_{hi}{tag}", B, hi, wi, cin)
                    u["conv1"].backward_data(ge, g_in)
            else:
                g_in = self.buf(f"g_{cin}_{hi}{tag}", B, hi, wi, cin)
                u["dw"].backward_weight(x_in, gd, also=se_params)
                u["dw"].backward_data(gd, g_in)
            u["dbg"] = {"g_out": g, "g_in": g_in, "gp": gp, "gd": gd}
            g, h, w = g_in, hi, wi
            if comm is not None and k in ("s5u1", "s3u1"):
                self.join_side_stream()
                comm.reduce_range(st.grad, *buckets[0 if k == "s5u1" else 1])
                self._reserve_for(comm)
        bn = self.bn0
        t0 = self.buf("t0", B, h, w, _pad32(self.stem_c))
        ops.bn_act_bwd(g, t0, bn.mean, bn.rstd, bn.gamma, bn.beta, self.scratch(bn.C), bn.dsum,
                       st.g(bn.name + ".weight"), st.g(bn.name + ".bias"), g, act=ACT)
        ops.stem_wgrad(self._img, g, st.g("features.init_block.conv.conv.weight"), self.stem_c, stride=2)
        self.join_side_stream()
        if comm is not None:
            comm.reduce_range(st.grad, *buckets[2])
            comm.finish(st.grad)
            self._reserve_for(None)
