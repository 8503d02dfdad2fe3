"""nn.Module facade over the HIP execution engine.

The reference's backbones are ordinary ``nn.Module``s whose forward/backward run stock aten ops.
Here the module tree (same attribute names, same ``state_dict`` keys and logical shapes) is only a
VIEW: every ``nn.Parameter`` aliases a slice of the engine's flat fp32 buffer (conv weights are
channels-last views of the ``[cout][kh][kw][cin]`` master), ``.grad`` aliases the flat gradient
buffer, and ``forward`` / ``backward`` are one autograd node that runs the engine's hand-written
launch sequences.  ``torch.optim.SGD(net.parameters(), ...)``, ``net.state_dict()``,
``net.load_state_dict()``, ``net.train()/eval()`` therefore work exactly as with the reference
models, while ``engine.train_step`` is the faster all-native path (fused SGD, no autograd).
"""
import torch
import torch.nn as nn


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, module, *params):
        eng = module.engine
        module._sync_mirrors()
        z = eng.forward(img, training=module.training)
        # backward() reads the activations this forward left in the engine's buffers: remember which forward
        # that was, and whether it normalised with batch statistics (the only BatchNorm backward implemented)
        module._fwd_generation += 1
        ctx.module = module
        ctx.generation = module._fwd_generation
        ctx.batch_stats = bool(module.training)
        return z.clone()

    @staticmethod
    def backward(ctx, gz):
        module = ctx.module
        eng = module.engine
        params = module._param_list
        if not ctx.batch_stats:
            raise RuntimeError(
                "HIP backbone: backward through an eval-mode forward is not supported -- the fused inference "
                "path keeps no activations and BatchNorm backward is implemented for batch statistics only "
                "(call net.train(), or run the eval forward under torch.no_grad())")
        if ctx.generation != module._fwd_generation:
            raise RuntimeError(
                "HIP backbone: another forward ran between this output's forward and its backward; the engine "
                "keeps ONE set of activations, so call backward() before the next forward (or use "
                "torch.no_grad() for forwards that need no gradient)")
        if all(p.grad is None for p in params):
            eng.zero_grad()              # optimizer.zero_grad(set_to_none=True) semantics
        eng.backward(gz.contiguous().float())
        grads = eng.named_params("grad")
        for name, p in module._params_by_name.items():
            if p.grad is None:
                p.grad = grads[name]     # alias of the flat gradient buffer
        return (None, None) + (None,) * len(params)


class HipBackbone(nn.Module):
    def __init__(self, engine):
        super().__init__()
        object.__setattr__(self, "engine", engine)
        self._params_by_name = {}
        for name, view in engine.named_params("flat").items():
            p = nn.Parameter(view, requires_grad=True)
            self._params_by_name[name] = p
            self._attach(name, p, is_param=True)
        self._nbt = {}
        for name, buf in engine.named_buffers().items():
            if name.endswith("num_batches_tracked"):
                buf = buf.to(engine.device)
                self._nbt[name] = buf
            self._attach(name, buf, is_param=False)
        self._param_list = list(self._params_by_name.values())
        self._fwd_generation = 0
        self._seen_version = engine.store.flat._version

    def _attach(self, dotted, tensor, is_param):
        mod = self
        parts = dotted.split(".")
        for part in parts[:-1]:
            if part not in mod._modules:
                mod.add_module(part, nn.Module())
            mod = mod._modules[part]
        if is_param:
            mod.register_parameter(parts[-1], tensor)
        else:
            mod.register_buffer(parts[-1], tensor)

    def _sync_mirrors(self):
        """Parameters changed through PyTorch (optimizer.step, load_state_dict, manual edits) bump the
        flat buffer's version counter: refresh the bf16 mirror and the dgrad weight copies once."""
        v = self.engine.store.flat._version
        if v != self._seen_version:
            self.engine.store.refresh_bf16()
            self.engine.refresh_derived_weights()
            self._seen_version = self.engine.store.flat._version

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("HIP backbone: input is on CPU; there is no CPU fallback (move it to the GPU)")
        if torch.is_grad_enabled():
            return _BackboneFn.apply(x, self, *self._param_list)
        self._sync_mirrors()
        self._fwd_generation += 1        # a grad-free forward also overwrites the activation buffers
        return self.engine.forward(x, training=self.training).clone()

    def state_dict(self, *args, **kwargs):
        for b in self.engine.bns:
            t = self._nbt.get(b.name + ".num_batches_tracked")
            if t is not None:
                t.fill_(b.num_batches_tracked)
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        out = super().load_state_dict(state_dict, strict=strict, **kwargs)
        for b in self.engine.bns:
            t = self._nbt.get(b.name + ".num_batches_tracked")
            if t is not None:
                b.num_batches_tracked = int(t.item())
        self._seen_version = -1
        return out
