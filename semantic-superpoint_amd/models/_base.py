"""nn.Module front of the HIP engine: same ctor kwargs, forward signature, output dict and state_dict keys as
the reference networks; parameters and BatchNorm buffers are VIEWS of the engine's flat device buffers, so
torch optimisers, state_dict()/load_state_dict() and checkpoints work unchanged while every FLOP runs in
libssp_hip.so.  There is no CPU path: forward() raises for tensors that are not on a HIP device."""
import torch
import torch.nn as nn

from .. import lib as L
from .unet_parts import down, inconv


class _SspFunction(torch.autograd.Function):
    """Connects engine.forward / engine.backward to autograd (loss.backward(), Train_model_heatmap_all.py:407)."""

    @staticmethod
    def forward(ctx, net, x, slot, want_sem, *params):
        eng = net._engine
        want = ("semi", "desc", "sem") if want_sem else ("semi", "desc")
        out = eng.forward(x, slot=slot, train=net.training, want=want)
        ctx.net, ctx.slot, ctx.gen, ctx.want_sem = net, slot, net._slot_gen[slot], want_sem
        return tuple(out[k] for k in want)

    @staticmethod
    def backward(ctx, *grads):
        net, eng = ctx.net, ctx.net._engine
        if net._slot_gen[ctx.slot] != ctx.gen:
            raise RuntimeError("activations of this forward were overwritten: at most two forwards may be pending "
                               "before backward (the pair step needs exactly two)")
        g = [None if t is None else t.contiguous() for t in grads]
        eng.zero_grad()
        eng.backward(ctx.slot, g[0], g[1], g[2] if ctx.want_sem else None)
        flat = eng.grads
        outs = [flat[off:off + p.numel()].view_as(p).clone() for p, off in zip(net._flat_params, net._flat_offsets)]
        return (None, None, None, None) + tuple(outs)


class SspNetBase(nn.Module):
    ARCH = None

    def _build(self, n_classes=None):
        c1, c2, c3, c4, c5, d1, det_h = 64, 64, 128, 128, 256, 256, 65
        self.inc = inconv(1, c1)
        self.down1 = down(c1, c2)
        self.down2 = down(c2, c3)
        self.down3 = down(c3, c4)
        self.relu = nn.ReLU(inplace=True)
        self.convPa = nn.Conv2d(c4, c5, kernel_size=3, stride=1, padding=1)
        self.bnPa = nn.BatchNorm2d(c5)
        self.convPb = nn.Conv2d(c5, det_h, kernel_size=1, stride=1, padding=0)
        self.bnPb = nn.BatchNorm2d(det_h)
        self.convDa = nn.Conv2d(c4, c5, kernel_size=3, stride=1, padding=1)
        self.bnDa = nn.BatchNorm2d(c5)
        self.convDb = nn.Conv2d(c5, d1, kernel_size=1, stride=1, padding=0)
        self.bnDb = nn.BatchNorm2d(d1)
        if n_classes is not None:
            self.convDS = nn.Conv2d(c4, c5, kernel_size=3, stride=1, padding=1)
            self.bnS1 = nn.BatchNorm2d(c5)
            self.convSout = nn.Conv2d(c5, n_classes, kernel_size=1, stride=1, padding=0)
        self.n_classes = n_classes if n_classes is not None else 133
        self.output = None
        self._engine = None
        self._slot_gen = [0, 0]
        self._next_slot = 0

    # ---- engine management ----
    def _apply(self, fn, *a, **k):
        """.to()/.cuda()/.float() replace the parameter tensors: detach from the engine first, rebind lazily."""
        self._release_engine()
        return super()._apply(fn, *a, **k)

    def _release_engine(self):
        if getattr(self, "_engine", None) is not None:
            with torch.no_grad():
                for _, t in list(self.named_parameters()) + list(self.named_buffers()):
                    t.data = t.data.clone()  # own storage again
            self._engine = None

    def engine(self, n=None, h=None, w=None, device=None):
        """The bound Engine (created / re-created when the shape grows or the device changes)."""
        e = self._engine
        if e is not None and (n is None or (n <= e.max_batch and (h, w) == (e.height, e.width) and
                                            torch.device(device) == e.device)):
            return e
        assert n is not None, "engine not created yet: run a forward first"
        sd = {k: v.detach().clone() for k, v in self.state_dict().items()}
        # Shape-independent training state of the old engine survives the re-creation (a larger validation batch or a
        # differently sized image must not reset Adam, MultiTaskLoss.eta or pending micro-batch gradients mid-training)
        old = self._engine
        carry = None
        if old is not None and old.grads is not None and torch.device(device) == old.device:
            carry = (old.eta.clone(), old.adam_m.clone(), old.adam_v.clone(), old.adam_t, old.grads.clone())
        if old is not None:
            n = max(n, old.max_batch)  # never shrink: the next training batch would re-create it again
        self._release_engine()
        del old
        e = L.Engine(self.ARCH, n, h, w, device, n_classes=self.n_classes, **getattr(self, "_engine_kwargs", {}))
        e.load_state_dict(sd)
        if carry is not None:
            e.params[e.n_params:] = carry[0]
            e.adam_m.copy_(carry[1]); e.adam_v.copy_(carry[2]); e.adam_t = carry[3]
            e.grads.copy_(carry[4])
        own = dict(self.named_parameters())
        bufs = dict(self.named_buffers())
        self._flat_params, self._flat_offsets = [], []
        for key, shape, off in e.layout:
            p = own[key]
            p.data = e.params[off:off + p.numel()].view(shape)
            self._flat_params.append(p)
            self._flat_offsets.append(off)
        for i, (bn, c, off) in enumerate(e.bns):
            bufs[bn + ".running_mean"].data = e.bn_running[off:off + c]
            bufs[bn + ".running_var"].data = e.bn_running[e.n_bn_ch + off:e.n_bn_ch + off + c]
            bufs[bn + ".num_batches_tracked"].data = e.nbt[i]
        self._engine = e
        self._slot_gen = [0, 0]
        self._next_slot = 0
        return e

    def _run(self, x, want_sem):
        if not x.is_cuda:
            raise RuntimeError("%s runs on MI355X only (input on %s): there is no CPU fallback" % (self.ARCH, x.device))
        x = x.contiguous().float()
        n, c, h, w = x.shape
        assert c == 1 and h % 8 == 0 and w % 8 == 0, "input must be [N,1,H,W] with H, W multiples of 8"
        self.engine(n, h, w, x.device)
        slot = self._next_slot
        self._next_slot ^= 1
        self._slot_gen[slot] += 1
        want = ("semi", "desc", "sem") if want_sem else ("semi", "desc")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._flat_params):
            outs = _SspFunction.apply(self, x, slot, want_sem, *self._flat_params)
        else:
            o = self._engine.forward(x, slot=slot, train=self.training, want=want)
            outs = tuple(o[k] for k in want)
        return dict(zip(want, outs))
