"""Host side of the attribute-reconstruction losses: autograd wrapper over
``mm_attribute_loss_forward / backward`` (csrc/mm_attloss.hip), the seven means of the reference's ``DiffRender.recon_att``
(/root/reference/networks.py:326-362).  Device tensors only."""
import ctypes

import torch

from . import _native as N

KEYS = ("azimuths", "elevations", "distances", "biases", "vertices", "textures", "lights")
AZIM, ELEV, DIST, BIAS, SHAPE, TEXTURE, LIGHT = range(7)


def _attributes(tensors):
    m = N.MMAttributes()
    for k, t in zip(KEYS, tensors):
        setattr(m, k, N.ptr(t))
    return m


class AttLossFn(torch.autograd.Function):
    """losses (7,) = [azim, elev, dist, bias, shape, texture, light] means; inputs: 7 pred tensors then 7 target tensors."""

    @staticmethod
    def forward(ctx, l1, *tensors):
        N.require_device(*tensors)
        dev = tensors[0].device
        ts = [t.detach().to(device=dev, dtype=torch.float32).contiguous() for t in tensors]
        pred, target = ts[:7], ts[7:]
        B = pred[0].reshape(-1).shape[0]
        V = pred[4].shape[1]
        Ht, Wt = pred[5].shape[2:]
        shapes = ((B,), (B,), (B,), (B, 2), (B, V, 3), (B, 3, Ht, Wt), (B, 9))
        for k, p, t, s in zip(KEYS, pred, target, shapes):
            if p.numel() != t.numel() or p.numel() != int(torch.tensor(s).prod()):
                raise RuntimeError("recon_att: %s must be %s in both attribute sets, got %s / %s" % (k, s, tuple(p.shape), tuple(t.shape)))
        losses = torch.empty(7, device=dev, dtype=torch.float32)
        d = N.MMAttLossDesc()
        d.B, d.V, d.Ht, d.Wt, d.l1 = B, V, Ht, Wt, int(bool(l1))
        d.pred, d.target = _attributes(pred), _attributes(target)
        d.losses = N.ptr(losses)
        ws = torch.zeros(N.lib().mm_attribute_loss_query_workspace(ctypes.byref(d)), device=dev, dtype=torch.uint8)    # zero-filled: ABI contract
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        N.check(N.lib().mm_attribute_loss_forward(ctypes.byref(d), N.current_stream(dev)), "mm_attribute_loss_forward")
        ctx.cfg = (B, V, Ht, Wt, int(bool(l1)))
        ctx.save_for_backward(*ts, ws)
        return losses

    @staticmethod
    def backward(ctx, g):
        *ts, ws = ctx.saved_tensors
        pred, target = ts[:7], ts[7:]
        dev = ws.device
        B, V, Ht, Wt, l1 = ctx.cfg
        d = N.MMAttLossDesc()
        d.B, d.V, d.Ht, d.Wt, d.l1 = B, V, Ht, Wt, l1
        d.pred, d.target = _attributes(pred), _attributes(target)
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        need = ctx.needs_input_grad[1:]
        grads = [torch.empty_like(t) if n else None for t, n in zip(ts, need)]
        if not any(need):
            return (None,) * 15
        w = g.detach().to(device=dev, dtype=torch.float32).contiguous()
        gr = N.MMAttLossGrads(N.ptr(w), _attributes(grads[:7]), _attributes(grads[7:]))
        N.check(N.lib().mm_attribute_loss_backward(ctypes.byref(d), ctypes.byref(gr), N.current_stream(dev)), "mm_attribute_loss_backward")
        return (None,) + tuple(grads)


def attribute_losses(pred_att, target_att, L1):
    pred = [pred_att[k] for k in KEYS]
    target = [target_att[k] for k in KEYS]
    shapes = [t.shape for t in pred]
    out = AttLossFn.apply(bool(L1), *[t.reshape(-1) if i < 3 else t for i, t in enumerate(pred)],
                          *[t.reshape(-1) if i < 3 else t for i, t in enumerate(target)])
    del shapes
    return out
