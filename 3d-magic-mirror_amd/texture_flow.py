"""Texture-flow sampling: host side of ``mm_texture_flow_forward / backward`` (csrc/mm_texflow.hip).

``sample_texture(img, texture_flow)`` is the tail of the reference's ``TextureEncoder.forward``
(/root/reference/network/model_res.py:597-612 with makeup == 0): bicubic ``grid_sample(align_corners=True)`` of the input image
at the decoder's flow, then the vertical mirror that makes the back of the texture equal to its front.  Device tensors only."""
import ctypes

import torch

from . import _native as N


class _TexFlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, flow):
        N.require_device(img, flow)
        dev = img.device
        img = img.detach().to(torch.float32).contiguous()
        flow = flow.detach().to(device=dev, dtype=torch.float32).contiguous()
        if img.dim() != 4 or flow.dim() != 4 or flow.shape[1] != 2 or flow.shape[0] != img.shape[0]:
            raise RuntimeError("sample_texture expects img (B,C,H,W) and texture_flow (B,2,Ho,Wo), got %s / %s" % (tuple(img.shape), tuple(flow.shape)))
        B, C, H, W = img.shape
        Ho, Wo = flow.shape[2:]
        out = torch.empty((B, C, 2 * Ho, Wo), device=dev, dtype=torch.float32)
        d = N.MMTexFlowDesc(B, C, H, W, Ho, Wo, N.ptr(img), N.ptr(flow), N.ptr(out))
        N.check(N.lib().mm_texture_flow_forward(ctypes.byref(d), N.current_stream(dev)), "mm_texture_flow_forward")
        ctx.save_for_backward(img, flow)
        return out

    @staticmethod
    def backward(ctx, g):
        img, flow = ctx.saved_tensors
        dev = img.device
        B, C, H, W = img.shape
        Ho, Wo = flow.shape[2:]
        g = g.to(torch.float32).contiguous()
        g_flow = torch.empty_like(flow)
        g_img = torch.empty_like(img) if ctx.needs_input_grad[0] else None
        d = N.MMTexFlowDesc(B, C, H, W, Ho, Wo, N.ptr(img), N.ptr(flow), None)
        gr = N.MMTexFlowGrads(N.ptr(g), N.ptr(g_flow), N.ptr(g_img))
        N.check(N.lib().mm_texture_flow_backward(ctypes.byref(d), ctypes.byref(gr), N.current_stream(dev)), "mm_texture_flow_backward")
        return g_img, g_flow


def sample_texture(img, texture_flow):
    """(B,C,2*Ho,Wo) texture from img (B,C,H,W) and the decoder's flow (B,2,Ho,Wo) -- model_res.py:597-612, makeup == 0."""
    return _TexFlowFn.apply(img, texture_flow)
