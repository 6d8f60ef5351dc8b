"""The three places of the library that still hold a FLOATING-POINT atomic (README "float atomics"; the render path, recon_data and every other
8(f) kernel have none), each pinned by its run-to-run spread on the device:

  mm_dibr.hip      dibr_rasterization backward with MORE THAN 8 feature channels: LDS float adds -- but only among the sixteen lanes of one
                   wave that sweep one face, in program order: the sums come out bit-identical run to run (asserted).
  mm_ops.hip       texture_mapping backward called WITHOUT the optional workspace (the C ABI allows it; ops.texture_mapping always passes one
                   and takes the 64-bit fixed-point path): global float atomics, order of arrival -- spread bounded, value against the
                   deterministic path.
  mm_texflow.hip   gradient to the SAMPLED IMAGE of the texture-flow sampler (the reference never asks for it: the image is an input): global
                   float atomics -- spread bounded.

These are the only translation units compiled with -munsafe-fp-atomics (build_native.FP_ATOMICS)."""
import ctypes
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, make_inputs
from parity_bar import rel_errors

pytestmark = pytest.mark.gpu
SHIM = os.path.join(ROOT, "3d-magic-mirror_amd", "shim")


def test_build_flags_name_exactly_the_files_with_float_atomics():
    bn = importlib.import_module("3d-magic-mirror_amd.build_native")
    assert "-munsafe-fp-atomics" not in bn.FLAGS
    with_flag = sorted(k for k, v in bn.SOURCES.items() if "-munsafe-fp-atomics" in v)
    assert with_flag == ["mm_dibr.hip", "mm_ops.hip", "mm_texflow.hip"]
    import re
    for src in bn.SOURCES:                                           # a float atomicAdd anywhere else would silently become a CAS loop: there is none
        text = open(os.path.join(bn.CSRC, src)).read()
        has = bool(re.search(r"atomicAdd\(\s*(&acc\[|g\.grad_textures|gi \+)", text))
        assert has == (src in with_flag), src


def test_dibr_backward_with_more_than_eight_channels_is_bitwise_reproducible(oracle):
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    import kaolin as kal
    B, S, D = 3, 64, 12
    inp, gt, proj = make_inputs("sphere", B, S, S, seed=9)
    T = oracle.camera(inp["distances"], inp["elevations"], inp["azimuths"], inp["biases"])
    fvc, fvi, fn = oracle.prepare_vertices(inp["vertices"], inp["faces"], T, proj)
    rng = np.random.default_rng(3)
    feats = rng.normal(size=(B, fvi.shape[1], 3, D)).astype(np.float32)
    g_i = torch.from_numpy(rng.normal(size=(B, S, S, D)).astype(np.float32)).cuda()
    g_s = torch.from_numpy(rng.normal(size=(B, S, S)).astype(np.float32)).cuda()
    runs = []
    for _ in range(3):
        fvi_t = torch.from_numpy(fvi).cuda().requires_grad_(True)
        ft = torch.from_numpy(feats).cuda().requires_grad_(True)
        interp, soft, fidx = kal.render.mesh.dibr_rasterization(S, S, torch.from_numpy(fvc[..., 2]).cuda(), fvi_t, ft, torch.from_numpy(fn[..., 2]).cuda())
        assert interp.shape == (B, S, S, D)
        ((interp * g_i).sum() + (soft * g_s).sum()).backward()
        runs.append((fvi_t.grad.clone(), ft.grad.clone()))
    assert float(runs[0][0].abs().max()) > 0 and float(runs[0][1].abs().max()) > 0
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1])


def test_texture_mapping_backward_without_a_workspace_has_a_bounded_spread():
    N = importlib.import_module("3d-magic-mirror_amd._native")
    ops = importlib.import_module("3d-magic-mirror_amd.ops")
    g = torch.Generator().manual_seed(11)
    B, n, C, Ht, Wt = 4, 128 * 128, 3, 256, 128
    uv = (torch.rand(B, n, 2, generator=g) * 0.25 + 0.3).cuda().contiguous()      # every point in a sixteenth of the texture: ~16 contributions per texel
    tex = torch.rand(B, C, Ht, Wt, generator=g).cuda().contiguous()
    dout = torch.randn(B, n, C, generator=g).cuda().contiguous()
    d = N.MMTexMapDesc(B, n, C, Ht, Wt, 1, N.ptr(uv), N.ptr(tex), None)
    dev = torch.device("cuda:0")

    def run(with_ws):
        gtex = torch.empty_like(tex)
        ws = torch.empty(N.lib().mm_texture_mapping_backward_query_workspace(ctypes.byref(d)), device=dev, dtype=torch.uint8) if with_ws else None
        gr = N.MMTexMapGrads(N.ptr(dout), None, N.ptr(gtex), N.ptr(ws), 0 if ws is None else ws.numel())
        N.check(N.lib().mm_texture_mapping_backward(ctypes.byref(d), ctypes.byref(gr), N.current_stream(dev)), "mm_texture_mapping_backward")
        torch.cuda.synchronize()
        return gtex
    fixed = [run(True) for _ in range(2)]
    assert torch.equal(fixed[0], fixed[1])                           # the product's path: bit-identical
    floats = [run(False) for _ in range(4)]
    spread = max(rel_errors(f, floats[0])[0] for f in floats[1:])
    off = max(rel_errors(f, fixed[0])[0] for f in floats)
    assert spread <= 5e-6, spread                                    # a few ulps of the largest sum: the order of ~16 float adds per texel
    assert off <= 5e-6, off
    print("texture_mapping backward without a workspace: run-to-run spread %.2e, distance from the fixed-point path %.2e (of max|grad|)" % (spread, off))


def test_texture_flow_image_gradient_has_a_bounded_spread():
    tf = importlib.import_module("3d-magic-mirror_amd.texture_flow")
    g = torch.Generator().manual_seed(13)
    B, C, H, W = 4, 3, 128, 128
    img = torch.rand(B, C, H, W, generator=g)
    flow = (torch.rand(B, 2, 64, 128, generator=g) * 1.6 - 0.8)
    gout = torch.randn(B, C, 128, 128, generator=g).cuda()
    grads = []
    for _ in range(4):
        im = img.cuda().requires_grad_(True); fl = flow.cuda().requires_grad_(True)
        tf.sample_texture(im, fl).backward(gout)
        torch.cuda.synchronize()
        grads.append((im.grad.clone(), fl.grad.clone()))
    for gi, gf in grads[1:]:
        assert torch.equal(gf, grads[0][1])                          # the flow gradient has no atomic: bit-identical
    spread = max(rel_errors(gi, grads[0][0])[0] for gi, _ in grads[1:])
    assert float(grads[0][0].abs().max()) > 0 and spread <= 5e-6, spread
    # against torch's own bicubic grid_sample (float64, CPU)
    im64 = img.double().requires_grad_(True)
    grid = flow.permute(0, 2, 3, 1).double()
    t = torch.nn.functional.grid_sample(im64, grid, mode="bicubic", align_corners=True)
    torch.cat([t, t.flip([2])], 2).backward(gout.cpu().double())
    assert rel_errors(grads[0][0], im64.grad.numpy())[0] <= 1e-4
    print("texture-flow image gradient: run-to-run spread %.2e of max|grad|" % spread)
