"""Debug of a fuzz_parity failure: the K4 (soft-mask) gradient per face, un-fused HIP operator vs oracle, for fuzz case (seed, index)."""
import sys, importlib, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "3d-magic-mirror_amd", "shim"))
import oracle
import kaolin as kal
pkg = importlib.import_module("3d-magic-mirror_amd")
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    name = rng.choice(["sphere", "smpl_uv_642", "ellipsoid", "sphere2", "smpl_uv"], p=[0.25, 0.3, 0.15, 0.15, 0.15])
    big = name in ("sphere2", "smpl_uv")
    S = int(rng.choice([24, 32, 40, 50, 64, 72, 96, 128] if not big else [24, 32, 48, 64, 80]))
    ratio = int(rng.choice([1, 1, 2])); B = int(rng.integers(1, 5 if big else 9)); no_mask = bool(rng.integers(0, 2))
    knum = int(rng.choice([30, 30, 30, 5, 70])); boxlen = float(rng.choice([0.02, 0.02, 0.05, 0.15])); sigmainv = float(rng.choice([7000.0, 7000.0, 900.0, 200.0]))
    mode = rng.choice(["default", "far", "near", "mixed"], p=[0.4, 0.25, 0.15, 0.2])
    dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio)
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=int(rng.integers(0, 1 << 30)))
    if mode == "far": att["distances"] = torch.full_like(att["distances"], float(rng.uniform(8.0, 30.0)))
    elif mode == "near": att["distances"] = torch.full_like(att["distances"], float(rng.uniform(1.5, 1.9)))
    elif mode == "mixed": att["distances"] = torch.from_numpy(rng.uniform(1.6, 25.0, size=B).astype(np.float32))
print(name, B, H, W, "knum", knum, "boxlen", boxlen, "sigmainv", sigmainv, mode)
dev = torch.device("cuda:0")
faces = dr.faces.numpy().astype(np.int32); proj = dr.cam_proj.numpy().reshape(3)
T = oracle.camera(att["distances"].numpy(), att["elevations"].numpy(), att["azimuths"].numpy(), att["biases"].numpy())
fvc, fvi, fn = oracle.prepare_vertices(att["vertices"].numpy(), faces, T, proj)
F = faces.shape[0]
ones = np.ones((B, F, 3, 1), np.float32)
valid = (fn[..., 2] >= 0).astype(np.uint8)
fidx, w_o, interp = oracle.rasterize(H, W, fvc[..., 2], fvi, ones, valid)
soft_o, prob, idx, typ = oracle.soft_mask(H, W, fvi, fidx, sigmainv=sigmainv, boxlen=boxlen, knum=knum)
g_s = np.random.default_rng(5).normal(size=soft_o.shape).astype(np.float32)
ref = oracle.soft_mask_backward(g_s, fidx, fvi, prob, idx, typ, sigmainv=sigmainv)
d = lambda a, g=False: torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(g)
fvi_t = d(fvi, True)
one, soft, fidx_h = kal.render.mesh.dibr_rasterization(H, W, d(fvc[..., 2]), fvi_t, d(ones), d(fn[..., 2]), sigmainv=sigmainv, boxlen=boxlen, knum=knum)
print("face_idx equal:", np.array_equal(fidx_h.cpu().numpy(), fidx), " soft max err:", float(np.abs(soft.detach().cpu().numpy() - soft_o).max()))
(soft * d(g_s)).sum().backward()
got = fvi_t.grad.cpu().numpy()
err = np.abs(got - ref).reshape(B, F, -1).max(-1)
print("K4 per-face: max |ref| %.3e, max err %.3e" % (np.abs(ref).max(), err.max()))
order = np.dstack(np.unravel_index(np.argsort(-err, axis=None)[:5], err.shape))[0]
for b, f in order:
    print("face (b=%d, f=%d): err %.3e  hip %s  oracle %s" % (b, f, err[b, f], np.round(got[b, f].reshape(-1), 6), np.round(ref[b, f].reshape(-1), 6)))
    ys, xs, ks = np.nonzero(idx[b] == f)
    keep = fidx[b, ys, xs] < 0
    for y, x, k in list(zip(ys[keep], xs[keep], ks[keep]))[:40]:
        n = int((prob[b, y, x] != 0).sum() if False else (idx[b, y, x] >= 0).sum())
        print("    pixel (y=%d,x=%d): slot %d of %d listed, prob %.6g, typ %d, soft %.6g, g_s %.3f" % (y, x, k, n, prob[b, y, x, k], typ[b, y, x, k], soft_o[b, y, x], g_s[b, y, x]))

# ---- which boundary deviates: un-fused chain (shim operators) vs fused render, both against the oracle's full backward ---------------
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_shim_ops as TS
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
dr.knum, dr.boxlen, dr.sigmainv = knum, boxlen, sigmainv
inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
inp["faces"] = faces; inp["face_uvs"] = dr.face_uvs.numpy()[0]
kw = dict(knum=knum, boxlen=boxlen, sigmainv=sigmainv)
rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, no_mask, proj, **kw)
loss_o, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
g_o = oracle.render_backward(inp, H, W, no_mask, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None, **kw)
A2 = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
rgbs, out = dr.render(no_mask=no_mask, **A2)
dr.recon_data(rgbs, gt.to(dev), no_mask=no_mask).backward()
gv = A2["vertices"].grad.cpu().numpy()
ev = np.abs(gv - g_o["vertices"])
print("fused: vertices max err %.3e at %s" % (ev.max(), np.unravel_index(ev.argmax(), ev.shape)))
# the same loss through the un-fused operators, with dibr constants passed down
A1 = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
Tt = torch.from_numpy(T).to(dev)
fvc_t, fvi_t, fn_t = kal.render.mesh.prepare_vertices(vertices=A1["vertices"], faces=dr.faces, camera_proj=dr.cam_proj, camera_transform=Tt)
fvi_t.retain_grad()
nrm = kal.ops.mesh.face_normals(fvc_t, unit=True).unsqueeze(-2).repeat(1, 1, 3, 1)
feats = [torch.ones((B, F, 3, 1), device=dev), dr.face_uvs.to(dev).repeat(B, 1, 1, 1), nrm]
(texmask, texcoord, imnormal), soft_t, fidx_t = kal.render.mesh.dibr_rasterization(H, W, fvc_t[:, :, :, -1], fvi_t, feats, fn_t[:, :, -1], **kw)
for t_ in (texmask, texcoord, imnormal): t_.retain_grad()
texcolor = kal.render.mesh.texture_mapping(texcoord, A1["textures"], mode='bilinear')
coef = kal.render.mesh.spherical_harmonic_lighting(imnormal, A1["lights"])
image = (texcolor * texmask + A1["bg"].permute(0, 2, 3, 1) * (1 - texmask)) * coef.unsqueeze(-1) if no_mask else texcolor * texmask * coef.unsqueeze(-1) + torch.ones_like(texcolor) * (1 - texmask)
r1 = torch.cat([torch.clamp(image, 0, 1), soft_t[..., None]], -1).permute(0, 3, 1, 2)
dr.recon_data(r1, gt.to(dev), no_mask=no_mask).backward()
gv1 = A1["vertices"].grad.cpu().numpy()
ev1 = np.abs(gv1 - g_o["vertices"])
print("un-fused chain: vertices max err %.3e at %s" % (ev1.max(), np.unravel_index(ev1.argmax(), ev1.shape)))
b0, v0, _ = np.unravel_index(ev.argmax(), ev.shape)
inc = np.nonzero((faces == v0).any(1))[0]
print("faces incident to vertex %d: %s" % (v0, inc.tolist()))
print("un-fused per-face dL/dfvi of those faces (image %d):" % b0)
for f in inc:
    print("   f=%d nz=%.3f  %s" % (f, fn[b0, f, 2], np.round(fvi_t.grad[b0, f].cpu().numpy().reshape(-1), 7)))

# ---- fused path per face: the sweep items' partial sums read back from the workspace (RenderLossStep keeps its workspace) -----------
import ctypes
stepmod = importlib.import_module("3d-magic-mirror_amd.step")
N = pkg._native
A3 = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
st = stepmod.RenderLossStep(dr, A3, gt.to(dev), no_mask=no_mask, fused=True)
st.run(); torch.cuda.synchronize()
lay = (ctypes.c_size_t * 16)()
N.lib().mm_debug_workspace_layout.argtypes = [ctypes.POINTER(N.MMRenderDesc), ctypes.POINTER(ctypes.c_size_t)]
assert N.lib().mm_debug_workspace_layout(ctypes.byref(st.d), lay) == 0
ws = st.ws.cpu().numpy()
cap = int(lay[4])
cm = ws[lay[0]:lay[0] + B * F * 8].view(np.int32).reshape(B, F, 2)
part = ws[lay[3]:lay[3] + B * cap * 48].view(np.float32).reshape(B, cap, 12)
print("fused vertex grad err (step path): %.3e" % np.abs(st.grads["vertices"].cpu().numpy() - g_o["vertices"]).max())
un = fvi_t.grad.cpu().numpy().reshape(B, F, 6) * 1.0
fused_face = np.zeros((B, F, 9), np.float64)
for b in range(B):
    for f in range(F):
        a0, n0 = cm[b, f]
        fused_face[b, f] = part[b, a0:a0 + n0, :9].sum(0)
# fused sums are dL/d(face xy in multiplier units); the un-fused operator's are per image unit: x multiplier
ferr = np.abs(fused_face[..., :6] * 1000.0 - un).max(-1)
print("per-face dL/dfvi, fused (x1000) vs un-fused: max err %.3e (max |un-fused| %.3e)" % (ferr.max(), np.abs(un).max()))
for b, f in np.dstack(np.unravel_index(np.argsort(-ferr, axis=None)[:6], ferr.shape))[0]:
    print("   (b=%d,f=%d) nz=%.3f chunks=%d  fused %s  un-fused %s" % (b, f, fn[b, f, 2], cm[b, f, 1], np.round(fused_face[b, f, :6] * 1000.0, 7), np.round(un[b, f], 7)))

items = ws[lay[1]:lay[1] + B * cap * 8].view(np.int32).reshape(B, cap, 2)
nit = ws[lay[2]:lay[2] + B * 8].view(np.int32).reshape(B, 2)
print("nitems:", nit.tolist(), "item_cap", cap)
for (b, f) in ((0, 933), (0, 382)):
    a0, n0 = cm[b, f]
    print("face (b=%d,f=%d): first item %d, chunks %d; items there: %s" % (b, f, a0, n0, items[b, a0:a0 + n0].tolist()))
    for c in range(n0):
        print("     chunk %d partial: %s" % (c, np.round(part[b, a0 + c, :9] * 1.0, 7)))
    hits = [(int(i), items[b, i].tolist()) for i in range(nit[b, 0]) if items[b, i, 0] == f]
    print("     all items of the image that name this face:", hits)
geo_f = fvi[0, 933] * 1000.0
print("face 933 corners (multiplier units):", geo_f.reshape(-1).round(2).tolist())

# ---- split the bad face's sum into K2 (owned pixels: the pixel pass's nine numbers, read back) and K4 (the rest) -------------------
gp = ws[lay[5]:lay[5] + B * H * W * 32].view(np.float32).reshape(B, H, W, 8)
gp2 = ws[lay[6]:lay[6] + B * H * W * 4].view(np.float32).reshape(B, H, W)
fid = st.face_idx.cpu().numpy()
b, f = 0, 933
own = fid[b] == f
k2 = gp[b][own][:, :6].astype(np.float64).sum(0)
print("face 933: owned pixels %d; K2 sum from gp: %s" % (int(own.sum()), np.round(k2, 7)))
print("          fused total - K2 = K4 (fused): %s" % np.round(fused_face[b, f, :6] - k2, 7))
# K4 of the un-fused operator for the same dL/dalpha: soft-only backward with g = gp2 on uncovered pixels
gsoft = np.where(fid < 0, gp2, 0.0).astype(np.float32)
k4_o = oracle.soft_mask_backward(gsoft, fidx, fvi, prob, idx, typ, sigmainv=sigmainv)
print("          K4 by the oracle for the same dL/dalpha (per image unit -> /1000): %s" % np.round(k4_o[b, f].reshape(-1) / 1000.0, 7))
ys, xs = np.nonzero(own)
print("          owned pixel box: y %d..%d x %d..%d" % (ys.min(), ys.max(), xs.min(), xs.max()))
mag = np.abs(gp[b][own][:, :6]).max(1)
o = np.argsort(-mag)[:12]
print("largest per-pixel K2 numbers of face 933 (y, x, six numbers):")
for i in o:
    print("   (%d,%d) %s" % (ys[i], xs[i], np.round(gp[b, ys[i], xs[i], :6], 7)))
print("median magnitude %.3e" % np.median(mag))

# ---- per-pixel K2 of the bad face: oracle.rasterize_backward on one pixel's dL/dinterp at a time (dinterp from the un-fused chain) ----
dint = torch.cat([texmask.grad, texcoord.grad, imnormal.grad], -1).cpu().numpy()            # (B,H,W,6)
featcat = torch.cat(feats, -1).detach().cpu().numpy()
worst = []
for y, x in zip(ys, xs):
    one = np.zeros_like(dint); one[b, y, x] = dint[b, y, x]
    ref_px = oracle.rasterize_backward(one, fidx, fvi, featcat)[0][b, f].reshape(-1)
    worst.append((float(np.abs(ref_px - gp[b, y, x, :6]).max()), y, x, ref_px, gp[b, y, x, :6].copy()))
worst.sort(key=lambda t: -t[0])
print("per-pixel K2, fused pixel pass vs oracle (largest differences):")
for e_, y, x, r_, g_ in worst[:8]:
    print("   (%d,%d) diff %.3e  oracle %s  hip %s  dinterp %s" % (y, x, e_, np.round(r_, 7), np.round(g_, 7), np.round(dint[b, y, x], 6)))
tot_ref = sum(w_[3] for w_ in worst); tot_hip = sum(w_[4] for w_ in worst)
print("sum over the owned pixels: oracle per-pixel %s | hip %s | un-fused face %s" % (np.round(tot_ref, 7), np.round(tot_hip, 7), np.round(un[b, f], 7)))
y0_, x0_ = worst[0][1], worst[0][2]
hip_rgba = st.rgba.cpu().numpy()
print("worst pixel (%d,%d): rgba hip %s | oracle %s | gt %s" % (y0_, x0_, hip_rgba[b, y0_, x0_].tolist(), rgba_o[b, y0_, x0_].tolist(), gt.numpy()[b, :, y0_, x0_].tolist()))
sf = ws[lay[7]:lay[7] + B * H * W * 8].view(np.int32).reshape(B, H, W, 2)
print("   forward clamp bits stored: %d ; un-fused chain image (pre-clamp) at the pixel: %s" % (sf[b, y0_, x0_, 1], image[b, y0_, x0_].detach().cpu().numpy().tolist()))
d_rgb = np.abs(hip_rgba[..., :3] - rgba_o[..., :3])
print("   rgb max abs diff hip vs oracle over the batch: %.3e ; pixels with any rgb diff: %d of %d" % (d_rgb.max(), int((d_rgb.max(-1) > 0).sum()), d_rgb.shape[0] * H * W))
uvp = texcoord[b, y0_, x0_].detach().cpu().numpy()
Ht_, Wt_ = att["textures"].shape[2:]
ixp = ((2 * uvp[0] - 1 + 1) * Wt_ - 1) / 2; iyp = ((-(2 * uvp[1] - 1) + 1) * Ht_ - 1) / 2
print("   uv %s -> ix %.5f iy %.5f  (texture %dx%d)  texmask %.6f  face uvs %s" % (uvp.tolist(), ixp, iyp, Ht_, Wt_, float(texmask[b, y0_, x0_]), dr.face_uvs.numpy()[0][f].reshape(-1).round(4).tolist()))
for (yy, xx) in ((y0_, x0_ - 1), (y0_, x0_ + 1), (y0_ - 1, x0_), (y0_ + 1, x0_)):
    u2 = texcoord[b, yy, xx].detach().cpu().numpy()
    print("   neighbour (%d,%d) face %d uv %s" % (yy, xx, fid[b, yy, xx], u2.round(5).tolist()))
