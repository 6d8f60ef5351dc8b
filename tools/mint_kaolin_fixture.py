#!/usr/bin/env python3
"""Mint the fixture that PINS this repo's oracle to real kaolin.  Runs ONLY where NVIDIAGameWorks/kaolin v0.12.0 (the reference's pin,
/root/reference/INSTALL.md:31-34) and a CUDA GPU exist -- neither does in the build container or on the MI355X box, which is why
DESIGN.md says "parity unpinned" for the rasteriser rows.  A maintainer with such a machine runs

    python tools/mint_kaolin_fixture.py            # writes tests/golden/kaolin_v0_12.npz (~150 KB)

and commits the file; tests/test_kaolin_pinning.py then stops skipping: it searches the SURVEY Appendix C option bits
(include/mm_render.h MM_OPT_*) for the combination under which the oracle reproduces kaolin's face_idx bit for bit and its image, soft mask
and input gradients to 1e-4 -- and fails if there is none.  The combination it reports becomes the library's default.

What is recorded: the inputs (seeded, numpy) and, from kaolin's own operators called in the order of the reference's DiffRender.render
(networks.py:278-317): face_idx, soft mask, rgba, face_normals, and the gradients of a fixed scalar of the outputs with respect to
vertices, textures, lights and the camera transform.  Nothing of kaolin's source is stored -- numbers only.

Plus two MICRO-CASES that settle, each by itself, the two recalled choices that matter (profiles/r04_appendix_c_table.md: they move 3.5 % / 12-25 %
of the pixels and the vertex / light gradients by 27-75 %; the other four bits change nothing beyond 4e-5 on the BASELINE inputs):
  sh_axis_*   kaolin.render.mesh.spherical_harmonic_lighting on the six axis normals (+-x, +-y, +-z) under nine DISTINCT light coefficients: which
              light multiplies which band (MM_OPT_SH_ORDER_XYZ) is read off directly (networks.py:306);
  bf_*        kaolin.render.mesh.dibr_rasterization of ONE triangle that faces away from the camera (face_normals_z < 0): whether the soft mask
              runs over culled faces too (MM_OPT_SOFT_SKIP_CULLED) is the difference between a silhouette blob and an all-zero mask
              (networks.py:297-299).
tests/test_kaolin_pinning.py reads both (``diagnose``) and prints the verdict per bit next to the full search.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "kaolin_v0_12.npz"))
    ap.add_argument("--template", default=os.path.join(ROOT, "tests", "golden", "templates", "sphere.npz"))
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=32)
    args = ap.parse_args()
    try:
        import torch
        import kaolin as kal
    except ImportError as e:
        sys.exit("this script needs torch + kaolin v0.12.0 on a CUDA machine (%s)" % e)
    if not torch.cuda.is_available():
        sys.exit("kaolin's DIB-R backend is CUDA-only: no CUDA device here")
    dev = torch.device("cuda")
    import importlib
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("3d-magic-mirror_amd")          # host-side template code only (no GPU library is touched here)
    B, S = args.batch, args.size
    H = W = S
    rng = np.random.default_rng(7)
    dr = pkg.DiffRender(args.template, S)                        # template normalised as networks.py:181-194 does
    vinit = dr.vertices_init.numpy().astype(np.float32)
    faces = dr.faces.numpy().astype(np.int64)
    tmpl = np.load(args.template)
    uvs, fuv = tmpl["uvs"].astype(np.float32), tmpl["face_uvs_idx"].astype(np.int64)
    V, F = vinit.shape[0], faces.shape[0]
    inp = {
        "vertices": (vinit[None] + 0.05 * rng.standard_normal((B, V, 3))).astype(np.float32),
        "textures": rng.random((B, 3, 2 * H, W)).astype(np.float32),
        "lights": (np.array([3.0] + [0.0] * 8) + np.array([0.5] + [0.1] * 8) * rng.uniform(-1, 1, (B, 9))).astype(np.float32),
        "bg": rng.random((B, 3, H, W)).astype(np.float32),
        "azimuths": rng.uniform(-180, 180, B).astype(np.float32), "elevations": rng.uniform(0, 30, B).astype(np.float32),
        "distances": rng.uniform(2, 4, B).astype(np.float32), "biases": rng.uniform(-0.3, 0.3, (B, 2)).astype(np.float32),
    }
    # snap half of the vertices to a coarse grid and look straight on in one image: pixel centres then sit exactly on edges and box
    # borders and some faces are exactly edge-on -- the cases the option bits differ in
    inp["vertices"][0] = np.round(inp["vertices"][0] * 16) / 16
    inp["azimuths"][0] = 0.0; inp["elevations"][0] = 0.0; inp["biases"][0] = 0.0; inp["distances"][0] = 2.5
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    for k in ("vertices", "textures", "lights", "bg"):
        t[k].requires_grad_(True)
    # camera exactly as networks.py:278-282 builds it (smr_utils), in torch
    e, a = t["elevations"] * (np.pi / 180), t["azimuths"] * (np.pi / 180)
    cam = torch.stack([t["distances"] * torch.cos(e) * torch.sin(a), t["distances"] * torch.sin(e), t["distances"] * torch.cos(e) * torch.cos(a)], -1)
    at = torch.cat([t["biases"], torch.zeros(B, 1, device=dev)], 1)
    up = torch.tensor([[0.0, 1.0, 0.0]], device=dev).repeat(B, 1)
    z = torch.nn.functional.normalize(cam - at, dim=1, eps=1e-5)
    x = torch.nn.functional.normalize(torch.cross(up, z, dim=1), dim=1, eps=1e-5)
    y = torch.cross(z, x, dim=1)
    rot = torch.stack([x, y, z], 2)
    T = torch.cat([rot, -torch.bmm(cam.unsqueeze(1), rot)], 1).detach().requires_grad_(True)
    proj = kal.render.camera.generate_perspective_projection(float(np.arctan(1.0 / 2.5) * 2), ratio=1.0).to(dev)
    f_t = torch.from_numpy(faces).to(dev)
    face_uvs = kal.ops.mesh.index_vertices_by_faces(torch.from_numpy(uvs).to(dev).unsqueeze(0), torch.from_numpy(fuv).to(dev)).detach()
    fvc, fvi, fn = kal.render.mesh.prepare_vertices(vertices=t["vertices"], faces=f_t, camera_proj=proj, camera_transform=T)
    nrm = kal.ops.mesh.face_normals(fvc, unit=True).unsqueeze(-2).repeat(1, 1, 3, 1)
    feats = [torch.ones((B, F, 3, 1), device=dev), face_uvs.repeat(B, 1, 1, 1), nrm]
    (texmask, texcoord, imnormal), soft, fidx = kal.render.mesh.dibr_rasterization(H, W, fvc[:, :, :, -1], fvi, feats, fn[:, :, -1])
    texcolor = kal.render.mesh.texture_mapping(texcoord, t["textures"], mode="bilinear")
    coef = kal.render.mesh.spherical_harmonic_lighting(imnormal, t["lights"])
    image = (texcolor * texmask + t["bg"].permute(0, 2, 3, 1) * (1 - texmask)) * coef.unsqueeze(-1)
    rgba = torch.cat([torch.clamp(image, 0, 1), soft[..., None]], -1)
    # a fixed scalar of every output: weights from the same generator, recorded
    w_rgba = torch.from_numpy(rng.standard_normal((B, H, W, 4)).astype(np.float32)).to(dev)
    w_fn = torch.from_numpy((1e-3 * rng.standard_normal((B, F, 3))).astype(np.float32)).to(dev)
    ((rgba * w_rgba).sum() + (fn * w_fn).sum()).backward()
    out = {"kaolin_version": np.array(kal.__version__), "H": np.array(H), "W": np.array(W), "faces": faces.astype(np.int32),
           "face_uvs": face_uvs[0].cpu().numpy(), "proj": proj.reshape(3).cpu().numpy(), "transform": T.detach().cpu().numpy(),
           "w_rgba": w_rgba.cpu().numpy(), "w_fn": w_fn.cpu().numpy(),
           "face_idx": fidx.cpu().numpy().astype(np.int32), "soft_mask": soft.detach().cpu().numpy(), "rgba": rgba.detach().cpu().numpy(),
           "face_normals": fn.detach().cpu().numpy(), "imnormal": imnormal.detach().cpu().numpy(),
           "grad_transform": T.grad.cpu().numpy()}
    # ---- micro-case 1: SH band <-> light pairing on the six axis normals
    axis = torch.tensor([[1., 0, 0], [-1., 0, 0], [0, 1., 0], [0, -1., 0], [0, 0, 1.], [0, 0, -1.]], device=dev).reshape(1, 6, 3)
    sh_lights = (0.1 * torch.arange(1, 10, device=dev, dtype=torch.float32)).reshape(1, 9)
    out["sh_axis_normals"] = axis.cpu().numpy(); out["sh_axis_lights"] = sh_lights.cpu().numpy()
    out["sh_axis_coef"] = kal.render.mesh.spherical_harmonic_lighting(axis, sh_lights).cpu().numpy()
    # ---- micro-case 2: one triangle facing away from the camera
    bf_fvi = torch.tensor([[[[-0.5, -0.4], [0.1, 0.6], [0.5, -0.3]]]], device=dev)      # (1,1,3,2) NDC; its orientation is irrelevant: the cull flag is face_normals_z
    bf_fz = torch.full((1, 1, 3), -3.0, device=dev)
    bf_feat = torch.ones((1, 1, 3, 1), device=dev)
    bf_nz = torch.tensor([[-1.0]], device=dev)
    bf_interp, bf_soft, bf_fidx = kal.render.mesh.dibr_rasterization(16, 16, bf_fz, bf_fvi, bf_feat, bf_nz)
    out["bf_fvi"] = bf_fvi.cpu().numpy(); out["bf_fz"] = bf_fz.cpu().numpy(); out["bf_nz"] = bf_nz.cpu().numpy()
    out["bf_soft"] = bf_soft.cpu().numpy(); out["bf_face_idx"] = bf_fidx.cpu().numpy().astype(np.int32)
    for k in ("vertices", "textures", "lights", "bg"):
        out["in_" + k] = inp[k]; out["grad_" + k] = t[k].grad.cpu().numpy()
    for k in ("azimuths", "elevations", "distances", "biases"):
        out["in_" + k] = inp[k]
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, "(kaolin %s, B=%d, %dx%d, F=%d)" % (kal.__version__, B, H, W, F))


if __name__ == "__main__":
    main()
