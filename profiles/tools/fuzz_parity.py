"""Randomised HIP-vs-oracle parity over templates, batch sizes, screen sizes, camera distances and dibr constants (face_idx bit-exact, RGBA
1e-4 absolute, every gradient within 1e-4 of ITS OWN maximum -- no floor of 1, tests/parity_bar.py): a hunt for corner cases the fixed parity cases miss (many windows / chunks of candidates in one tile, cooperative
tiles with more than one window, ragged screens, far and near cameras).   python profiles/tools/fuzz_parity.py [cases] [seed]"""
import sys, importlib, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from parity_bar import rel_errors
pkg = importlib.import_module("3d-magic-mirror_amd")
if os.environ.get("MM_DBG_LIB"):                                  # an alternative build of the library (a compile-time variant under test)
    pkg._native.LIB_PATH = os.environ["MM_DBG_LIB"]
    importlib.import_module("3d-magic-mirror_amd.build_native").needs_build = lambda: False
sys.path.insert(0, os.path.join(ROOT, "3d-magic-mirror_amd", "shim"))
import kaolin as kal
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0; ncond = 0; nnegl = 0; npool = 0
only = os.environ.get("MM_FUZZ_ONLY")
for case in range(ncase):
    name = rng.choice(["sphere", "smpl_uv_642", "ellipsoid", "sphere2", "smpl_uv"], p=[0.25, 0.3, 0.15, 0.15, 0.15])
    big = name in ("sphere2", "smpl_uv")
    S = int(rng.choice([8, 16, 20, 24, 32, 40, 50, 64, 72, 96, 128] if not big else [16, 24, 32, 48, 64, 80]))
    ratio = int(rng.choice([1, 1, 2]))
    B = int(rng.integers(1, 5 if big else 9)) if rng.random() < 0.85 or big else int(rng.integers(9, 80))
    no_mask = bool(rng.integers(0, 2))
    knum = int(rng.choice([30, 30, 30, 5, 70]))
    boxlen = float(rng.choice([0.02, 0.02, 0.05, 0.15]))
    sigmainv = float(rng.choice([7000.0, 7000.0, 900.0, 200.0]))
    mode = rng.choice(["default", "far", "near", "mixed"], p=[0.4, 0.25, 0.15, 0.2])
    dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio)
    dr.knum, dr.boxlen, dr.sigmainv = knum, boxlen, sigmainv
    optbit = int(rng.choice([0, 0, 0, 1 << 4, 1 << 5, 1 << 6, 1 << 7, 1 << 8, 1 << 9, (1 << 4) | (1 << 5), (1 << 9) | (1 << 7)]))   # Appendix-C switches, mirrored by the oracle
    # the four forms of the forward walk kernel (workgroup shape x per-batch / compacting queue): identical results by contract
    walk = int(rng.choice([0, 0, 1 << 10, 1 << 11, 2, 4, (1 << 10) | 2, (1 << 10) | 4, (1 << 11) | 4, (1 << 11) | 2]))
    dr.options = int(os.environ.get("MM_OPTIONS", "0")) | optbit | walk
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=int(rng.integers(0, 1 << 30)))
    if mode == "far":
        att["distances"] = torch.full_like(att["distances"], float(rng.uniform(8.0, 30.0)))
    elif mode == "near":
        att["distances"] = torch.full_like(att["distances"], float(rng.uniform(1.5, 1.9)))
    elif mode == "mixed":
        att["distances"] = torch.from_numpy(rng.uniform(1.6, 25.0, size=B).astype(np.float32))
    datt = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
    inp["faces"] = dr.faces.numpy().astype(np.int32); inp["face_uvs"] = dr.face_uvs.numpy()[0]
    proj = dr.cam_proj.numpy().reshape(3)
    tag = "%s B=%d %dx%d no_mask=%d knum=%d boxlen=%g sigmainv=%g %s opt=%d walk=%d" % (name, B, H, W, no_mask, knum, boxlen, sigmainv, mode, optbit, walk)
    api = str(rng.choice(["render+recon_data", "render_recon", "shim operators"], p=[0.4, 0.3, 0.3]))
    if optbit and api == "shim operators":
        api = "render_recon"                                     # (the kaolin-shaped operators take no option bits through their signatures)
    # recon_data's contour term (networks.py:379-388): un-fused at any size, folded into the render kernels at sizes that are multiples of 4
    contour = float(rng.choice([0.0, 0.0, 0.5, 2.0]))
    if api == "render_recon" and (H % 4 or W % 4):
        contour = 0.0
    # upstream gradient: the loss (a batch mean: gradients of 1e-8 ... 1e-2) or, for the un-fused class API, O(1) random weights on every output
    # channel and on face_normals -- (rgbs * w).sum() + (face_normals * wfn).sum() -- which exercises every gradient path at full magnitude
    w_up = wfn_up = None
    if api == "render+recon_data" and rng.random() < 0.4:
        w_up = rng.normal(size=(B, H, W, 4)).astype(np.float32); wfn_up = rng.normal(size=(B, dr.num_faces, 3)).astype(np.float32)
        contour = 0.0
        api = "render + random upstream"
    tag += " | " + api + (" contour=%g" % contour if contour else "")
    if only is not None and case != int(only):                  # (after EVERY draw of the case: a skipped case consumes the same random numbers)
        continue
    try:
        def run_api():
            if api == "render + random upstream":
                rgbs, out = dr.render(no_mask=no_mask, **datt)
                ((rgbs.permute(0, 2, 3, 1) * torch.from_numpy(w_up).to(dev)).sum() + (out["face_normals"] * torch.from_numpy(wfn_up).to(dev)).sum()).backward()
            elif api == "render+recon_data":
                rgbs, out = dr.render(no_mask=no_mask, **datt)
                dr.recon_data(rgbs, gt.to(dev), no_mask=no_mask, contour=contour).backward()
            elif api == "render_recon":
                loss, rgbs, out = dr.render_recon(gt.to(dev), no_mask=no_mask, contour=contour, **datt)
                loss.backward()
            else:                                                    # the reference's own composition of the kaolin-shaped operators
                Tt = torch.from_numpy(oracle.camera(inp["distances"], inp["elevations"], inp["azimuths"], inp["biases"])).to(dev).requires_grad_(True)
                F = dr.num_faces
                fvc_t, fvi_t, fn_t = kal.render.mesh.prepare_vertices(vertices=datt["vertices"], faces=dr.faces, camera_proj=dr.cam_proj, camera_transform=Tt)
                nrm = kal.ops.mesh.face_normals(fvc_t, unit=True).unsqueeze(-2).repeat(1, 1, 3, 1)
                feats = [torch.ones((B, F, 3, 1), device=dev), dr.face_uvs.to(dev).repeat(B, 1, 1, 1), nrm]
                (texmask, texcoord, imnormal), soft_t, fidx_t = kal.render.mesh.dibr_rasterization(
                    H, W, fvc_t[:, :, :, -1], fvi_t, feats, fn_t[:, :, -1], knum=knum, boxlen=boxlen, sigmainv=sigmainv)
                texcolor = kal.render.mesh.texture_mapping(texcoord, datt["textures"], mode='bilinear')
                coef = kal.render.mesh.spherical_harmonic_lighting(imnormal, datt["lights"])
                image = (texcolor * texmask + datt["bg"].permute(0, 2, 3, 1) * (1 - texmask)) * coef.unsqueeze(-1) if no_mask else \
                    texcolor * texmask * coef.unsqueeze(-1) + torch.ones_like(texcolor) * (1 - texmask)
                rgbs = torch.cat([torch.clamp(image, 0, 1), soft_t[..., None]], -1).permute(0, 3, 1, 2)
                dr.recon_data(rgbs, gt.to(dev), no_mask=no_mask, contour=contour).backward()
                dr.last_face_idx = fidx_t.int()
                # the camera chain is the oracle's here (T is a leaf): push dL/dT through it so that all eight gradients can be compared
                dd, de, da, db = oracle.camera_backward(inp["distances"], inp["elevations"], inp["azimuths"], inp["biases"], Tt.grad.cpu().numpy())
                for k, v in (("distances", dd), ("elevations", de), ("azimuths", da), ("biases", db)):
                    datt[k].grad = torch.from_numpy(np.ascontiguousarray(v)).to(dev).reshape(datt[k].shape)
            return rgbs
        rgbs = run_api()
        torch.cuda.synchronize()
        pool = ""
        ndrop = dr.poll_dropped_records()
        if ndrop:
            # the minimum workspace's texture-record array (9/8 records per pixel) overflowed: loud by contract (NaN texture gradients + the status word,
            # tests/test_gpu_parity.py::test_texture_record_pool_overflow_...) -- verified here, then the case is run again with the array enlarged
            assert api != "shim operators" and bool(torch.isnan(datt["textures"].grad).any()), "records dropped without NaN texture gradients"
            for k in LEAVES:
                if datt.get(k) is not None:
                    datt[k].grad = None
            dr.extra_texture_records_per_pixel = 3.0
            rgbs = run_api()
            torch.cuda.synchronize()
            assert dr.poll_dropped_records() == 0
            pool = " | record pool overflowed (%d dropped, NaN texture gradients: loud), run again with room for 4 1/8 records per pixel" % ndrop
            npool += 1
        kw = dict(knum=knum, boxlen=boxlen, sigmainv=sigmainv)
        with oracle.options(optbit):
            rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, no_mask, proj, **kw)
            if w_up is not None:
                dpred_nhwc, wfn_o = w_up, wfn_up
            else:
                loss_o, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, contour=contour, want_grad=True)
                dpred_nhwc, wfn_o = np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None
            g_o = oracle.render_backward(inp, H, W, no_mask, proj, dpred_nhwc, wfn_o, **kw)
        nf = int((dr.last_face_idx.cpu().numpy() != fidx_o).sum())
        errs = {"rgba": float(np.abs(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy() - rgba_o).max())}
        for k in LEAVES:
            if datt.get(k) is None or (k == "bg" and not no_mask):
                continue
            ref = g_o[k]
            errs[k] = rel_errors(datt[k].grad, ref)[0]          # max|got - ref| / max|ref|: no floor
        # NEGL: an input whose WHOLE reference gradient is below 1e-9 of the case's yardstick (e.g. 5e-15 for the vertices beside 1e-3 for the lights: the
        # only silhouette pixel of a far-away 8x8 image is saturated -- alpha = 1 to the last bit -- so d alpha / d geometry is a product of ~1e-13 factors)
        # lies below the resolution of the backward's fixed-point sums (2^-40 of the image's K4 bound): it is compared ABSOLUTELY against that
        # 1e-9 fraction, reported apart and counted, never as ok
        # (the yardstick: the case's largest input gradient or its largest UPSTREAM gradient dL/d(pixel), whichever is larger -- in an 8x8 far-away
        #  white-background image EVERY input gradient can be of that kind, next to a dL/dalpha of 0.03)
        gmax_all = max([float(np.abs(g_o[k]).max()) for k in LEAVES if datt.get(k) is not None and not (k == "bg" and not no_mask)] + [float(np.abs(dpred_nhwc).max())])
        negl = [k for k in errs if k != "rgba" and errs[k] > 1e-4 and float(np.abs(g_o[k]).max()) <= 1e-9 * gmax_all
                and float(np.abs(datt[k].grad.cpu().numpy() - g_o[k]).max()) <= 1e-9 * gmax_all]
        worst = max(v for k, v in errs.items() if k not in negl)
        ok = nf == 0 and worst <= 1e-4
        label = ("NEGL" if negl else "ok  ") if ok else "FAIL"
        nnegl += bool(negl) and ok
        if not ok and nf == 0 and errs["rgba"] <= 1e-4:
            # A gradient beyond 1e-4 of the fp32 oracle with the image itself in agreement: is fp32 the problem?  The same backward in float64 is the
            # judge: where the fp32 ORACLE is itself far from it and the HIP result is no farther (twice its distance + 1e-4), the case is ill-conditioned
            # in fp32 (tiny screens with huge soft margins: a few pixels carry the whole loss) -- reported as COND and counted apart, never as ok.
            with oracle.options(optbit):
                g64 = oracle.render_backward(inp, H, W, no_mask, proj, dpred_nhwc.astype(np.float64), None if wfn_o is None else wfn_o.astype(np.float64), dtype=np.float64, **kw)
            cond = True
            late = []
            for k in LEAVES:
                if datt.get(k) is None or (k == "bg" and not no_mask):
                    continue
                e_hip = rel_errors(datt[k].grad, g64[k])[0]; e_o32 = rel_errors(g_o[k], g64[k])[0]
                if e_hip > 2.0 * e_o32 + 1e-4:
                    late.append((k, e_hip))
            if late:
                # The fp32 oracle can be LUCKY (r06, seed 8308 case 126: dL/d distance of one image = a sum over 2 562 vertices that cancels to 1e-3 of its
                # terms; oracle 4.5e-4 from float64, HIP 2.0e-3).  Second judge, independent of either fp32 result: the float64 backward's own sensitivity to
                # inputs moved by ONE fp32 rounding (vertices and camera scalars times 1 + 2^-24 N(0,1), three draws) -- what ANY fp32 evaluation may suffer
                # before it has done a single operation.  A gradient no farther from float64 than four times that (+ 1e-4) is a conditioning case.
                sens = {k: 0.0 for k, _ in late}
                prng = np.random.default_rng(12345 + case)
                for _ in range(3):
                    inp_p = dict(inp)
                    for kk in ("vertices", "azimuths", "elevations", "distances", "biases"):
                        inp_p[kk] = (inp[kk].astype(np.float64) * (1.0 + 2.0 ** -24 * prng.standard_normal(inp[kk].shape))).astype(np.float64)
                    with oracle.options(optbit):
                        g64p = oracle.render_backward(inp_p, H, W, no_mask, proj, dpred_nhwc.astype(np.float64), None if wfn_o is None else wfn_o.astype(np.float64), dtype=np.float64, **kw)
                    for k, _ in late:
                        sens[k] = max(sens[k], rel_errors(g64p[k], g64[k])[0])
                cond = all(e <= 4.0 * sens[k] + 1e-4 for k, e in late)
                if os.environ.get("MM_FUZZ_DETAIL"):
                    print("      float64 sensitivity to one fp32 rounding of the inputs:", {k: "%.2e (HIP %.2e)" % (sens[k], e) for k, e in late})
                if not cond:
                    # Third judge (r06, seed 7607 case 272: 13 776 faces on 24x24 pixels, dL/d distance of the one image = the sum of 6 890 vertex gradients of up
                    # to 3.8e3 that cancels to 0.86; every one of those terms is itself 1e-3 of the maximum away from float64 in BOTH fp32 results, which agree
                    # with each other to 1e-7 there -- the float64 judge above sees only the inputs' rounding, not the intermediate ones).  The fp32 ORACLE's own
                    # spread when its inputs move by one fp32 rounding: where the checker's result moves by more than the bar under a perturbation no fp32
                    # caller can avoid, the bar cannot be held by any fp32 evaluation.  HIP no farther from the unperturbed fp32 oracle than four times that
                    # spread (+ 1e-4) is a conditioning case.
                    sens32 = {k: 0.0 for k, _ in late}
                    for _ in range(3):
                        inp_p = dict(inp)
                        for kk in ("vertices", "azimuths", "elevations", "distances", "biases"):
                            inp_p[kk] = (inp[kk].astype(np.float64) * (1.0 + 2.0 ** -24 * prng.standard_normal(inp[kk].shape))).astype(np.float32)
                        with oracle.options(optbit):
                            g32p = oracle.render_backward(inp_p, H, W, no_mask, proj, dpred_nhwc, wfn_o, **kw)
                        for k, _ in late:
                            sens32[k] = max(sens32[k], rel_errors(g32p[k], g_o[k])[0])
                    cond = all(errs[k] <= 4.0 * sens32[k] + 1e-4 for k, _ in late)
                    if os.environ.get("MM_FUZZ_DETAIL"):
                        print("      fp32 oracle's own spread under one fp32 rounding of the inputs:", {k: "%.2e (HIP vs oracle %.2e)" % (sens32[k], errs[k]) for k, _ in late})
            if cond:
                label = "COND"; ncond += 1
        print("%s  case %2d  %-90s face_idx diff %d, worst err %.2e (%s)%s" % (label, case, tag, nf, worst, max((k for k in errs if k not in negl), key=errs.get),
                                                                                  (" | negligible gradients (< 1e-9 of the case's largest), compared absolutely: %s" % negl if negl else "") + pool), flush=True)
        bad += label == "FAIL"
        if not ok and os.environ.get("MM_FUZZ_DETAIL"):
            print("      all errors:", {k: "%.2e" % v for k, v in errs.items()})
            with oracle.options(optbit):
              g64 = oracle.render_backward(inp, H, W, no_mask, proj, dpred_nhwc.astype(np.float64), None if wfn_o is None else wfn_o.astype(np.float64), dtype=np.float64, **kw)
            for k in ("vertices", "distances", "azimuths"):
                got = datt[k].grad.cpu().numpy().astype(np.float64); r32 = g_o[k].astype(np.float64); r64 = g64[k]
                i = np.unravel_index(np.abs(got - r32).argmax(), got.shape)
                print("      %s worst at %s: hip %.6e oracle32 %.6e oracle64 %.6e | max|ref| %.3e | hip-vs-64 %.2e, o32-vs-64 %.2e (relative to max)" % (
                    k, i, got[i], r32[i], r64[i], np.abs(r32).max(), np.abs(got - r64).max() / np.abs(r64).max(), np.abs(r32 - r64).max() / np.abs(r64).max()))
    except Exception as e:                                          # noqa: BLE001
        print("EXC   case %2d  %s: %r" % (case, tag, e), flush=True)
        bad += 1
print("failures:", bad, "| ill-conditioned in fp32 (HIP no farther from the float64 backward than the fp32 oracle is):", ncond,
      "| cases with a negligible gradient below the fixed-point resolution (compared absolutely, see NEGL):", nnegl,
      "| cases whose texture-record pool overflowed loudly and were run again with a larger one:", npool)
