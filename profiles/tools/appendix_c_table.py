"""What each Appendix-C switch (choices recalled from kaolin v0.12.0, unpinned here) changes on the BASELINE configs 1 and 2:
pixels whose face_idx changes, max |dRGBA|, relative change of the loss and of every input gradient against the default semantics.
Tells a maintainer which bit matters before minting tests/golden/kaolin_v0_12.npz.   python profiles/tools/appendix_c_table.py  (GPU box)
Writes profiles/r04_appendix_c_table.md (the table DESIGN.md section 2 quotes)."""
import sys, importlib, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d-magic-mirror_amd")
N = pkg._native
dev = torch.device("cuda:0")
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
BITS = [("MM_OPT_CULL_STRICT", N.OPT_CULL_STRICT, "C-1 cull `>` instead of `>=`"),
        ("MM_OPT_SOFT_SKIP_CULLED", N.OPT_SOFT_SKIP_CULLED, "C-1 soft mask skips culled faces"),
        ("MM_OPT_BBOX_MIN_CLOSED_MAX_OPEN", N.OPT_BBOX_MIN_CLOSED_MAX_OPEN, "C-4 bbox `[min, max)`"),
        ("MM_OPT_BBOX_HALF_OPEN", N.OPT_BBOX_HALF_OPEN, "C-4 bbox `(min, max)`"),
        ("MM_OPT_BARY_ONE_MINUS", N.OPT_BARY_ONE_MINUS, "C-3 `w0 = 1 - w1 - w2` over `sum + eps`"),
        ("MM_OPT_SH_ORDER_XYZ", N.OPT_SH_ORDER_XYZ, "C-6 SH bands in x,y,z order")]
CONFIGS = [("config 1", "sphere", 4, 64), ("config 2", "smpl_uv_642", 48, 128)]


def run(dr, datt, gt, bits):
    dr.options = bits
    lv = {k: datt[k].detach().clone().requires_grad_(True) for k in LEAVES}
    a = dict(datt); a.update(lv)
    rgbs, _ = dr.render(no_mask=True, **a)
    loss = dr.recon_data(rgbs, gt, no_mask=True)
    loss.backward()
    torch.cuda.synchronize()
    return rgbs.detach(), dr.last_face_idx.clone(), float(loss), {k: lv[k].grad.clone() for k in LEAVES}


lines = ["| config | switch | pixels whose face_idx changes | pixels whose RGBA changes > 1e-4 | max abs dRGBA | rel. change of the loss | largest relative change of an input gradient (which) |",
         "|---|---|---|---|---|---|---|"]
for cname, tmpl, B, S in CONFIGS:
    dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", tmpl + ".npz"), S)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, S, S, seed=0)
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    gt = gt.to(dev)
    r0, f0, l0, g0 = run(dr, datt, gt, 0)
    npx = f0.numel()
    for name, bit, what in BITS:
        r, f, l, g = run(dr, datt, gt, bit)
        dpix = int((f != f0).sum())
        drgba = (r - r0).abs()
        rel = {k: float((g[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)) for k in LEAVES}
        worst = max(rel, key=rel.get)
        lines.append("| %s | `%s` (%s) | %d of %d | %d | %.3g | %.3g | %.3g (%s) |" % (cname, name, what, dpix, npx, int((drgba.amax(1) > 1e-4).sum()),
                                                                                float(drgba.max()), abs(l - l0) / abs(l0), rel[worst], worst))
out = "\n".join(lines)
print(out)
dst = os.path.join(os.environ.get("MM_PROFILE_OUT") or os.path.join(ROOT, "profiles"), "r04_appendix_c_table.md")
open(dst, "w").write("# r04: what each Appendix-C switch changes against the default semantics (HIP path, seed 0 synthetic batches of SURVEY 8(d))\n\n"
                     "`python profiles/tools/appendix_c_table.py` on one MI355X.  Gradients are those of render + recon_data (image_weight 0.1); a relative change is\n"
                     "max |g_bit - g_default| / max |g_default| per input tensor.\n\n" + out + "\n")
