import sys, os, importlib, numpy as np, torch
ROOT=os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
pkg = importlib.import_module("3d-magic-mirror_amd"); N=pkg._native
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
dev=torch.device("cuda:0")
def run(name,B,S,opt,seed):
    dr = pkg.DiffRender(os.path.join(ROOT,"tests/golden/templates",name+".npz"), S, emit_imnormal=True)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, dr.render_height, dr.image_size, seed=seed)
    datt = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    dr.options=opt
    rgbs,out=dr.render(no_mask=True, **datt)
    dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
    torch.cuda.synchronize()
    return {k: datt[k].grad.clone() for k in LEAVES}
for name,B,S,seed in (("smpl_uv",2,512,12),("smpl_uv",16,512,12),("ellipsoid",48,256,8)):
    r={t:run(name,B,S,o,seed) for t,o in (("default",0),("block",N.OPT_WALK_BLOCK),("wave",N.OPT_WALK_WAVE),("hint",N.OPT_MANY_IN_FLIGHT))}
    for t in ("block","wave","hint"):
        print(name,B,S,t,{k: "%.1e"%(float((r[t][k]-r["default"][k]).abs().max())/max(float(r["default"][k].abs().max()),1e-300)) for k in LEAVES})
