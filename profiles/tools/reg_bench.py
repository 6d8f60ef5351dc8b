"""Times the mesh-regulariser step (trainer.py:54-74 shape: two attribute sets with every term + one flip-only set, forward and
backward) as (a) the library's HIP kernels and (b) the reference's own formulation in eager torch ops on the same GPU
(oracle/reg_oracle.py run on device tensors -- test infrastructure used here only as the thing to compare against)."""
import sys, importlib, os, time, types, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import reg_oracle as R
pkg = importlib.import_module("3d-magic-mirror_amd")
dev = torch.device("cuda:0")
dr = pkg.DiffRender("/root/repo/tests/golden/templates/smpl_uv_642.npz", 128)
B = 48
opt = types.SimpleNamespace(lambda_reg=1.0, lambda_flipz=0.1, flipL1=False, lambda_edge=0.5, lambda_depth=0.1, lambda_depthR=0.3,
                            lambda_depthC=0.2, lambda_deform=0.05, temp=2.0, L1=True, chamfer=False, azim=1.0, lambda_ic=1.0)
host = types.SimpleNamespace(flip_index=dr.flip_index.to(dev), sign_init=dr.sign_init.to(dev), edges=dr.edges.to(dev), edge2faces=dr.edge2faces.to(dev),
                             vertices_laplacian_matrix=dr.vertices_laplacian_matrix.to(dev), ratio=dr.ratio, lambda_lpl=dr.lambda_lpl, lambda_flat=dr.lambda_flat)
sets = []
for seed in range(3):
    dv = (0.1 * torch.randn(B, dr.num_vertices, 3)).to(dev).requires_grad_(True)
    fn = torch.nn.functional.normalize(torch.randn(B, dr.num_faces, 3), dim=2).to(dev).requires_grad_(True)
    sets.append({"delta_vertices": dv, "face_normals": fn})
vinit = dr.vertices_init[None].to(dev)

def hip():
    M = pkg.mesh_reg
    for s in sets: s["vertices"] = vinit + s["delta_vertices"]
    terms = M.mask(M.LAPLACIAN, M.FLAT, M.FLIP, M.EDGE, M.DEPTH, M.DEPTHR, M.DEPTHC, M.DEFORM)
    le, li = [dr._reg(terms, vertices=A["vertices"], delta=A["delta_vertices"], fn=A["face_normals"], temp=opt.temp) for A in sets[:2]]
    lr = dr._reg(M.mask(M.FLIP), delta=sets[2]["delta_vertices"])
    tot = (le + li).sum() + lr.sum()
    tot.backward()

def torch_ops():
    for s in sets: s["vertices"] = vinit + s["delta_vertices"]
    Ae, Ai, Aire = sets
    tot = R.calc_reg_loss(host, Ae) + R.calc_reg_loss(host, Ai) + R.recon_flip(host, Ae, False) + R.recon_flip(host, Ai, False) + R.recon_flip(host, Aire, False)
    for f in (R.calc_reg_edge, R.calc_reg_depth, R.calc_reg_depthR, R.calc_reg_depthC):
        tot = tot + f(host, Ae["vertices"]) + f(host, Ai["vertices"])
    tot = tot + R.calc_reg_deform(host, Ae["delta_vertices"]) + R.calc_reg_deform(host, Ai["delta_vertices"])
    tot.backward()

for name, fn in (("hip kernels", hip), ("eager torch ops", torch_ops)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 100
    for _ in range(K): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("%-16s %8.1f us per regularisation step (fwd+bwd, B=%d, V=%d)" % (name, dt * 1e6, B, dr.num_vertices))
