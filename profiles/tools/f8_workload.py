"""The SURVEY 8(f) kernels at the trainer's CUB sizes (B=48, image 128x128, texture 256x128, 642-vertex template), forward + backward, 60
times each -- meant to be run under `rocprofv3 --kernel-trace --stats` (profiles/tools/f8_profile.sh): the per-kernel averages come from
the profiler, not from host timers.   python profiles/tools/f8_workload.py"""
import sys, importlib, os, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d-magic-mirror_amd")
chamfer = importlib.import_module("3d-magic-mirror_amd.chamfer")
dev = torch.device("cuda:0")
B, S = 48, 128
dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", "smpl_uv_642.npz"), S)
M = pkg.mesh_reg
g = torch.Generator().manual_seed(0)
img = torch.rand(B, 3, S, S, generator=g).to(dev)
ys, xs = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
flow = (torch.stack([xs, ys], 0)[None].repeat(B, 1, 1, 1) * 0.95 + 0.1 * torch.randn(B, 2, S, S, generator=g)).to(dev).requires_grad_(True)
wt = torch.randn(B, 3, 2 * S, S, generator=g).to(dev)
vinit = dr.vertices_init[None].to(dev)
def att(seed):
    a, _ = pkg.synthetic.synthetic_batch(dr.vertices_init, B, S, S, seed=seed)
    a = {k: (v.to(dev).requires_grad_(k in ("vertices", "textures", "lights", "azimuths", "elevations", "distances", "biases")) if torch.is_tensor(v) else v) for k, v in a.items()}
    a["delta_vertices"] = (0.1 * torch.randn(B, dr.num_vertices, 3, generator=g)).to(dev).requires_grad_(True)
    a["face_normals"] = torch.nn.functional.normalize(torch.randn(B, dr.num_faces, 3, generator=g), dim=2).to(dev).requires_grad_(True)
    return a
A1, A2 = att(1), att(2)
terms = M.mask(M.LAPLACIAN, M.FLAT, M.FLIP, M.EDGE, M.DEPTH, M.DEPTHR, M.DEPTHC, M.DEFORM)
x = torch.randn(B, dr.num_vertices, 3, generator=g).to(dev).requires_grad_(True)
y = (torch.randn(B, dr.num_vertices, 3, generator=g) * 0.9).to(dev)
for it in range(60):
    (pkg.sample_texture(img, flow) * wt).sum().backward()
    dr._reg(terms, vertices=vinit + A1["delta_vertices"], delta=A1["delta_vertices"], fn=A1["face_normals"], temp=2.0).sum().backward()
    sum(dr.recon_att(A1, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in A2.items()}, L1=True)).backward()
    chamfer.chamfer_distance(x, y)[0].backward()
torch.cuda.synchronize()
print("done")
