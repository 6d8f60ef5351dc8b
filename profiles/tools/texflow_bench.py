"""Times texture-flow sampling (forward + backward to the flow) at the reference's CUB size: (a) mm_texture_flow_*, (b) the
reference's own three torch ops (grid_sample bicubic + flip + cat, model_res.py:597-612) on the same GPU."""
import sys, importlib, time, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("3d-magic-mirror_amd")
dev = torch.device("cuda:0")
B, H = 48, 128
img = torch.rand(B, 3, H, H, device=dev)
ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, H), indexing="ij")
flow = (torch.stack([xs, ys], 0)[None].repeat(B, 1, 1, 1) * 0.95 + 0.1 * torch.randn(B, 2, H, H)).to(dev).requires_grad_(True)
w = torch.randn(B, 3, 2 * H, H, device=dev)
def hip():
    flow.grad = None
    (pkg.sample_texture(img, flow) * w).sum().backward()
def ref():
    flow.grad = None
    t = F.grid_sample(img, flow.permute(0, 2, 3, 1), mode='bicubic', align_corners=True)
    (torch.cat([t, t.flip([2])], dim=2) * w).sum().backward()
for name, fn in (("hip kernels", hip), ("torch ops", ref)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize()
    print("%-12s %7.1f us per forward+backward (B=%d, %dx%d -> texture %dx%d), incl. the weighting mul/sum" % (name, (time.perf_counter() - t0) / 200 * 1e6, B, H, H, 2 * H, H))
