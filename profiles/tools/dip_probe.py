"""How often does the four-stream rate fall into its slow mode, and does it depend on where the steps' buffers lie?
Each trial builds four fresh steps (own inputs, own workspaces) and times 400 steps; `pad` inserts a dummy allocation of a
different size before each step is built (shifts the addresses of everything that follows)."""
import sys, importlib, os, time, torch
sys.path.insert(0, '/root/repo')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
dev = torch.device("cuda:0")
dr = pkg.DiffRender("/root/repo/tests/golden/templates/smpl_uv_642.npz", 128, emit_imnormal=False)
def mk(seed):
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128, seed=seed)
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    return stepmod.RenderLossStep(dr, datt, gt.to(dev), fused=True)
def run(steps, streams, K=400):
    for i in range(100): steps[i % len(steps)].run(streams[i % len(steps)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): steps[i % len(steps)].run(streams[i % len(steps)])
    torch.cuda.synchronize(); return 48 * K / (time.perf_counter() - t0)
streams = [torch.cuda.Stream(dev) for _ in range(4)]
for pad in (False, True):
    out = []
    for trial in range(12):
        keep, steps = [], []
        for i in range(4):
            if pad: keep.append(torch.empty((trial * 4 + i + 1) * 37 * 1024 + 333, dtype=torch.uint8, device=dev))
            steps.append(mk(10 * trial + i))
        out.append(run(steps, streams) / 1e3)
        del steps, keep
        torch.cuda.empty_cache()
    print("pad=%s: " % pad + " ".join("%.0f" % v for v in out))
