import sys, importlib, os, torch, numpy as np, time
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ.get('GRAFT_REPO_ROOT','.')
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
dev = torch.device("cuda:0")
for cfg in sys.argv[1:]:
    name, B, S, ratio = bench.CONFIGS[cfg]
    for opt in (0, 1 << 10, 1 << 11):
        dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio, emit_imnormal=False)
        dr.options = opt
        H, W = dr.render_height, dr.image_size
        steps = []
        for s_ in range(4):
            att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=100 * s_)
            steps.append(stepmod.RenderLossStep(dr, {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}, gt.to(dev), fused=True))
        streams = [torch.cuda.Stream(dev) for _ in range(4)]
        for ns in (1, 4):
            for i in range(20): steps[i % ns].run(streams[i % ns])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 200 if B < 100 else 60
            for i in range(n): steps[i % ns].run(streams[i % ns])
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print(cfg, "options", opt, "streams", ns, "%.1f us/step = %.0f img/s" % (dt * 1e6, B / dt), flush=True)
