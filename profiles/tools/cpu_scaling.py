"""Thread scaling of the CPU oracle's step (bench.py's cpu_baseline) on this host:   python profiles/tools/cpu_scaling.py [config2]"""
import sys, time, os, importlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle, bench
pkg = importlib.import_module("3d-magic-mirror_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
name, B, S, ratio = bench.CONFIGS[cfg]
dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio)
H, W = dr.render_height, dr.image_size
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=0)
inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
inp["faces"] = dr.faces.numpy().astype(np.int32); inp["face_uvs"] = dr.face_uvs.numpy()[0]
proj = dr.cam_proj.numpy().reshape(3); gtn = gt.numpy()
nmax = oracle.num_threads()
base = None
for th in [t for t in (1, 8, 16, 32, 64, 128, 256) if t <= nmax]:
    oracle.set_threads(th)
    def t(fn, n=3):
        fn(); c = time.perf_counter()
        for _ in range(n): fn()
        return (time.perf_counter() - c) / n
    n = 1 if th == 1 else 3
    ts = t(lambda: oracle.step(inp, gtn, H, W, True, proj, image_weight=0.1), n)
    tf = t(lambda: oracle.render_forward(inp, H, W, True, proj), n)
    rgba = oracle.render_forward(inp, H, W, True, proj)[0]
    tb = t(lambda: oracle.render_backward(inp, H, W, True, proj, rgba), n)
    base = base or ts
    print("%3d threads: step %.1f ms = %.1f img/s (x%.1f) | render_forward %.1f ms, render_backward (incl. its forward) %.1f ms" % (th, ts * 1e3, B / ts, base / ts, tf * 1e3, tb * 1e3), flush=True)
