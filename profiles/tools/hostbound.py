import sys, time, importlib, os, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
dev = torch.device("cuda:0")
dr = pkg.DiffRender("/root/repo/tests/golden/templates/smpl_uv_642.npz", 128, emit_imnormal=False)
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}; gtd = gt.to(dev)
for ns in (1, 4, 8):
    steps = [stepmod.RenderLossStep(dr, datt, gtd, fused=True) for _ in range(ns)]
    streams = [torch.cuda.Stream(dev) for _ in steps]
    for i in range(30): steps[i % ns].run(streams[i % ns])
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for i in range(K): steps[i % ns].run(streams[i % ns])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(ns, "streams: enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
