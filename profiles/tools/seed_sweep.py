import sys, importlib, os, time, torch
sys.path.insert(0, '/root/repo')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
dev = torch.device("cuda:0")
dr = pkg.DiffRender("/root/repo/tests/golden/templates/smpl_uv_642.npz", 128, emit_imnormal=False)
def mk(seed):
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128, seed=seed)
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    return stepmod.RenderLossStep(dr, datt, gt.to(dev), fused=True), float(att["distances"].mean()), float((1.0 / att["distances"] ** 2).mean())
def run(steps, K=600):
    streams = [torch.cuda.Stream(dev) for _ in steps]
    for i in range(200): steps[i % len(steps)].run(streams[i % len(steps)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): steps[i % len(steps)].run(streams[i % len(steps)])
    torch.cuda.synchronize(); return 48 * K / (time.perf_counter() - t0)
res = {}
for seed in (0, 1, 2, 3, 1000, 2000, 3000):
    st, dm, d2 = mk(seed); res[seed] = st
    print("seed %4d: mean dist %.2f mean 1/d^2 %.3f | one stream %.0f img/s | 4 streams same batch %.0f" % (seed, dm, d2, run([st]), run([st, mk(seed)[0], mk(seed)[0], mk(seed)[0]])))
print("4 streams seeds 0,1000,2000,3000: %.0f" % run([res[0], res[1000], res[2000], res[3000]]))
print("4 streams seeds 0,1,2,3: %.0f" % run([res[0], res[1], res[2], res[3]]))
