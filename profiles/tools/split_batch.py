"""One batch, split: the B images of a step as k independent sub-batches on k HIP streams, forked from and joined to the caller's stream
with events -- does the step (one batch at a time, the bench's `value`) finish sooner than as one launch sequence?
   python profiles/tools/split_batch.py config2 [config3 config5]"""
import sys, importlib, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["config2"]):
    name, B, S, ratio = bench.CONFIGS[cfg]
    dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio, emit_imnormal=True)
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=0)
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    gtd = gt.to(dev)
    for k in (1, 2, 3, 4, 6):
        if B % k:
            continue
        n = B // k
        subs = [stepmod.RenderLossStep(dr, {kk: (v[i * n:(i + 1) * n].contiguous() if torch.is_tensor(v) else v) for kk, v in datt.items()},
                                       gtd[i * n:(i + 1) * n].contiguous(), fused=True, emit_imnormal=True) for i in range(k)]
        streams = [torch.cuda.Stream(dev) for _ in range(k)]
        main = torch.cuda.current_stream(dev)
        fork, joins = torch.cuda.Event(), [torch.cuda.Event() for _ in range(k)]

        def one():
            if k == 1:
                subs[0].run(main); return
            fork.record(main)
            for st, s, j in zip(subs, streams, joins):
                s.wait_event(fork); st.run(s); j.record(s)
            for j in joins:
                main.wait_event(j)
        for _ in range(50): one()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(7):
            t0 = time.perf_counter()
            for _ in range(300): one()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 300)
        print("%s split %d x B=%d: %.1f us per step = %.0f img/s" % (cfg, k, n, best * 1e6, B / best), flush=True)
