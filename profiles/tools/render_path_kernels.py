"""Where the trainer-shaped render path's GPU time goes at config 3 (four renders + recon_data + regularisers + one backward on detached
attributes, trainer_step.render_path_only): device kernels by total time, the library's against torch's own (zero-fills, copies, reductions,
concatenations), for the four-call form and the lean form (render_many + render_geometry).   python profiles/tools/render_path_kernels.py"""
import sys, importlib, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ts_mod = importlib.import_module("3d-magic-mirror_amd.trainer_step")
dev = torch.device("cuda:0")
tmpl = os.path.join(ROOT, "tests", "golden", "templates", "ellipsoid.npz")
from torch.profiler import profile, ProfilerActivity
for lean in (False, True):
    ts = ts_mod.TrainerStep(tmpl, 256, 48, dev, lean=lean)
    ts.step()
    rp = ts.render_path_only()
    for _ in range(3):
        rp()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        rp()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 8
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(4):
            rp()
        torch.cuda.synchronize()
    rows = [(e.key, e.device_time_total / 4.0, e.count / 4.0) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
    rows.sort(key=lambda r: -r[1])
    tot = sum(r[1] for r in rows)
    ours = sum(r[1] for r in rows if "mm::" in r[0])
    print("== lean=%s: render path %.3f ms wall per run; device kernels %.3f ms (library %.3f ms, torch %.3f ms), %d launches per run"
          % (lean, wall * 1e3, tot / 1e3, ours / 1e3, (tot - ours) / 1e3, int(sum(r[2] for r in rows))))
    for k, t, n in rows[:22]:
        print("   %8.1f us  x%-5.1f %s" % (t, n, k[:110]))
    del ts, rp
    torch.cuda.empty_cache()
