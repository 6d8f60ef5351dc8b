"""Feasibility probe: ONE batch of B images rendered as S sub-batches of B / S images on S streams, forked from and joined into the
caller's stream every step (what mm_render_forward / backward would do inside the library).  Prints microseconds per whole-batch step
for S = 1, 2, 3, 4, 6, 8 -- S = 1 is the product path -- with the forward and the backward forked together (one join per step) and
separately (a join between them, as two ABI calls would have it).

    python profiles/tools/split_batch_probe.py [config] [steps]
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("3d-magic-mirror_amd")
stepmod = importlib.import_module("3d-magic-mirror_amd.step")


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    name, B, S, ratio = bench.CONFIGS[cfg]
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio, emit_imnormal=True)
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=0)
    att = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    gt = gt.to(dev)
    main_s = torch.cuda.current_stream(dev)
    pool = [torch.cuda.Stream(dev) for _ in range(8)]
    for nsub in (1, 2, 3, 4, 6, 8):
        if B % nsub:
            continue
        n = B // nsub
        subs = []
        for i in range(nsub):
            a = {k: (v[i * n:(i + 1) * n].contiguous() if torch.is_tensor(v) and v.shape[0] == B else v) for k, v in att.items()}
            subs.append(stepmod.RenderLossStep(dr, a, gt[i * n:(i + 1) * n].contiguous(), no_mask=True, fused=True, emit_imnormal=True))
        ev_fork = torch.cuda.Event(); ev_join = [torch.cuda.Event() for _ in range(nsub)]
        ev_mid = [torch.cuda.Event() for _ in range(nsub)]

        def step_joined():
            if nsub == 1:
                subs[0].run(None); return
            ev_fork.record(main_s)
            for i, st in enumerate(subs):
                pool[i].wait_event(ev_fork)
                st.run(pool[i])
                ev_join[i].record(pool[i])
            for i in range(nsub):
                main_s.wait_event(ev_join[i])

        def step_two_calls():
            if nsub == 1:
                subs[0].run_forward(None); subs[0].run_backward(None); return
            ev_fork.record(main_s)
            for i, st in enumerate(subs):
                pool[i].wait_event(ev_fork)
                st.run_forward(pool[i])
                ev_mid[i].record(pool[i])
            for i in range(nsub):
                main_s.wait_event(ev_mid[i])
            ev_fork.record(main_s)
            for i, st in enumerate(subs):
                pool[i].wait_event(ev_fork)
                st.run_backward(pool[i])
                ev_join[i].record(pool[i])
            for i in range(nsub):
                main_s.wait_event(ev_join[i])

        out = []
        for fn in (step_joined, step_two_calls):
            for _ in range(30):
                fn()
            torch.cuda.synchronize(dev)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(nsteps):
                    fn()
                torch.cuda.synchronize(dev)
                ts.append((time.perf_counter() - t0) / nsteps)
            # host-side cost of enqueueing a step (no synchronisation inside)
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            host = (time.perf_counter() - t0) / 50
            torch.cuda.synchronize(dev)
            out.append((float(np.median(ts)) * 1e6, host * 1e6))
        print("%s B=%d as %d x %d: one join %.1f us/step (%.0f img/s; host enqueue %.1f us) | join between fwd and bwd %.1f us/step (%.0f img/s; host %.1f us)"
              % (cfg, B, nsub, n, out[0][0], B / out[0][0] * 1e6, out[0][1], out[1][0], B / out[1][0] * 1e6, out[1][1]), flush=True)


if __name__ == "__main__":
    main()
