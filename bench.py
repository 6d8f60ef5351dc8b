#!/usr/bin/env python3
"""bench.py -- render + recon-loss + backward images/s of the MI355X path (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: mm_render_forward -> mm_recon_data_forward ->
mm_recon_data_backward -> mm_render_backward (gradients to vertices, textures, lights, bg, distances, elevations,
azimuths, biases), inputs resident in HBM.  Every step does all of that work on a full B=48 batch; successive steps are
independent (the reference's trainer issues four independent renders per iteration, trainer.py:276,345,347,367) and are
enqueued round-robin on --streams HIP streams (default 4, one hardware queue each) so that the steps' kernels overlap;
"value_one_stream" is the same loop on a single stream.  Workload at every N: BASELINE config 2 (template smpl_uv_642, B=48 per
GPU, 128x128, texture 256x128, no_mask).  The batch shards across ranks with no data-path collective (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per stream: HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, and two of the bench's streams
# landing on one queue serialise their steps (475 k instead of 600 k images/s); main() therefore picks, untimed, a set of streams
# that do not share one.  Four queues for four streams measured best (615 k; 5: 606 k, 8: 597 k, fewer than 4: < 490 k).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

PEAK_HBM_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)

CONFIGS = {
    # name: (template, B, image_size, ratio)
    "config2": ("smpl_uv_642", 48, 128, 1),
    "config1": ("sphere", 4, 64, 1),
    "config2x3": ("smpl_uv_642", 144, 128, 1),      # experiment: three config-2 batches in one call
    "config2x8": ("smpl_uv_642", 384, 128, 1),
    "market": ("smpl_uv_642", 48, 64, 2),
    "config3": ("ellipsoid", 48, 256, 1),
    "config5": ("smpl_uv", 16, 512, 1),
}


def algorithmic_bytes(kernel, B, F, V, HW, T):
    """Per-launch algorithmic HBM bytes (DESIGN.md 'Kernels'): the share of SURVEY 8(d)'s A = 140F + 36T + 56HW that one
    kernel owns (fp32, int32 face_idx, every logical tensor crossing HBM once per direction it is needed)."""
    per_image = {
        "vertex_fwd": 12 * V + 52 * F,
        "raster_fwd": 52 * F + 12 * T + 20 * HW,
        "recon_partial": 32 * HW,
        "recon_bwd": 48 * HW,
        "bin": 16 * F,
        "order": 0,
        "pixel_bwd": 52 * F + 12 * T + 20 * HW,
        "gather_bwd": 36 * F + 12 * T,
        "vertex_bwd": 36 * F + 24 * V,
    }.get(kernel)
    return None if per_image is None else per_image * B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="config2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="eager", choices=["hipgraph", "eager", "torch"],
                    help="hipgraph: whole step replayed as one HIP graph; eager: 4 ABI calls per step; torch: DiffRender autograd API")
    ap.add_argument("--streams", type=int, default=4,
                    help="successive (independent) steps are enqueued round-robin on this many HIP streams, each with its own buffers")
    ap.add_argument("--unfused", action="store_true", help="recon_data as its own three launches instead of folded into the render kernels")
    ap.add_argument("--resident", action="store_true", help="opt into the LDS-resident forward kernel (MM_OPT_RESIDENT)")
    ap.add_argument("--settle-seconds", type=float, default=1.0, help="untimed run-in before the warmup steps (clock ramp)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--profile-steps", type=int, default=30, help="extra eager steps with per-kernel HIP events")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE\n" % (world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    pkg = importlib.import_module("3d-magic-mirror_amd")
    stepmod = importlib.import_module("3d-magic-mirror_amd.step")
    name, B, S, ratio = CONFIGS[args.config]
    dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio, emit_imnormal=False)
    if args.resident:
        dr.options = pkg._native.OPT_RESIDENT
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=rank)
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    gtd = gt.to(dev)
    Ht, Wt = att["textures"].shape[2:]

    def barrier():
        if world > 1:
            dist.barrier()

    step = stepmod.RenderLossStep(dr, datt, gtd, no_mask=True, fused=not args.unfused)
    if args.mode == "hipgraph":
        step.capture()
        one = step.replay
    elif args.mode == "eager" and args.streams > 1:
        # every stream renders its OWN batch (distinct synthetic draws): no input is shared between the steps in flight
        steps_ = [step]
        for i in range(1, args.streams):
            att_i, gt_i = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=1000 * i + rank)
            datt_i = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att_i.items()}
            steps_.append(stepmod.RenderLossStep(dr, datt_i, gt_i.to(dev), no_mask=True, fused=not args.unfused))
        # HIP maps streams onto a fixed number of hardware queues; two of our streams landing on ONE queue serialise their steps
        # (measured: 475 k instead of 600 k images/s).  Untimed: try a few sets out of twice as many streams and keep the best.
        pool = [torch.cuda.Stream(dev) for _ in range(2 * len(steps_))]
        n_ = len(steps_)
        cands = [pool[:n_], pool[n_:], pool[0::2], pool[1::2]]
        rates = []
        for cand in cands:
            for i in range(3 * n_):
                steps_[i % n_].run(cand[i % n_])
            torch.cuda.synchronize(dev)
            c0 = time.perf_counter()
            for i in range(24 * n_):
                steps_[i % n_].run(cand[i % n_])
            torch.cuda.synchronize(dev)
            rates.append(24 * n_ / (time.perf_counter() - c0))
        streams_ = cands[int(np.argmax(rates))]
        ctr = [0]

        def one():
            i = ctr[0] % len(steps_); ctr[0] += 1
            steps_[i].run(streams_[i])
    elif args.mode == "eager":
        one = step.run
    else:
        leaves = {k: datt[k].clone().requires_grad_(True) for k in stepmod.LEAVES}

        def one():
            for v in leaves.values():
                v.grad = None
            a = dict(datt); a.update(leaves)
            rgbs, _ = dr.render(no_mask=True, **a)
            dr.recon_data(rgbs, gtd, no_mask=True).backward()

    # untimed settling phase before the W warmup steps: clocks and queues of a device that has just been idle (or has just
    # finished another process's work) ramp over hundreds of milliseconds, longer than W short steps last
    settle = time.perf_counter()
    while time.perf_counter() - settle < args.settle_seconds:
        for _ in range(32):
            one()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        one()
    barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    torch.cuda.synchronize(dev); barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_value = float(step.loss) if args.mode != "torch" else None
    # the same K steps strictly one after the other on one stream (no overlap between steps), for reference
    one_stream = None
    if args.mode == "eager" and args.streams > 1:
        torch.cuda.synchronize(dev); barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step.run()
        torch.cuda.synchronize(dev); barrier()
        e1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([e1], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e1 = float(t.item())
        one_stream = round(world * B * args.steps / e1, 1)

    # ---- per-kernel durations (HIP events recorded by the library around each launch, same stream) ------------
    roofline, kernels_us = None, {}
    if rank == 0 and args.profile_steps > 0:
        step.enable_profiling()
        acc = {}
        for i in range(args.profile_steps + 3):
            step.run()
            torch.cuda.synchronize(dev)
            if i >= 3:
                for k, v in step.kernel_times_ms().items():
                    acc.setdefault(k, []).append(v * 1e3)
        step.disable_profiling()
        kernels_us = {k: float(np.mean(v)) for k, v in acc.items() if np.isfinite(np.mean(v))}
        dom = max(kernels_us, key=kernels_us.get)
        nbytes = algorithmic_bytes(dom, B, dr.num_faces, dr.num_vertices, H * W, Ht * Wt)
        achieved = nbytes / (kernels_us[dom] * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.config, {}).get(dom)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBPS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(kernels_us[dom], 3)}
        # What actually bounds this path is vector-instruction issue, not HBM: a wave64 VALU instruction holds
        # its SIMD16 for 4 cycles, so one step cannot take less than sum(SQ_INSTS_VALU) * 4 / (1024 SIMDs * 2.4 GHz).  The counts
        # come from the committed rocprofv3 --pmc pass of this same workload (profiles/valu_latest.json).
        vpath = os.path.join(ROOT, "profiles", "valu_latest.json")
        if os.path.exists(vpath):
            try:
                insts = json.load(open(vpath)).get(args.config)
                if insts:
                    floor_us = sum(insts.values()) * 4.0 / 1024.0 / 2400.0
                    step_us = elapsed / args.steps * 1e6 / 1.0
                    roofline["valu_issue"] = {"insts_per_step": int(sum(insts.values())), "floor_us_per_step": round(floor_us, 2),
                                              "measured_us_per_step": round(step_us, 2), "frac": round(floor_us / step_us, 4),
                                              "dominant_kernel_frac": round(insts.get(dom, 0) * 4.0 / 1024.0 / 2400.0 / kernels_us[dom], 4)}
            except Exception:
                pass

    # ---- CPU baseline: the oracle's step (C restatement of the kaolin DIB-R semantics) on the host cores ----------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
        inp["faces"] = dr.faces.numpy().astype(np.int32)
        inp["face_uvs"] = dr.face_uvs.numpy()[0]
        proj = dr.cam_proj.numpy().reshape(3)
        nb = min(B, 16)
        sub = {k: (v[:nb] if isinstance(v, np.ndarray) and k not in ("faces", "face_uvs") else v) for k, v in inp.items()}
        oracle.step(sub, gt.numpy()[:nb], H, W, True, proj, image_weight=dr.image_weight)     # warm
        n, c0 = 0, time.perf_counter()
        while time.perf_counter() - c0 < args.cpu_seconds:
            oracle.step(sub, gt.numpy()[:nb], H, W, True, proj, image_weight=dr.image_weight)
            n += nb
        cdt = time.perf_counter() - c0
        cpu = {"value": round(n / cdt, 2), "unit": "images/s", "cores": oracle.num_threads(), "kind": "port",
               "sample": "%d steps of the first %d images of the same batch (%s, %dx%d), render+loss+backward, OpenMP over "
                         "(image,row)/(image)" % (n // nb, nb, name, H, W)}

    if rank == 0:
        total_images = world * B * args.steps
        out = {
            "metric": "render+loss+bwd images/sec at B=48 128x128, ~1.3k faces",
            "value": round(total_images / elapsed, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: template %s (V=%d,F=%d), B=%d per GPU, %dx%d, texture %dx%d, no_mask, fwd+loss+bwd to all "
                                   "8 inputs" % (args.config, name, dr.num_vertices, dr.num_faces, B, H, W, Ht, Wt),
                       "mode": args.mode, "streams": args.streams if args.mode == "eager" else 1, "fused_loss": not args.unfused,
                       "sharding": "batch, no data-path collective"},
            "value_one_stream": one_stream, "roofline": roofline, "cpu_baseline": cpu, "kernels_us": {k: round(v, 3) for k, v in kernels_us.items()},
            "loss": loss_value,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
