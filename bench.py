#!/usr/bin/env python3
"""bench.py -- render + recon-loss + backward images/s of the MI355X path (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of B images: render forward (+ fused recon_data forward) -> recon_data
backward -> render backward, gradients to vertices, textures, lights, bg, distances, elevations, azimuths, biases; inputs are
resident in HBM, and every step does all of that work on a full batch.  Workload at every N: BASELINE config 2 (template
smpl_uv_642, B=48 per GPU, 128x128, texture 256x128, no_mask).

What the ONE JSON line (rank 0) reports, all on the same workload (every timed figure is the MEDIAN of --reps repetitions of --steps
steps, each repetition bracketed by barrier + synchronize: the driver's short --steps reproduces the long run):
  value               K steps strictly one after the other on ONE stream, C ABI, recon_data folded into the render kernels, imnormal
                      materialised as the reference does (networks.py:320): "one B=48 batch at a time", the metric as BASELINE.json words it.
  value_four_streams  the same steps enqueued round-robin on --streams HIP streams (default 4): successive steps are independent batches
                      (the reference's trainer issues four renders per iteration, trainer.py:276,345,347,367) and their kernels overlap.
  value_without_imnormal  `value` without the visualise-only imnormal output.
  value_four_streams_hinted  value_four_streams with MM_OPT_MANY_IN_FLIGHT (the caller's hint that several calls share the chip: the kernels take the shapes large
                      batches take on their own -- one tile per workgroup in the forward walk, four lanes per sweep item; DiffRender.options,
                      bit-identical results).
  value_api_undeferred  value_api with DiffRender.defer_recon_fusion = False (rounds 1-5's value_api: recon_data's own backward launch, dL/d image through memory)
  value_api           DiffRender.render -> DiffRender.recon_data -> loss.backward() through the torch.autograd wrappers (the calls
                      trainer.py makes, :276,441,509-518), one stream; recon_data's backward deferred into the render node (round 6).
                      value_api* = the best of three interleaved rounds of 200 steps (host-bound figures on a host with slow spells:
                      host_us_per_step.rounds keeps every round).  value_api*_st: the same with torch.autograd.set_multithreading_enabled(False)
                      (the whole backward on the calling thread: one line in train.py; the default's two thread wake-ups per step are what a dozing
                      host charges 60-100 us for).
  value_api_fused     the same step through DiffRender.render_recon (loss folded into the render kernels).
                      (Rounds 3-4 also reported value_api_graphed*: captured class-API steps, removed in round 5 -- slower than value_api_fused on
                      every box, profiles/r05_api_paths.md.)
  host_us_per_step    host time per step of each API flavour (and of the C-ABI step): 60 steps enqueued on an idle device, clock stopped before the
                      device is waited for -- value_api* are host-bound, so they are comparable between boxes only next to this figure.
  value_shim          the UN-FUSED compatibility path: the kaolin-shaped operators of the import boundary called in the order of the reference's
                      DiffRender.render (networks.py:278-317; shim_chain.py) + recon_data + backward -- what a maintainer gets who only switches sys.path.
Every stream rotates through --rotate distinct synthetic batches (default 8 per stream: > 256 MiB of inputs in total, more than
the Infinity Cache holds), so inputs are not cache-resident from one step to the next.
N > 1: the batch shards across ranks with no data-path collective (weak scaling); what crosses xGMI in a training step is the
gradient of the attribute-producing networks (--grad-mb megabytes of fp32, default 135 = the reference's custom encoders).
One such ring all-reduce takes milliseconds of xGMI time against < 0.1 ms per render step, so it cannot run once per render step (the
trainer hides it behind the encoder); the bench keeps ONE reduction in flight at all times instead -- back to back on a side stream,
concurrently with the render streams, every K steps with K agreed on by all ranks (grad_allreduce.every_k_steps in the line).
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per stream: HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, and two of the bench's streams
# landing on one queue serialise their steps; main() therefore picks, untimed, a set of streams that do not share one.
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

PEAK_HBM_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
CLOCK_MHZ = 2400.0          # MI355X_MICROARCH.md: max shader clock
SIMDS = 1024                # 256 CUs x 4 SIMD-32
VALU_CYCLES = 2.0           # cycles a wave64 VALU instruction occupies its SIMD-32 (MI355X_MICROARCH.md: SIMD-32, 157.3 TF fp32 peak)
VALU_RATE = 875.8           # MEASURED sustained issue rate, wave-instructions per microsecond per SIMD (independent v_fma_f32, 8 waves
                            # per SIMD, all 256 CUs: profiles/tools/valu_calib.hip -> profiles/r02_valu_calibration.json).  That is
                            # 2.74 cycles at the nominal 2.4 GHz, i.e. 2 cycles at the ~1.75 GHz the chip sustains under an all-VALU
                            # load; a SIMD-16 (4 cycles) could not exceed 600.  Floors below use the measured rate.

CONFIGS = {
    # name: (template, B, image_size, ratio)
    "config2": ("smpl_uv_642", 48, 128, 1),
    "config1": ("sphere", 4, 64, 1),
    "config2x8": ("smpl_uv_642", 384, 128, 1),
    "market": ("smpl_uv_642", 48, 64, 2),
    "config3": ("ellipsoid", 48, 256, 1),
    "config5": ("smpl_uv", 16, 512, 1),
}


def algorithmic_bytes(kernel, B, F, V, HW, T, fused=True, imnormal=True):
    """Per-launch algorithmic HBM bytes (DESIGN.md 'Kernels'): the share of SURVEY 8(d)'s A = 140F + 36T + 56HW that one
    kernel owns (fp32, int32 face_idx, every logical tensor crossing HBM once per direction it is needed).
    raster_fwd is SURVEY 8(d)'s forward minus the vertex stage: it reads the face records (52F) and the texture (12T) and writes rgba
    (16HW) and face_idx (4HW); with the loss folded in (fused) it is also the kernel that reads the ground truth (16HW: 8(d) lists it
    under "fwd reads", the un-fused build reads it in recon_partial instead), and when attributes['imnormal'] is materialised
    (networks.py:320; the headline does) it writes 12HW more.  Not counted although moved: the background it composites over
    (12HW read), the silhouette state it leaves for the backward (8HW)."""
    per_image = {
        "vertex_fwd": 12 * V + 52 * F,
        "raster_fwd": 52 * F + 12 * T + 20 * HW + (16 * HW if fused else 0) + (12 * HW if imnormal else 0),
        "recon_partial": 32 * HW,
        "recon_bwd": 48 * HW,
        "order": 0,
        "pixel_bwd": 52 * F + 12 * T + 20 * HW,
        "gather_bwd": 36 * F + 12 * T,
        "vertex_bwd": 36 * F + 24 * V,
    }.get(kernel)
    return None if per_image is None else per_image * B


def csrc_digest():
    """sha of the kernel sources: the committed PMC summaries (profiles/*_latest.json) carry the digest they were measured
    on, so a bench line never quotes counters of kernels that have changed since."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "3d-magic-mirror_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h")):
            h.update(n.encode()); h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def load_counters(name, config, suffix=""):
    """(per-kernel dict or None, note) from profiles/<name>_latest.json for this config, only if measured on these sources."""
    path = os.path.join(ROOT, "profiles", name + "_latest.json")
    try:
        j = json.load(open(path))
    except Exception:
        return None, "no " + os.path.basename(path)
    if config + suffix not in j:
        return None, "not collected for " + config
    if j.get("csrc_digest", {}).get(config) != csrc_digest():
        return None, "stale: kernels changed since the PMC pass (%s)" % j.get("note", "")
    return j[config + suffix], j.get("note", "")


def usable_cpus():
    """(threads worth running, facts): the CPUs this process may actually use -- the affinity mask cut down to the container's CPU-time
    quota (cgroup v2 cpu.max / v1 cfs_quota): on a box whose container is capped at 16 CPUs' worth of time, 128 OpenMP threads only
    time-slice (measured: the oracle's step 2.4x slower on 128 threads than on 16, profiles/tools/cpu_scaling.py)."""
    facts = {"logical_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_quota_cpus": None}
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    facts["cgroup_quota_cpus"] = quota
    n = facts["affinity"]
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n, facts


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="config2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="eager", choices=["hipgraph", "eager", "torch"],
                    help="what `value` times.  eager: C-ABI calls per step on --streams streams; hipgraph: the step replayed as one HIP "
                         "graph; torch: the DiffRender autograd API (same as value_api)")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the independent steps are enqueued on round-robin (value_four_streams)")
    ap.add_argument("--reps", type=int, default=9, help="repetitions of the timed region of --steps steps; the MEDIAN is reported")
    ap.add_argument("--shim-steps", type=int, default=30, help="steps of the un-fused kaolin-shaped operator chain timed for value_shim (0 disables)")
    ap.add_argument("--ddp-encoder", type=int, default=1, help="N > 1: all-reduce the REAL gradient buffer of trainer_step.AttributeNet (after one "
                    "DistributedDataParallel step of it, checked) instead of a synthetic buffer of --grad-mb megabytes")
    ap.add_argument("--rotate", type=int, default=8, help="distinct synthetic input batches per stream, visited in turn")
    ap.add_argument("--unfused", action="store_true", help="recon_data as its own three launches instead of folded into the render kernels")
    ap.add_argument("--settle-seconds", type=float, default=1.0, help="untimed run-in before the warmup steps (clock ramp)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget, all threads (0 disables)")
    ap.add_argument("--cpu-single-seconds", type=float, default=8.0, help="CPU-baseline budget, one thread (0 disables)")
    ap.add_argument("--profile-steps", type=int, default=30, help="extra eager steps with per-kernel HIP events")
    ap.add_argument("--grad-every", type=int, default=None, help="N > 1: launch the gradient all-reduce every K steps (default: derived, see grad_allreduce.policy)")
    ap.add_argument("--api-steps", type=int, default=200, help="steps of the DiffRender autograd path timed for value_api (0 disables)")
    ap.add_argument("--options-steps", type=int, default=200, help="steps of `value`'s step timed under each Appendix C option combination a kaolin fixture may select (0 disables; N=1 only)")
    ap.add_argument("--api-warmup", type=int, default=300, help="untimed steps of EACH autograd-API flavour before any of them is timed")
    ap.add_argument("--trainer-steps", type=int, default=8, help="trainer-shaped config-3 steps timed for value_config3 (0 disables; N=1 only)")
    ap.add_argument("--grad-mb", type=float, default=None, help="fp32 gradient bytes all-reduced per step over RCCL (default 135 when N>1, else 0)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE\n" % (world, args.gpus))
    # MM_BENCH_SHARE_GPU=1 + MM_BENCH_DIST_BACKEND=gloo: every rank on cuda:0 with a host-staged backend -- how tests/test_gpu_bench_ranks.py runs the
    # N > 1 leg on a one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    if os.environ.get("MM_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("MM_BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # The gradient all-reduce shares the GPU with the render kernels it overlaps: a RCCL channel is one workgroup pinned to a CU for the
        # whole reduction, and the library's default for a 135 MB message is dozens of them -- a fifth of the chip taken from a path that is
        # bound by how many of its own waves are in flight.  Eight channels (3 % of the CUs) still move a ring step per xGMI link; the cadence
        # below adapts to whatever the reduction then takes.  (A caller's own setting wins.)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("3d-magic-mirror_amd")
    stepmod = importlib.import_module("3d-magic-mirror_amd.step")
    par = importlib.import_module("3d-magic-mirror_amd.parallel")
    name, B, S, ratio = CONFIGS[args.config]
    tpath = os.path.join(ROOT, "tests", "golden", "templates", name + ".npz")
    dr = pkg.DiffRender(tpath, S, ratio=ratio, emit_imnormal=True)
    H, W = dr.render_height, dr.image_size
    nstreams = max(1, args.streams) if args.mode == "eager" else 1
    nrot = max(1, args.rotate)

    def make_batch(seed):
        att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=seed)
        return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}, gt.to(dev), att, gt

    # batches[s][r]: rotation slot r of stream s; batch (0,0) is the one the CPU baseline and the parity tests use (seed = rank)
    batches, host0 = [], None
    for s_ in range(nstreams):
        row = []
        for r_ in range(nrot):
            datt, gtd, att, gt = make_batch(rank + 1000 * s_ + 100000 * r_)
            if host0 is None:
                host0 = (att, gt)
            row.append((datt, gtd))
        batches.append(row)
    att0, gt0 = host0
    Ht, Wt = att0["textures"].shape[2:]
    input_bytes = sum(v.numel() * 4 for v in batches[0][0][0].values() if torch.is_tensor(v)) + batches[0][0][1].numel() * 4

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(fn, n):
        """n calls of fn bracketed by barrier + synchronize on both sides; max over ranks."""
        barrier(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev); barrier()
        return par.max_over_ranks(time.perf_counter() - t0, dev)

    def host_us(fn, n=60):
        """Host time per call of fn: n calls enqueued back to back on an idle device, clock stopped BEFORE the device is waited for.  What a box's
        host costs the class API per step (value_api* are host-bound: comparable between boxes only next to this figure); the median of 3."""
        vals = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            vals.append((time.perf_counter() - t0) / n * 1e6)
            torch.cuda.synchronize(dev)
        return round(float(np.median(vals)), 1)

    def timed_median(fn, n, reps=None):
        """The timed region repeated: median over the repetitions (every one bracketed like timed()), and all of them."""
        ts_ = [timed(fn, n) for _ in range(max(1, reps or args.reps))]
        return float(np.median(ts_)), ts_

    steps_ = [stepmod.RenderLossStep(dr, batches[s_][0][0], batches[s_][0][1], no_mask=True, fused=not args.unfused, emit_imnormal=True)
              for s_ in range(nstreams)]
    step = steps_[0]
    step_noimn = stepmod.RenderLossStep(dr, batches[0][0][0], batches[0][0][1], no_mask=True, fused=not args.unfused, emit_imnormal=False)

    # ---- the gradient all-reduce a data-parallel trainer adds to every step (N > 1) ---------------------------------------------
    grad_mb = args.grad_mb if args.grad_mb is not None else (135.0 if world > 1 else 0.0)
    reducer = None
    ddp_info = None
    if world > 1 and grad_mb > 0:
        flat = None
        if args.ddp_encoder:
            # the REAL message: the gradient of an attribute-producing network (trainer_step.AttributeNet: two ResNet-18 trunks + conv stacks),
            # produced by one DistributedDataParallel step on this rank's shard (bucketed all-reduce overlapped with its backward, checked
            # against the ranks' mean) -- its flat gradient buffer is what the timed region keeps reducing
            try:
                flat, ddp_info = par.ddp_gradient_buffer(dr, B, H, W, dev, rank)
            except Exception as e:                                # never lose the scaling line to the set-up of its message
                ddp_info = {"error": "%s: %s" % (type(e).__name__, e)}
                flat = None
        if flat is None:
            flat = torch.full((int(grad_mb * 1e6 / 4),), float(rank + 1), device=dev, dtype=torch.float32)
        reducer = par.GradAllReducer(flat)

    streams_ = [torch.cuda.current_stream(dev)]
    if args.mode == "eager" and nstreams > 1:
        # HIP maps streams onto a fixed number of hardware queues; two of our streams landing on ONE queue serialise their steps.
        # Untimed: try a few sets out of twice as many streams and keep the best.
        pool = [torch.cuda.Stream(dev) for _ in range(2 * nstreams)]
        cands = [pool[:nstreams], pool[nstreams:], pool[0::2], pool[1::2]]
        rates = []
        for cand in cands:
            for i in range(3 * nstreams):
                steps_[i % nstreams].run(cand[i % nstreams])
            torch.cuda.synchronize(dev)
            c0 = time.perf_counter()
            for i in range(24 * nstreams):
                steps_[i % nstreams].run(cand[i % nstreams])
            torch.cuda.synchronize(dev)
            rates.append(24 * nstreams / (time.perf_counter() - c0))
        streams_ = cands[int(np.argmax(rates))]

    allreduce_ms = None
    ctr = [0]
    sched = par.ReduceSchedule(reducer)                          # which steps launch the gradient all-reduce (N > 1): off until the cadence is agreed on
    sched1 = par.ReduceSchedule(reducer)                         # the same for the one-stream leg
    every_k = None

    def one_multi():
        k = ctr[0]; ctr[0] += 1
        s_ = k % nstreams
        st = steps_[s_]
        if nrot > 1:
            st.set_inputs(*batches[s_][(k // nstreams) % nrot])
        st.run(streams_[s_] if nstreams > 1 else None)
        sched.step()                  # (a launch waits, on its own stream, for the current stream only; it overlaps the following steps)

    ctr1 = [0]

    def one_single():
        k = ctr1[0]; ctr1[0] += 1
        if nrot > 1:
            step.set_inputs(*batches[0][k % nrot])
        step.run()
        sched1.step()

    # the DiffRender autograd path (what trainer.py calls): imnormal materialised, workspace from the per-object pool
    dr_api = pkg.DiffRender(tpath, S, ratio=ratio)
    leaves_rot = [{k: batches[0][r_][0][k].clone().requires_grad_(True) for k in stepmod.LEAVES} for r_ in range(nrot)]
    ctr2 = [0]

    def one_api():
        k = ctr2[0] % nrot; ctr2[0] += 1
        lv = leaves_rot[k]
        for v in lv.values():
            v.grad = None
        a = dict(batches[0][k][0]); a.update(lv)
        rgbs, _ = dr_api.render(no_mask=True, **a)
        dr_api.recon_data(rgbs, batches[0][k][1], no_mask=True).backward()

    def one_api_fused():                                         # the same step through DiffRender.render_recon (loss folded into the render kernels)
        k = ctr2[0] % nrot; ctr2[0] += 1
        lv = leaves_rot[k]
        for v in lv.values():
            v.grad = None
        a = dict(batches[0][k][0]); a.update(lv)
        dr_api.render_recon(batches[0][k][1], no_mask=True, **a)[0].backward()

    if args.mode == "hipgraph":
        step.capture()
        one = step.replay
    elif args.mode == "torch":
        one = one_api
    else:
        one = one_multi

    # untimed settling phase before the W warmup steps: clocks and queues of a device that has just been idle ramp over hundreds of
    # milliseconds, longer than W short steps last
    # (no collective in here: this loop runs for a TIME, i.e. a different number of steps on every rank, and ranks that issue different
    #  numbers of all-reduces deadlock; the reducer's cadence is switched on below, from counters that are equal on all ranks)
    sched.off()
    settle = time.perf_counter()
    while time.perf_counter() - settle < args.settle_seconds:
        for _ in range(32):
            one()
        torch.cuda.synchronize(dev)
    if reducer is not None:
        # A 135 MB ring all-reduce takes milliseconds of xGMI time; a render step takes well under 0.1 ms: no schedule hides one reduction
        # per render step (the trainer hides it behind the encoder's ~90 ms per iteration).  What the render path can be measured against is
        # xGMI traffic that never stops: ONE reduction in flight at all times, back to back on a side stream, concurrently with the render
        # streams.  Every rank must issue the same collectives in the same order, so the cadence is a fixed step count K agreed on from
        # max-over-ranks timings (untimed here): K = ceil(1.25 x reduction alone / step alone).
        reducer.wait(); torch.cuda.synchronize(dev)
        e3 = timed(lambda: (reducer.launch(), reducer.wait()), 10)
        allreduce_ms = round(e3 / 10 * 1e3, 3)
        ew = timed(lambda: one(), 200)                           # (schedule still off: the step alone)
        every_k = max(1, int(np.ceil(1.25 * (e3 / 10) / (ew / 200)))) if args.grad_every is None else max(1, args.grad_every)
        reducer.launched = 0
        sched.start(every_k)
    for _ in range(args.warmup):
        one()
    if reducer is not None:
        reducer.wait()
        sched.start(every_k)                                     # the first timed step launches a reduction
    elapsed, elapsed_all = timed_median(lambda: one(), args.steps)
    launched_timed = reducer.launched if reducer is not None else 0
    if reducer is not None:
        reducer.wait(); torch.cuda.synchronize(dev)
    loss_value = float(step.loss) if args.mode != "torch" else None

    one_stream = api_value = api_fused_value = shim_value = no_imn_value = api_undeferred_value = options_ab = four_streams_hinted = None
    api_st_values = {}
    host_us_per_step = {}
    e1, e1_all = elapsed, elapsed_all
    if args.mode == "eager":
        if reducer is not None:
            sched1.start(every_k)
        for _ in range(min(args.warmup, 20)):
            one_single()
        e1, e1_all = timed_median(one_single, args.steps)
        one_stream = round(world * B * args.steps / e1, 1)
        if reducer is not None:
            reducer.wait(); sched1.off()
        ctr3 = [0]

        def one_noimn():
            k = ctr3[0]; ctr3[0] += 1
            if nrot > 1:
                step_noimn.set_inputs(*batches[0][k % nrot])
            step_noimn.run()
        for _ in range(min(args.warmup, 20)):
            one_noimn()
        e1n, _ = timed_median(one_noimn, args.steps, reps=3)
        no_imn_value = round(world * B * args.steps / e1n, 1)
        # `value` again under the SURVEY Appendix C option combinations a kaolin fixture may select (tests/test_kaolin_pinning.py; parity at these
        # sizes: tests/test_gpu_parity.py::test_the_option_combinations_...): whichever it selects is already timed.  Same step, same inputs.
        if world == 1 and args.options_steps > 0:
            Nn = pkg._native
            combos = [("BARY_ONE_MINUS|BBOX_MIN_CLOSED_MAX_OPEN", Nn.OPT_BARY_ONE_MINUS | Nn.OPT_BBOX_MIN_CLOSED_MAX_OPEN),
                      ("SOFT_SKIP_CULLED", Nn.OPT_SOFT_SKIP_CULLED), ("SH_ORDER_XYZ", Nn.OPT_SH_ORDER_XYZ),
                      ("all four", Nn.OPT_BARY_ONE_MINUS | Nn.OPT_BBOX_MIN_CLOSED_MAX_OPEN | Nn.OPT_SOFT_SKIP_CULLED | Nn.OPT_SH_ORDER_XYZ)]
            options_ab = {"defaults": {"options": 0, "images_per_s": one_stream}}
            dr_opt = pkg.DiffRender(tpath, S, ratio=ratio, emit_imnormal=True)
            for label, bits in combos:
                dr_opt.options = int(bits)
                st_o = stepmod.RenderLossStep(dr_opt, batches[0][0][0], batches[0][0][1], no_mask=True, fused=not args.unfused, emit_imnormal=True)
                ctr4 = [0]

                def one_opt():
                    k = ctr4[0]; ctr4[0] += 1
                    if nrot > 1:
                        st_o.set_inputs(*batches[0][k % nrot])
                    st_o.run()
                for _ in range(20):
                    one_opt()
                eo, _ = timed_median(one_opt, args.options_steps, reps=3)
                options_ab[label] = {"options": int(bits), "images_per_s": round(B * args.options_steps / eo, 1)}
                del st_o
        # Several steps in flight: the chip then runs every launch in many rounds, the regime in which large batches take the one-tile-per-workgroup
        # forward walk on their own (walk_block_mode, csrc/mm_raster_common.h).  The library cannot see the caller's concurrency: a caller that
        # keeps independent steps in flight on several streams says so with MM_OPT_MANY_IN_FLIGHT (DiffRender.options; results bit-identical,
        # tests/test_gpu_parity.py::test_the_two_walk_kernel_shapes_agree_bit_for_bit).  Same steps, same streams, same inputs as value_four_streams.
        if world == 1 and args.options_steps > 0 and nstreams > 1:
            dr_ww = pkg.DiffRender(tpath, S, ratio=ratio, emit_imnormal=True)
            dr_ww.options = int(pkg._native.OPT_MANY_IN_FLIGHT)
            st_w = [stepmod.RenderLossStep(dr_ww, batches[s_][0][0], batches[s_][0][1], no_mask=True, fused=not args.unfused, emit_imnormal=True)
                    for s_ in range(nstreams)]
            ctr5 = [0]

            def one_ww():
                k = ctr5[0]; ctr5[0] += 1
                s_ = k % nstreams
                if nrot > 1:
                    st_w[s_].set_inputs(*batches[s_][(k // nstreams) % nrot])
                st_w[s_].run(streams_[s_])
            for _ in range(20 * nstreams):
                one_ww()
            ew_, _ = timed_median(one_ww, args.steps, reps=3)
            four_streams_hinted = round(B * args.steps / ew_, 1)
            del st_w
    if args.api_steps > 0:
        # Run-in: the FIRST autograd flavour measured in a process used to carry ~80 us of host time per step for its first few hundred steps
        # (measured, profiles/tools/api_noise.py: 175-186 us against 100-108 us from the second pass on, whichever flavour comes first: the
        # engine's thread, the caching allocator's size classes, the host's clocks) -- r05's driver line read 226 us for value_api and 89 us
        # for value_api_fused for that reason.  All three flavours are run in before any of them is timed.
        api_warm = max(10, args.api_warmup)
        for flavour in range(3):
            dr_api.defer_recon_fusion = flavour != 1
            for _ in range(api_warm):
                (one_api_fused if flavour == 2 else one_api)()
        torch.cuda.synchronize(dev)
        # The three flavours INTERLEAVED, three rounds, best round per flavour: this path is host-bound, and the box's host has slow spells of a second
        # or two (all flavours at ~200 us per step instead of ~100: profiles/tools/api_noise2.py, profiles/r06_api_noise.md) that would otherwise land on
        # whichever flavour is being timed; every round's figures are kept in the line (api_rounds).
        flavours = (("api", True, one_api), ("api_undeferred", False, one_api), ("api_fused", True, one_api_fused))
        # ... and each flavour a second time with the WHOLE backward on the calling thread (torch.autograd.set_multithreading_enabled(False): one line in
        # train.py, same results): the default hands every backward to the engine's device thread and back -- two thread wake-ups per step, which is what
        # the host's slow spells are made of (60-100 us per step on a dozing host; profiles/r06_api_noise.md).  *_st = single-threaded backward.
        names = [n for n, _, _ in flavours] + [n + "_st" for n, _, _ in flavours]
        api_rounds = {fl: [] for fl in names}
        host_rounds = {fl: [] for fl in names}
        for _round in range(3):
            for st_mode in (False, True):
                torch.autograd.set_multithreading_enabled(not st_mode)
                for fl, defer, fn in flavours:
                    key = fl + ("_st" if st_mode else "")
                    dr_api.defer_recon_fusion = defer
                    for _ in range(10):
                        fn()
                    e_, _ = timed_median(fn, args.api_steps, reps=1)
                    api_rounds[key].append(round(world * B * args.api_steps / e_, 1))
                    host_rounds[key].append(host_us(fn))
        torch.autograd.set_multithreading_enabled(True)
        dr_api.defer_recon_fusion = True
        api_value, api_undeferred_value, api_fused_value = (max(api_rounds[n]) for n in ("api", "api_undeferred", "api_fused"))
        api_st_values = {"value_" + n: max(api_rounds[n]) for n in names if n.endswith("_st")}
        host_us_per_step["c_abi_one_stream"] = host_us(one_single)
        for fl in names:
            host_us_per_step[fl] = min(host_rounds[fl])
        host_us_per_step["rounds"] = {"images_per_s": api_rounds, "host_us": host_rounds, "statistic": "value_api* = best of three interleaved rounds; host_us = the smallest"}
    if args.shim_steps > 0 and rank == 0:
        # the un-fused kaolin-shaped operator chain in the reference's order (networks.py:278-317): ~40 launches per render, float atomics
        try:
            chain = importlib.import_module("3d-magic-mirror_amd.shim_chain")

            def one_shim():
                k = ctr2[0] % nrot; ctr2[0] += 1
                lv = leaves_rot[k]
                for v in lv.values():
                    v.grad = None
                a = dict(batches[0][k][0]); a.update(lv)
                rgbs, _, _ = chain.render(dr_api, no_mask=True, **a)
                chain.recon_data(dr_api, rgbs, batches[0][k][1]).backward()
            for _ in range(3):
                one_shim()
            torch.cuda.synchronize(dev)
            c0 = time.perf_counter()
            for _ in range(args.shim_steps):
                one_shim()
            torch.cuda.synchronize(dev)
            shim_value = round(B * args.shim_steps / (time.perf_counter() - c0), 1)
        except Exception as e:                                    # a secondary figure must not cost the line
            shim_value = "%s: %s" % (type(e).__name__, e)

    # ---- per-kernel durations (HIP events recorded by the library around each launch, same stream) ------------
    roofline, kernels_us = None, {}
    if rank == 0 and args.profile_steps > 0:
        step.enable_profiling()
        acc = {}
        for i in range(args.profile_steps + 3):
            if nrot > 1:
                step.set_inputs(*batches[0][i % nrot])
            step.run()
            torch.cuda.synchronize(dev)
            if i >= 3:
                for k, v in step.kernel_times_ms().items():
                    acc.setdefault(k, []).append(v * 1e3)
        step.disable_profiling()
        # the mean over the profiled steps, without the rare sample that a host or clock hiccup stretched beyond three times the kernel's median
        # (one such step among thirty used to move a kernel's "average" by its whole duration); how many were dropped is reported
        dropped = 0
        kernels_us_unfiltered = {}
        for k, v in acc.items():
            v = np.asarray(v, dtype=np.float64)
            keep = v <= 3.0 * np.median(v) if np.all(np.isfinite(v)) else np.ones(len(v), bool)
            dropped += int((~keep).sum())
            if np.isfinite(np.mean(v[keep])):
                kernels_us[k] = float(np.mean(v[keep]))
                kernels_us_unfiltered[k] = float(np.mean(v))
        dom = max(kernels_us, key=kernels_us.get)
        # what the TIMED step really moves: the loss is fused unless --unfused, and imnormal is counted iff the profiled step emits it
        step_emits_imnormal = step.imnormal is not None
        nbytes = algorithmic_bytes(dom, B, dr.num_faces, dr.num_vertices, H * W, Ht * Wt, fused=step.fused, imnormal=step_emits_imnormal)
        achieved = nbytes / (kernels_us[dom] * 1e-6) / 1e9
        traffic_all, tnote = load_counters("traffic", args.config)
        traffic_rw, _ = load_counters("traffic", args.config, suffix="_rw")
        step_bytes = (140 * dr.num_faces + 36 * Ht * Wt + 56 * H * W + (12 * H * W if step_emits_imnormal else 0)) * B   # SURVEY 8(d)'s A + the imnormal output the step writes
        step_us_one = (1e6 * B * world / one_stream) if one_stream else None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBPS, 5), "traffic": (traffic_all or {}).get(dom), "traffic_note": tnote,
                    "traffic_read": (traffic_rw or {}).get(dom, [None, None])[0], "traffic_write": (traffic_rw or {}).get(dom, [None, None])[1],
                    "traffic_formula": "2*FETCH_SIZE + WRITE_SIZE (KiB x 1024), separate --pmc passes; per-pattern factors measured on this path's "
                                       "access patterns: profiles/r04_fetch_calibration.json",
                    "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(kernels_us[dom], 3),
                    "avg_launch_us_unfiltered": round(kernels_us_unfiltered[dom], 3),
                    "frac_unfiltered": round(nbytes / (kernels_us_unfiltered[dom] * 1e-6) / 1e9 / PEAK_HBM_GBPS, 5),
                    "avg_over": "%d profiled steps, %d kernel samples beyond 3x their kernel's median dropped" % (args.profile_steps, dropped),
                    "duration_source": "hip_events (recorded by the library around the launch on its own stream; ~2.5 us longer per kernel than "
                                       "rocprofv3's kernel-trace durations, which profiles/*_kernel_stats.md quote)",
                    "algorithmic_includes": "face records 52F, texture 12T, rgba 16HW, face_idx 4HW" + (", ground truth 16HW (fused loss)" if step.fused else "")
                                            + (", imnormal 12HW" if step_emits_imnormal else ""),
                    "whole_step": {"algorithmic_bytes": step_bytes,
                                   "frac_overlapped": round(step_bytes / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBPS, 5),
                                   "frac_one_stream": round(step_bytes / (step_us_one * 1e-6) / 1e9 / PEAK_HBM_GBPS, 5) if step_us_one else None}}
        # Second roof, reported beside the HBM one: vector-instruction issue.  A wave64 VALU instruction occupies its SIMD-32 for
        # 2 cycles; the chip SUSTAINS VALU_RATE wave-instructions per microsecond per SIMD (measured, profiles/r02_valu_calibration.json),
        # so a step cannot take less than sum(SQ_INSTS_VALU) / (1024 SIMDs * VALU_RATE).  Counts: the committed rocprofv3 --pmc pass.
        insts, vnote = load_counters("valu", args.config)
        if insts:
            tot = sum(v for v in insts.values() if isinstance(v, (int, float)))
            floor_us = tot / SIMDS / VALU_RATE
            step_us = elapsed / args.steps * 1e6
            roofline["valu_issue"] = {"insts_per_step": int(tot), "cycles_per_inst": VALU_CYCLES, "measured_issue_rate_per_simd_per_us": VALU_RATE,
                                      "floor_us_per_step": round(floor_us, 2), "floor_us_at_nominal_clock": round(tot * VALU_CYCLES / SIMDS / CLOCK_MHZ, 2),
                                      "measured_us_per_step": round(step_us, 2), "frac": round(floor_us / step_us, 4),
                                      "frac_one_stream": round(floor_us / step_us_one, 4) if step_us_one else None,
                                      "dominant_kernel_frac": round(insts.get(dom, 0) / SIMDS / VALU_RATE / kernels_us[dom], 4)}
        else:
            roofline["valu_issue"] = {"note": vnote}

    # ---- trainer-shaped config-3 step (BASELINE config 3): encoder -> 4 renders -> recon_data -> regularisers -> backward -> Adam
    config3 = None
    if rank == 0 and world == 1 and args.trainer_steps > 0:
        try:
            ts = importlib.import_module("3d-magic-mirror_amd.trainer_step")
            config3 = ts.bench(dev, steps=args.trainer_steps, warmup=3)
        except Exception as e:                                    # never lose the headline line to the secondary measurement
            config3 = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- CPU baseline: the oracle's step (C restatement of the kaolin DIB-R semantics) on the host cores ----------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att0.items()}
        inp["faces"] = dr.faces.numpy().astype(np.int32)
        inp["face_uvs"] = dr.face_uvs.numpy()[0]
        proj = dr.cam_proj.numpy().reshape(3)
        gtn = gt0.numpy()

        def cpu_rate(nb, budget):
            sub = {k: (v[:nb] if isinstance(v, np.ndarray) and k not in ("faces", "face_uvs") else v) for k, v in inp.items()}
            oracle.step(sub, gtn[:nb], H, W, True, proj, image_weight=dr.image_weight)     # warm
            n, c0 = 0, time.perf_counter()
            while n == 0 or time.perf_counter() - c0 < budget:
                oracle.step(sub, gtn[:nb], H, W, True, proj, image_weight=dr.image_weight)
                n += nb
            return n / (time.perf_counter() - c0), n // nb

        nthreads, cpu_facts = usable_cpus()
        nthreads = min(nthreads, oracle.num_threads()) if oracle.num_threads() > 1 else 1
        oracle.set_threads(nthreads)
        rate_all, n_all = cpu_rate(B, args.cpu_seconds)
        single = None
        if args.cpu_single_seconds > 0:
            oracle.set_threads(1)
            rate_1, n_1 = cpu_rate(B, args.cpu_single_seconds)   # the SAME batch as the all-thread sample
            oracle.set_threads(nthreads)
            single = {"value": round(rate_1, 3), "cores": 1, "sample": "%d steps of the full batch of %d images" % (n_1, B)}
        cpu = {"value": round(rate_all, 2), "unit": "images/s", "cores": nthreads, "kind": "port", "cpu_model": cpu_model(),
               "single_thread": single, "thread_scaling": None if not single else round(rate_all / rate_1, 1), "host": cpu_facts,
               "sample": "%d steps of the full batch of %d images (%s, %dx%d), render+loss+backward; OpenMP over (image, row) in the forward and "
                         "(image, band of rows) with band-private accumulators in the backward's scatters; "
                         "CPU restatement of the kaolin DIB-R semantics, not kaolin" % (n_all, B, name, H, W)}

    # N > 1: proof, in the line itself, that N ranks took part over the communicator the timed region used
    ranks_seen, dist_backend = 1, None
    if world > 1:
        one_t = torch.ones(1, device=dev)
        dist.all_reduce(one_t)
        ranks_seen, dist_backend = int(one_t.item()), dist.get_backend()
    if rank == 0:
        total_images = world * B * args.steps
        head = e1 if args.mode == "eager" else elapsed           # eager: the one-stream leg is the headline
        spread = lambda xs: round(100.0 * (max(xs) - min(xs)) / float(np.median(xs)), 2)
        out = {
            "metric": "render+loss+bwd images/sec at B=%d %dx%d, ~%.1fk faces" % (B, H, W, dr.num_faces / 1000.0),
            "value": round(total_images / head, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "value_is_b48_one_stream": args.mode == "eager",
            "ranks_seen": ranks_seen, "dist_backend": dist_backend,
            "value_per_gpu": round(total_images / head / world, 1),     # the N=1-equivalent figure of an N > 1 line (weak scaling: divide by the N=1 line's value)
            "timing": {"reps": max(1, args.reps), "statistic": "median over the repetitions of the K-step timed region",
                       "spread_pct_of_median": spread(e1_all if args.mode == "eager" else elapsed_all)},
            "config": {"workload": "%s: template %s (V=%d,F=%d), B=%d per GPU, %dx%d, texture %dx%d, no_mask, fwd+loss+bwd to all "
                                   "8 inputs" % (args.config, name, dr.num_vertices, dr.num_faces, B, H, W, Ht, Wt),
                       "value_is": {"eager": "one B=%d batch at a time on one HIP stream (C ABI, fused loss, imnormal materialised); %d independent "
                                             "steps in flight on %d streams = value_four_streams; the DiffRender autograd API = value_api*" % (B, nstreams, nstreams),
                                    "hipgraph": "one step replayed as a HIP graph on one stream",
                                    "torch": "the DiffRender autograd API on one stream"}[args.mode],
                       "mode": args.mode, "streams": nstreams, "fused_loss": not args.unfused,
                       "inputs": "%d distinct batches per stream visited in turn (%.0f MB of inputs in total; Infinity Cache 256 MiB)"
                                 % (nrot, nstreams * nrot * input_bytes / 1e6),
                       "imnormal": "materialised in value, value_four_streams and value_api* like the reference does (networks.py:320); value_without_imnormal leaves it out",
                       "sharding": "batch, no data-path collective",
                       "grad_allreduce": None if reducer is None else
                                         {"mb_per_reduction_per_rank": round(reducer.bytes_per_step() / 1e6, 1), "every_k_steps": every_k,
                                          "launched_in_timed_region": launched_timed, "alone_ms": allreduce_ms, "overlapped": True,
                                          "rccl_max_channels": os.environ.get("NCCL_MAX_NCHANNELS"),
                                          "policy": "one reduction in flight at all times, back to back on a side stream, concurrent with the render "
                                                    "streams: K = ceil(1.25 x reduction alone / step alone), agreed on from max-over-ranks timings"}},
            "value_one_stream": one_stream, "value_four_streams": round(total_images / elapsed, 1) if args.mode == "eager" else None,
            "ms_per_step_four_streams": round(elapsed / args.steps * 1e3, 4) if args.mode == "eager" else None,
            "value_without_imnormal": no_imn_value,
            "value_four_streams_hinted": four_streams_hinted,
            "options_ab": options_ab,
            "value_api": api_value, "value_api_undeferred": api_undeferred_value, "value_api_fused": api_fused_value,
            **api_st_values, "host_us_per_step": host_us_per_step, "value_shim": shim_value, "ddp_encoder": ddp_info,
            "value_config3": None if not config3 else config3.get("images_per_s"), "config3": config3,
            "roofline": roofline, "cpu_baseline": cpu, "kernels_us": {k: round(v, 3) for k, v in kernels_us.items()},
            "loss": loss_value,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                                           # (rank 0's extra legs are done: nobody tears the communicator down under it)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
