"""bench.py's N > 1 leg on the one-GPU test box: two ranks launched exactly as the driver launches them (torch.distributed.run, one
process per rank, 127.0.0.1 rendezvous), both on cuda:0 with the gloo backend (RCCL refuses two ranks on one device; backend "nccl" IS
RCCL in the driver's run and goes through the same torch.distributed calls).  What is checked is the leg's logic on real hardware: the
DistributedDataParallel step that produces the gradient message, the reduction cadence all ranks agree on, barrier + max-over-ranks timing,
and ONE JSON line from rank 0 with the whole-job aggregate."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(900)
def test_two_ranks_produce_one_aggregate_line():
    env = dict(os.environ, MM_BENCH_SHARE_GPU="1", MM_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--reps", "3",
           "--cpu-seconds", "0", "--api-steps", "0", "--shim-steps", "0", "--trainer-steps", "0", "--profile-steps", "0", "--settle-seconds", "0.2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                    # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 10 and line["warmup"] == 2
    assert line["value"] > 0 and line["higher_is_better"] is True
    # whole-job aggregate: 2 ranks x 48 images per step over the max-over-ranks time
    assert abs(line["value"] - 2 * 48 / (line["ms_per_step"] * 1e-3)) / line["value"] < 0.02
    assert line["ranks_seen"] == 2 and line["dist_backend"] == "gloo"       # the line itself proves who took part (an all-reduced ones tensor)
    ga = line["config"]["grad_allreduce"]
    assert ga is not None and ga["every_k_steps"] >= 1
    assert line["ddp_encoder"] is not None and "error" not in line["ddp_encoder"], line["ddp_encoder"]
