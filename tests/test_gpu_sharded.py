"""SURVEY.md 8(e) on the GPU: the HIP render path itself, sharded.  Two processes (both on cuda:0, gloo for the collective -- the
driver's test box has one GPU; RCCL is the same torch.distributed call with backend "nccl") each render their half of a batch of 8
through libmm_render.so, back-propagate to a small attribute-producing module, and average its gradient with the bucketed all-reduce
of parallel.py: the result must be the single-process full-batch step -- the images bit for bit, the gradients to 1e-6.
The reference has nothing to compare with here (trainer.py:94-95 is a broken DataParallel)."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, TEMPLATES

pytestmark = pytest.mark.gpu
B, S, Z = 8, 64, 6


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _module(V, dev):
    torch.manual_seed(0)                                         # the same weights in every process
    return {"Wl": (torch.randn(Z, 9) * 0.05).to(dev).requires_grad_(True), "Wv": (torch.randn(Z, V * 3) * 0.01).to(dev).requires_grad_(True)}


def _step(pkg, dr, att, gt, z, params, dev, lo, hi):
    """loss (mean over images lo..hi), image, face_idx and the module's gradient for the shard [lo, hi)."""
    a = {k: (v[lo:hi].to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    zz = z[lo:hi].to(dev)
    a["lights"] = a["lights"] + zz @ params["Wl"]
    a["vertices"] = a["vertices"] + (zz @ params["Wv"]).view(hi - lo, -1, 3)
    loss, rgbs, _ = dr.render_recon(gt[lo:hi].to(dev), no_mask=True, **a)
    loss.backward()
    return loss.detach(), rgbs.detach().clone(), dr.last_face_idx.clone(), [params["Wl"].grad.clone(), params["Wv"].grad.clone()]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("3d-magic-mirror_amd")
    par = importlib.import_module("3d-magic-mirror_amd.parallel")
    import torch.distributed as dist
    dist.init_process_group("gloo")
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), S)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, S, S, seed=21)
    z = torch.randn(B, Z, generator=torch.Generator().manual_seed(5))
    params = _module(dr.num_vertices, dev)
    lo, hi = par.shard_bounds(B, rank, world)
    loss, rgbs, fidx, grads = _step(pkg, dr, att, gt, z, params, dev, lo, hi)
    torch.cuda.synchronize()
    host = [g.cpu() for g in grads]                              # gloo reduces host memory
    lt = loss.cpu().reshape(1)
    par.allreduce_mean_(host + [lt])
    torch.save({"rgbs": rgbs.cpu(), "fidx": fidx.cpu(), "grads": host, "loss": lt, "lo": lo, "hi": hi}, out % rank)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_process_sharded_render_matches_the_full_batch(pkg, tmp_path):
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), S)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, S, S, seed=21)
    z = torch.randn(B, Z, generator=torch.Generator().manual_seed(5))
    params = _module(dr.num_vertices, dev)
    loss, rgbs, fidx, grads = _step(pkg, dr, att, gt, z, params, dev, 0, B)
    got = [torch.load(out % r) for r in range(2)]
    assert (got[0]["lo"], got[0]["hi"], got[1]["lo"], got[1]["hi"]) == (0, 4, 4, 8)
    # forward: an image does not depend on the batch (or the process) it is rendered in -- bit for bit
    assert torch.equal(torch.cat([got[0]["rgbs"], got[1]["rgbs"]]), rgbs.cpu())
    assert torch.equal(torch.cat([got[0]["fidx"], got[1]["fidx"]]), fidx.cpu())
    # recon_data is a mean of per-image terms: with equal shards the mean of the ranks' losses is the full-batch loss, and the averaged
    # gradient of the module is the full-batch gradient
    assert abs(float(got[0]["loss"]) - float(loss)) < 1e-6 and torch.equal(got[0]["loss"], got[1]["loss"])
    for g0, g1, ref in zip(got[0]["grads"], got[1]["grads"], grads):
        assert torch.equal(g0, g1)                               # every rank holds the same averaged gradient
        err = float((g0 - ref.cpu()).abs().max())
        assert err <= 1e-6 * max(1.0, float(ref.abs().max())), err
        assert float(ref.abs().max()) > 0
