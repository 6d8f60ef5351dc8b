#!/usr/bin/env python3
"""Mints tests/golden/template_em.npz: outputs of the reference's template EM update (the statements of
/root/reference/trainer.py:1019-1097, which live inline in the training loop) EXECUTED here on seeded inputs, for every
fusion mode.  The statements are read from the reference at minting time and run with the loop's variables supplied by this
script; only inputs and outputs are committed.  Run from the repo root:  python tests/golden/make_golden_em.py"""
import os
import sys
import textwrap
import types

import numpy as np
import torch

REF = "/root/reference/trainer.py"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

mm = importlib.import_module("3d-magic-mirror_amd")


def reference_block():
    lines = open(REF).read().split("\n")
    start = next(i for i, l in enumerate(lines) if "# delete base case with extreme large change" in l)
    end = next(i for i, l in enumerate(lines) if "netE.vertices_init.data = old_template" in l)
    return textwrap.dedent("\n".join(lines[start:end + 1]))


def run_reference(code, dr, all_vertices, all_delta, opt, warm_up):
    from sklearn.cluster import DBSCAN
    V = dr.num_vertices
    netE = types.SimpleNamespace(vertices_init=torch.nn.Parameter(dr.vertices_init[None].clone()), num_vertices=V)
    diff = types.SimpleNamespace(vertices_laplacian_matrix=dr.vertices_laplacian_matrix, vertices_init=dr.vertices_init)
    env = dict(torch=torch, np=np, DBSCAN=DBSCAN, opt=opt, netE=netE, diffRender=diff, warm_up=warm_up,
               all_vertices=all_vertices.clone(), all_delta_vertices=all_delta.clone(), sample_number=all_vertices.shape[0],
               current_delta_vertices=torch.zeros(V, 3), print=lambda *a, **k: None)
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # the loop's .cuda() calls, on a box without a GPU
    try:
        exec(code, env)
    finally:
        torch.Tensor.cuda = cuda
    return netE.vertices_init.data.clone(), env.get("count", -1)


def main():
    code = reference_block()
    dr = mm.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", "sphere.npz"), 32)
    V = dr.num_vertices
    g = torch.Generator().manual_seed(7)
    N = 40
    delta = 0.05 * torch.randn(N, V, 3, generator=g)
    delta[3] *= 12.0                                         # an "extreme bad case" (mean |delta| of the last vertex > 0.4)
    delta[5:15, :, 2] += 0.02
    verts = dr.vertices_init[None] + delta
    verts[20:25, :, 0] += 0.6                               # lopsided samples for the symmetry filter (em == 3)
    out = {"all_vertices": verts.numpy(), "all_delta_vertices": delta.numpy()}
    cases = []
    for em in (1, 2, 3, 4, 5, 6, 7):
        for smooth, cross, white in ((0.0, 0, 0), (0.3, 1, 0), (0.3, 0, 1)):
            opt = types.SimpleNamespace(em=em, smooth=smooth, clip=0.05, em_step=0.8, white=white, cross=cross, topK=0.5, eps=0.9)
            try:
                new_t, count = run_reference(code, dr, verts, delta, opt, warm_up=0.7)
            except Exception as e:                           # a mode the reference itself cannot run on these shapes is pinned as such
                cases.append((em, smooth, cross, white, -1))
                out["raises_em%d_s%g_c%d_w%d" % (em, smooth, cross, white)] = np.asarray(str(type(e).__name__))
                continue
            cases.append((em, smooth, cross, white, int(count)))
            out["new_em%d_s%g_c%d_w%d" % (em, smooth, cross, white)] = new_t.numpy()
    out["cases"] = np.asarray(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "template_em.npz"), **out)
    for c in cases:
        print(c)


if __name__ == "__main__":
    main()
