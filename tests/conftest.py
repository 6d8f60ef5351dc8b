import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TEMPLATES = os.path.join(GOLDEN, "templates")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("3d-magic-mirror_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o
    _o.lib()
    return _o


def load_template_npz(name):
    z = np.load(os.path.join(TEMPLATES, name + ".npz"))
    return {k: z[k] for k in z.files}


def make_inputs(template_name, B, H, W, seed=0, ell=1, ratio=1.0, with_bg=True):
    """Seeded synthetic inputs (numpy dict for the oracle + the same as torch tensors) for a template fixture."""
    tmpl = importlib.import_module("3d-magic-mirror_amd.template")
    syn = importlib.import_module("3d-magic-mirror_amd.synthetic")
    m = load_template_npz(template_name)
    vinit = tmpl.normalize_template(torch.from_numpy(m["vertices"]), ell)
    att, gt = syn.synthetic_batch(vinit, B, H, W, seed=seed, with_bg=with_bg)
    face_uvs = tmpl.index_vertices_by_faces(torch.from_numpy(m["uvs"])[None], torch.from_numpy(m["face_uvs_idx"]).long())[0]
    proj = tmpl.generate_perspective_projection(np.arctan(1.0 / 2.5) * 2, ratio=1.0 / ratio).numpy().reshape(3)
    inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
    inp["faces"] = m["faces"].astype(np.int32)
    inp["face_uvs"] = face_uvs.numpy()
    return inp, gt.numpy(), proj
