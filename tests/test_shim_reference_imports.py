"""SURVEY.md 8(b) row 2, the import boundary: with 3d-magic-mirror_amd/shim on sys.path the reference's OWN modules resolve
every kaolin / pytorch3d name they import at module top (networks.py:6-19, trainer.py:31-40) against this repo's package.

Runs in the build container only (it imports /root/reference, which never ships; skipped where it is absent).  The dense-network
dependencies of the reference that are outside the path (torchvision, timm, tensorboard, pytorch_msssim, ...) are replaced by
inert placeholders -- the shim provides kaolin and pytorch3d, nothing else."""
import importlib
import inspect
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

REF = "/root/reference"
SHIM = os.path.join(ROOT, "3d-magic-mirror_amd", "shim")


def test_shim_exposes_the_names_and_signatures_the_reference_uses():
    sys.path.insert(0, SHIM)
    try:
        import kaolin as kal
        from kaolin.render.camera import generate_perspective_projection
        from kaolin.render.mesh import dibr_rasterization, prepare_vertices, spherical_harmonic_lighting, texture_mapping
        from kaolin.metrics.render import mask_iou
        from pytorch3d.loss import chamfer_distance
    finally:
        sys.path.remove(SHIM)
    assert kal.io.obj.import_mesh and kal.ops.mesh.index_vertices_by_faces and kal.ops.mesh.uniform_laplacian and kal.ops.mesh.face_normals
    assert kal.metrics.render.mask_iou is mask_iou
    p = inspect.signature(dibr_rasterization).parameters
    assert list(p)[:6] == ["height", "width", "face_vertices_z", "face_vertices_image", "face_features", "face_normals_z"]
    assert (p["sigmainv"].default, p["boxlen"].default, p["knum"].default, p["multiplier"].default, p["eps"].default,
            p["rast_backend"].default) == (7000, 0.02, 30, None, None, "cuda")
    assert list(inspect.signature(prepare_vertices).parameters) == ["vertices", "faces", "camera_proj", "camera_rot", "camera_trans", "camera_transform"]
    assert inspect.signature(texture_mapping).parameters["mode"].default == "nearest"
    assert list(inspect.signature(spherical_harmonic_lighting).parameters) == ["imnormal", "lights"]
    assert list(inspect.signature(mask_iou).parameters) == ["lhs_mask", "rhs_mask"]
    assert list(inspect.signature(generate_perspective_projection).parameters)[:2] == ["fovyangle", "ratio"]
    assert list(inspect.signature(chamfer_distance).parameters)[:2] == ["x", "y"]
    assert list(inspect.signature(kal.ops.mesh.face_normals).parameters) == ["face_vertices", "unit"]
    assert list(inspect.signature(kal.io.obj.import_mesh).parameters)[:2] == ["path", "with_materials"]
    for m in ("kaolin", "pytorch3d"):                            # leave the interpreter as found for the other tests
        for k in [k for k in sys.modules if k == m or k.startswith(m + ".")]:
            del sys.modules[k]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_reference_modules_import_against_the_shim():
    code = textwrap.dedent("""
        import sys, types
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        class _Any:
            def __init__(self, *a, **k): pass
            def __call__(self, *a, **k): return _Any()
            def __getattr__(self, n): return _Any()
        import importlib.abc, importlib.machinery
        OUTSIDE = ("torchvision", "timm", "pytorch_msssim", "tensorboardX", "imageio", "cv2", "trimesh", "fid_score",
                   "inception", "matplotlib", "skimage", "lpips", "ROMP")
        class Placeholders(importlib.abc.MetaPathFinder, importlib.abc.Loader):     # any (sub)module of the packages outside the path
            def find_spec(self, name, path=None, target=None):
                if name.split(".")[0] in OUTSIDE:
                    return importlib.machinery.ModuleSpec(name, self, is_package=True)
            def create_module(self, spec): return None
            def exec_module(self, m):
                def ga(n):
                    if n.startswith("__"): raise AttributeError(n)
                    return _Any()
                m.__getattr__ = ga
        sys.meta_path.insert(0, Placeholders())
        class _TB(Placeholders):
            def find_spec(self, name, path=None, target=None):
                if name == "torch.utils.tensorboard": return importlib.machinery.ModuleSpec(name, self, is_package=True)
        sys.meta_path.insert(0, _TB())             # tensorboard itself is not installed; torch's wrapper refuses to import without it
        ds = types.ModuleType("datasets"); ds.__path__ = [%r]; sys.modules["datasets"] = ds   # the reference's datasets/ has no __init__.py
        import kaolin, pytorch3d                                                               # and loses to an installed `datasets`
        assert kaolin.__file__.startswith(%r) and pytorch3d.__file__.startswith(%r)
        import networks, trainer                  # the reference's own modules
        import kaolin.render.mesh as krm
        assert networks.dibr_rasterization is krm.dibr_rasterization and networks.prepare_vertices is krm.prepare_vertices
        assert networks.texture_mapping is krm.texture_mapping and networks.spherical_harmonic_lighting is krm.spherical_harmonic_lighting
        assert networks.chamfer_distance is pytorch3d.loss.chamfer_distance and trainer.mask_iou is kaolin.metrics.render.mask_iou
        assert networks.kal is kaolin and trainer.kal is kaolin
        # DiffRender.__init__ (networks.py:165-256) runs on the shim's host-side template helpers
        import torch
        torch.Tensor.cuda = lambda self, *a, **k: self          # networks.py:252 hard-codes .cuda(); no GPU in this container
        dr = networks.DiffRender(%r, 64)
        assert dr.num_vertices == 642 and dr.num_faces == 1280 and tuple(dr.face_uvs.shape) == (1, 1280, 3, 2)
        assert tuple(dr.vertices_laplacian_matrix.shape) == (642, 642) and tuple(dr.cam_proj.shape) == (3, 1)
        print("REFERENCE-IMPORTS-OK")
    """) % (REF, SHIM, os.path.join(REF, "datasets"), SHIM, SHIM, os.path.join(REF, "template", "sphere.obj"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFERENCE-IMPORTS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
