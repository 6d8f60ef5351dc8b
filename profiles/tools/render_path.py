"""The trainer-shaped config-3 step (trainer_step.bench): whole step and render path (four renders + recon_data + regularisers + backward on
detached attributes) through the eager class API, and the same with render #4 as DiffRender.render_geometry ("lean").
   python profiles/tools/render_path.py"""
import sys, importlib, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ts_mod = importlib.import_module("3d-magic-mirror_amd.trainer_step")
for rep in range(2):
    r = ts_mod.bench(torch.device("cuda:0"), steps=8, warmup=3)
    print({k: r[k] for k in ("images_per_s", "ms_per_step", "render_path_ms", "render_path_share", "lean")}, flush=True)
