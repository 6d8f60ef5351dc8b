import sys, importlib, os, torch, time
sys.path.insert(0, "/root/repo")
ts_mod = importlib.import_module("3d-magic-mirror_amd.trainer_step")
r = ts_mod.bench(torch.device("cuda:0"), steps=8, warmup=3)
print({k: r[k] for k in ("images_per_s", "ms_per_step", "render_path_ms", "render_path_share")})
