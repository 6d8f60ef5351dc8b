"""The chamfer nearest-neighbour search at the trainer's size (B=48, 642 x 642), both directions: microseconds per call (HIP events).
python profiles/tools/nn_time.py [size index]"""
import sys, importlib, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ch = importlib.import_module("3d-magic-mirror_amd.chamfer")
dev = torch.device("cuda:0")
SIZES = ((48, 642, 642), (48, 6890, 6890), (384, 642, 642))
if len(sys.argv) > 1:                                           # one size only (so that a rocprofv3 --stats row is that size's)
    SIZES = (SIZES[int(sys.argv[1])],)
for B, n, m in SIZES:
    x, y = torch.randn(B, n, 3, device=dev), torch.randn(B, m, 3, device=dev) * 0.9
    for fn, name in ((lambda: ch.nearest_both(x, y), "mm_chamfer_nearest (both directions, one launch)"),
                     (lambda: (ch.nearest_neighbour(x, y), ch.nearest_neighbour(y, x)), "mm_nearest_neighbour x 2")):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        print("B=%d %dx%d  %-52s %.1f us per call (host + device, 50 calls back to back)" % (B, n, m, name, e0.elapsed_time(e1) * 1e3 / 50))
