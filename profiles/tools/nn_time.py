"""The chamfer nearest-neighbour search at the trainer's size (B=48, 642 x 642), both directions: microseconds per call (HIP events).
python profiles/tools/nn_time.py [size index] [both|check]      both: only the one-launch form is timed (so that a rocprofv3 --stats row is
that size and that form alone); check: the indices against a float64 brute force first"""
import sys, importlib, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("MM_DBG_LIB"):                                # a variant build of the library (profiles/tools/variant_sweep.py build ...)
    importlib.import_module("3d-magic-mirror_amd")._native.LIB_PATH = os.environ["MM_DBG_LIB"]
ch = importlib.import_module("3d-magic-mirror_amd.chamfer")
dev = torch.device("cuda:0")
SIZES = ((48, 642, 642), (48, 6890, 6890), (384, 642, 642))
ONLY_BOTH, CHECK = "both" in sys.argv[2:], "check" in sys.argv[2:]
if len(sys.argv) > 1:                                           # one size only (so that a rocprofv3 --stats row is that size's)
    SIZES = (SIZES[int(sys.argv[1])],)
for B, n, m in SIZES:
    x, y = torch.randn(B, n, 3, device=dev), torch.randn(B, m, 3, device=dev) * 0.9
    if CHECK and n <= 1024:                                     # the answer, against a float64 brute force (ragged sizes too)
        for xx, yy in ((x[:4], y[:4]), (x[:3, :601], y[:3, :77]), (x[:2, :5], y[:2, :642])):
            xx, yy = xx.contiguous(), yy.contiguous()
            ix, iy = ch.nearest_both(xx, yy)
            d = torch.cdist(xx.double(), yy.double()).pow(2)
            assert torch.equal(ix, d.min(2)[1]) and torch.equal(iy, d.min(1)[1]), "indices differ from the brute force"
        print("indices == float64 brute force")
    for fn, name in ((lambda: ch.nearest_both(x, y), "mm_chamfer_nearest (both directions, one launch)"),
                     (lambda: (ch.nearest_neighbour(x, y), ch.nearest_neighbour(y, x)), "mm_nearest_neighbour x 2"))[:1 if ONLY_BOTH else 2]:
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        print("B=%d %dx%d  %-52s %.1f us per call (host + device, 50 calls back to back)" % (B, n, m, name, e0.elapsed_time(e1) * 1e3 / 50))
