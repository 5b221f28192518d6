import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
pkg = g.load_package()
def timed(model, dtype, steps=30):
    a = torch.ones(model.N, dtype=dtype, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): pkg.apply_H(b, a, model)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): pkg.apply_H(b, a, model)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
LL = int(os.environ.get("SD_ABL_L", "30"))
for dt in (torch.float64, torch.complex128):
    for skip in ("", "16", "4", "16", "4", "1", "2", "3", "7"):
        if skip: os.environ["SD_DEBUG_SKIP"] = skip
        else: os.environ.pop("SD_DEBUG_SKIP", None)
        m = pkg.XXZChain(LL, nup=LL // 2)
        print(json.dumps({"dtype": str(dt), "skip": skip, "ms": timed(m, dt)}), flush=True)
        del m
