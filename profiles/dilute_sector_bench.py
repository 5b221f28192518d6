import os, sys, time, json
sys.path.insert(0, "/root/repo")
import torch
import __graft_entry__ as g
pkg = g.load_package()
L, nup = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
m = pkg.XXZChain(L, nup=nup)
t1 = time.time()
a = torch.randn(m.N, dtype=torch.complex128, device="cuda"); b = torch.empty_like(a)
for _ in range(2): pkg.apply_H(b, a, m)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): pkg.apply_H(b, a, m)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(json.dumps({"L": L, "nup": nup, "N": m.N, "SD_SUFFIX_BITS": os.environ.get("SD_SUFFIX_BITS"), "path": m.device_path, "model_s": round(t1 - t0, 2), "ms": round(ms, 3), "Grows_per_s": round(m.N / ms / 1e6, 2)}), flush=True)
