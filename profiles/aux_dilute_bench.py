import sys, time, json
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import __graft_entry__ as g
pkg = g.load_package()
for L, nup in ((40, 10), (44, 8), (36, 9)):
    m = pkg.XXZChain(L, nup=nup)
    a = torch.randn(m.N, dtype=torch.complex128, device="cuda")
    res = {"L": L, "nup": nup, "N": m.N, "path": m.device_path}
    for name, fn in (("Sz_q_vector", lambda: pkg.Sz_q_vector(m, a, 1.3)), ("magnetization_per_site", lambda: pkg.magnetization_per_site(a, m)),
                     ("connected_correlations", lambda: pkg.connected_correlations(a, m))):
        fn(); torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); res[name + "_ms"] = round((time.time() - t0) * 1e3, 2)
    print(json.dumps(res), flush=True)
    del a, m
