"""groundstate() (Lanczos with the reference's full re-orthogonalisation, src/Lanczos.jl:87-181) at scale: python profiles/groundstate_bench.py [L] [lanc_m]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
lm = int(sys.argv[2]) if len(sys.argv) > 2 else 100
m = pkg.XXZChain(L, nup=L // 2)
pkg.groundstate(m, lanc_m=3)
for blocked in (True, False, True, False):      # blocks of 8 columns (default) / column by column (the reference's order)
    m.ctx.set_gs_blocked(blocked)
    t0 = time.time()
    E0, psi = pkg.groundstate(m, lanc_m=lm)
    dt = time.time() - t0
    out = np.empty_like(psi)
    pkg.apply_H(out, psi, m)
    print(json.dumps({"what": "groundstate (lanczos, full re-orthogonalisation)", "gram_schmidt": "blocks of 8" if blocked else "column by column",
                      "L": L, "N": m.N, "lanc_m": lm, "seconds": dt, "E0": E0,
                      "E0_per_site": E0 / L, "residual": float(np.linalg.norm(out - E0 * psi))}), flush=True)
m.ctx.set_gs_blocked(True)
