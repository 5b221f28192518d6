"""S(q, w) at the reference's own documented sizes (examples/example_kpmSqw.jl: L=20, 20 momenta, kpm_m=80; example_lanczosSqw.jl: L=16,
lanc_m=100), momenta sharing their launches (default) against one momentum at a time: python profiles/smallL_sqw_bench.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

sd = g.load_package()


def med(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.time(); r = fn(); ts.append(time.time() - t0)
    return sorted(ts)[reps // 2], r


for L, method, kw in ((20, "kpm", {"kpm_m": 80, "kernel": "jackson"}), (16, "kpm", {"kpm_m": 80}), (16, "lanczos", {"lanc_m": 100, "eta": 0.05}),
                      (20, "lanczos", {"lanc_m": 100, "eta": 0.05})):
    model = sd.XXZChain(L, nup=L // 2)
    E0, psi0 = sd.groundstate(model, lanc_m=100)
    q = sd.momenta(model)
    omega = np.linspace(0.0, 5.0, 100)
    if method == "kpm":
        a, b = sd.get_rescaling_params(sd.apply_H, model, seed=1)
        kw = dict(kw, a=a, b=b)
    res = {"L": L, "N": model.N, "method": method, "momenta": len(q)}
    S = {}
    for batch in (True, False):
        model.ctx.set_q_batch(batch)
        dt, S[batch] = med(lambda: sd.dynamical_structure_factor(model, psi0, q, omega, method=method, **kw))
        res["batched_ms" if batch else "one_at_a_time_ms"] = dt * 1e3
    model.ctx.set_q_batch(True)
    res["speedup"] = res["one_at_a_time_ms"] / res["batched_ms"]
    res["max_abs_diff"] = float(np.abs(S[True] - S[False]).max())
    print(json.dumps(res), flush=True)
