"""A/B of an environment knob in one process tree: python profiles/ab_env.py KNOB v1,v2[,..] [L] [reps]
Each value runs in its own child (knobs are read once per process); apply time for c128 and f64, alternating `reps` times."""
import json
import os
import subprocess
import sys

knob, vals = sys.argv[1], sys.argv[2].split(",")
L = sys.argv[3] if len(sys.argv) > 3 else "30"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
child = r'''
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
L = int(sys.argv[1])
m = pkg.XXZChain(L, nup=L // 2)
res = {}
for name, dt in (("c128", torch.complex128), ("f64", torch.float64)):
    a = torch.ones(m.N, dtype=dt, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): pkg.apply_H(b, a, m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): pkg.apply_H(b, a, m)
    e1.record(); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 30, 4)
    del a, b
op = pkg.ShardedOperator(m, 0, 1)
phi = op.empty(torch.complex128, "cuda"); op.fill_randn(phi, 3); phi /= op.norm(phi)
op.kpm_moments(phi, 8, 20.0, 0.0)
import time
torch.cuda.synchronize(); t0 = time.time()
op.kpm_moments(phi, 128, 20.0, 0.0)
torch.cuda.synchronize(); res["kpm_step"] = round((time.time() - t0) / 64 * 1e3, 4)
print(json.dumps(res))
'''
for _ in range(reps):
    for v in vals:
        env = dict(os.environ)
        if v == "unset":
            env.pop(knob, None)
        else:
            env[knob] = v
        out = subprocess.run([sys.executable, "-c", child, L], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(knob, v, "L=" + L, line[-1] if line else out.stderr[-400:], flush=True)
