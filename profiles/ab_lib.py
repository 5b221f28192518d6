"""Same-box A/B of two builds of the library (same ABI): python profiles/ab_lib.py <other.so> [L] [reps]
Alternates the in-tree libspindyn.so and the other build, one child process each; apply (c128, f64) and KPM-step times."""
import json
import os
import subprocess
import sys

other = os.path.abspath(sys.argv[1])
L = sys.argv[2] if len(sys.argv) > 2 else "30"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
child = r'''
import json, os, sys, time, torch
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
torch.cuda.synchronize(); t0 = time.time()
op.kpm_moments(phi, 128, 20.0, 0.0)
torch.cuda.synchronize(); res["kpm_step"] = round((time.time() - t0) / 64 * 1e3, 4)
print(json.dumps(res))
'''
for _ in range(reps):
    for tag, path in (("in-tree", None), (os.path.basename(other), other)):
        env = dict(os.environ)
        if path:
            env["SD_LIB_PATH"] = path
        out = subprocess.run([sys.executable, "-c", child, L], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(tag, "L=" + L, line[-1] if line else out.stderr[-600:], flush=True)
