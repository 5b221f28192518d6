"""What a dependent chain of small applies costs per launch (HIP events around `reps` back-to-back launches inside the library, no
Python in the loop), streaming against latency form of k_apply_tiled, plus wall time of the same loop (host-bound or GPU-bound?):
python profiles/launch_floor.py"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import ctypes as C, json, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
for L in (12, 16, 18, 20, 22):
    m = pkg.XXZChain(L, nup=L // 2)
    res = {"L": L, "N": m.N, "SD_FLAT": os.environ.get("SD_FLAT", "1")}
    for name, dt, code in (("c128", torch.complex128, 2), ("f64", torch.float64, 1)):
        a = torch.ones(m.N, dtype=dt, device="cuda"); b = torch.empty_like(a)
        ms = C.c_float()
        m.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for reps in (50, 2000):
            torch.cuda.synchronize(); t0 = time.time()
            pkg.check(pkg.lib().sd_bench_apply_dev(m.ctx.h, m.h, code, a.data_ptr(), b.data_ptr(), m.N, reps, C.byref(ms)), m.ctx.h)
            wall = (time.time() - t0) / reps * 1e6
        res[name + "_us_per_apply_events"] = round(ms.value * 1e3, 3)
        res[name + "_us_per_apply_wall"] = round(wall, 3)
    print(json.dumps(res), flush=True)
'''
for flat in ("1", "0"):
    env = dict(os.environ, SD_FLAT=flat)
    out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    print(out.stdout, end="")
    if out.returncode:
        print(out.stderr[-800:])
