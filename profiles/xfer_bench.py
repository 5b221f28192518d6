"""Host <-> device transfer modes of the host-pointer entry points (csrc/xfer.cpp): one host-pointer apply_H (psi in, out
back) per mode, with a reused and with a fresh result array.  usage: python profiles/xfer_bench.py [L=30]
Each mode runs in its own process (SD_XFER is read once)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
if os.environ.get("SD_XFER_CHILD"):
    sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as g
    pkg = g.load_package()
    m = pkg.XXZChain(L, nup=L // 2)
    psi = np.random.default_rng(1).standard_normal(2 * m.N).view(np.complex128)
    out = np.empty_like(psi)
    pkg.apply_H(out, psi, m)              # warm: staging buffers, pinned ring, thread team
    res = {"mode": os.environ.get("SD_XFER", "auto"), "L": L, "GB_each_way": m.N * 16 / 1e9}
    t = []
    for _ in range(3):
        t0 = time.time(); pkg.apply_H(out, psi, m); t.append(time.time() - t0)
    res["reused_out_s"] = min(t)
    t = []
    for _ in range(3):
        fresh = np.empty_like(psi)
        t0 = time.time(); pkg.apply_H(fresh, psi, m); t.append(time.time() - t0)
        del fresh
    res["fresh_out_s"] = min(t)
    res["GBs_reused"] = 2 * m.N * 16 / 1e9 / res["reused_out_s"]
    res["GBs_fresh"] = 2 * m.N * 16 / 1e9 / res["fresh_out_s"]
    print(json.dumps(res), flush=True)
else:
    for mode in ("plain", "staged", "register"):
        for extra in ({},) if mode != "staged" else ({}, {"SD_XFER_THREADS": "4"}, {"SD_XFER_THREADS": "8"}, {"SD_XFER_CHUNK_MB": "8"}, {"SD_XFER_CHUNK_MB": "128"}):
            env = dict(os.environ, SD_XFER=mode, SD_XFER_CHILD="1", **extra)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(L)], env=env, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"mode": mode, "failed": r.stderr[-400:]})
            print(line[:-1] + ', "env": %s}' % json.dumps(extra) if line.endswith("}") else line, flush=True)
