#!/usr/bin/env python3
"""Diagnostic: per-phase mean shader cycles of the tiled apply (in-kernel s_memtime stamps, debug entry only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = pkg.XXZChain(L, nup=L // 2)
a = torch.randn(m.N, dtype=torch.complex128, device="cuda")
b = torch.empty_like(a)
pkg.apply_H(b, a, m)
torch.cuda.synchronize()
ph = (C.c_double * 8)()
f = pkg.lib().sd_debug_phase_profile
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
rc = f(m.ctx.h, m.h, 2, b.data_ptr(), a.data_ptr(), ph)
names = ["prologue(meta,own issue,list)", "first far issue", "diag+LDS write+lbin", "far-bond loop", "barrier", "suffix bonds", "tile lifetime", "kernel ns"]
print("rc", rc, "L", L, "tiles", m.N, "LS", os.environ.get("SD_SUFFIX_BITS", "default"))
for n, v in zip(names, ph):
    print(f"  {n:32s} {v:12.0f}")
