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
# intervals between the kernel's stamps (every stamp waits for all outstanding loads, so the phases are serialised here,
# unlike in the production kernel): 0-1 record, own rows / far-bond bases requested and arrived; 1-2 first far stream issued,
# diagonal, own rows to LDS; 2-3 barrier + far-bond loop; 3-4 two stamps back to back (the cost of a stamp); 4-5 suffix bonds
# (+ general bonds); 5-6 epilogue and store
names = ["record + own rows + far-bond list", "first far issue + diagonal + LDS write", "barrier + far-bond loop", "(stamp cost)",
         "suffix bonds (+ general bonds)", "epilogue + store", "tile lifetime", "kernel ns"]
print("rc", rc, "L", L, "tiles", m.N, "LS", os.environ.get("SD_SUFFIX_BITS", "default"))
for n, v in zip(names, ph):
    print(f"  {n:40s} {v:12.0f}")
