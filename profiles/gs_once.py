"""one groundstate call for profiling: python profiles/gs_once.py L lanc_m"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
L, lm = int(sys.argv[1]), int(sys.argv[2])
m = pkg.XXZChain(L, nup=L // 2)
pkg.groundstate(m, lanc_m=3)
t0 = time.time(); E0, psi = pkg.groundstate(m, lanc_m=lm); print("seconds", time.time() - t0, "E0", E0)
