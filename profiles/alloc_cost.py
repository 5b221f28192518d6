"""What a fresh device allocation of a vector costs on this box: hipMalloc, first touch (hipMemset), a second memset, hipFree.
python profiles/alloc_cost.py [GB ...]   (the recursions' work vectors come from a per-context pool because of this)"""
import ctypes as C
import json
import sys
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
sizes = [float(x) for x in sys.argv[1:]] or [0.64, 2.48, 9.62]
hip.hipDeviceSynchronize()
for gb in sizes:
    n = int(gb * 1e9)
    for rep in range(2):
        p = C.c_void_p()
        t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), n); t1 = time.perf_counter()
        hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipMemset(p, 1, n); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
        hip.hipFree(p); t4 = time.perf_counter()
        print(json.dumps({"GB": gb, "rep": rep, "rc": rc, "hipMalloc_ms": (t1 - t0) * 1e3, "first_touch_memset_ms": (t2 - t1) * 1e3,
                          "second_memset_ms": (t3 - t2) * 1e3, "hipFree_ms": (t4 - t3) * 1e3}), flush=True)
