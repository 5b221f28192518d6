"""Rehearsal of kpm_sqw_replicas with several processes on ONE GPU (gloo for the final all-reduce): every rank must
return the same Q x W matrix as the single-process kpm_sqw.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 profiles/replicas_rehearsal.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import __graft_entry__ as g

os.environ["LOCAL_RANK"] = "0"          # all ranks share the one GPU of the box
pkg = g.load_package()
dist.init_process_group("gloo")
L = 16
m = pkg.XXZChain(L, nup=L // 2)
psi0 = np.random.default_rng(4).standard_normal(m.N)
psi0 /= np.linalg.norm(psi0)
q = pkg.momenta(m)
omega = np.arange(0.0, 4.0, 0.1)
a, b = L / 2 + 1.0, 0.0
S = pkg.kpm_sqw_replicas(psi0, m, q, omega, a, b, kpm_m=64)
ref = pkg.kpm_sqw(psi0, m, q, omega, a=a, b=b, kpm_m=64)
ok = np.array_equal(S, ref)
sys.stdout.write("rank %d of %d replicas == single: %s\n" % (dist.get_rank(), dist.get_world_size(), bool(ok)))   # one write: no interleaving
sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
