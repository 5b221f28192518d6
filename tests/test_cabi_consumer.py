"""A plain-C program (tests/cabi_consumer.c) compiled with gcc against include/spindyn.h and linked with libspindyn.so:
the C compiler checks the argument TYPES of the prototypes a foreign binding uses (tests/test_cabi_host.py compares
names only).  CPU: it must compile warning-free as C99 and, run without a GPU, leave through the library's no-device path.
GPU: it must reproduce the golden fixture of the dense numpy oracle (tests/golden/L12n6_open.npz)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "spindynamics.jl_amd")


def build(tmp_path):
    import __graft_entry__ as g
    g.load_package().lib()          # libspindyn.so exists (built by __graft_entry__.build)
    exe = str(tmp_path / "cabi_consumer")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cabi_consumer.c"), "-o", exe, "-L", LIBDIR, "-lspindyn", "-lm",
           "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def write_fixture(path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "L12n6_open.npz"))
    N = len(g["states"])
    lo, hi = g["evals_minmax"]
    a, b = g["kpm_ab"]
    with open(path, "wb") as f:
        f.write(struct.pack("<5q", int(g["L"]), int(g["nup"]), N, len(g["kpm_omega"]), int(g["kpm_M"])))
        f.write(struct.pack("<8d", float(g["Jxy"]), float(g["Jz"]), float(g["hz"]), float(g["t"]), float(lo), float(hi),
                            float(a), float(b)))
        f.write(np.ascontiguousarray(g["states"], dtype="<u8").tobytes())
        for k in ("psi_c", "Hpsi_c", "psi0", "expm_psi0", "gs"):
            f.write(np.ascontiguousarray(g[k], dtype=np.complex128).tobytes())
        f.write(np.ascontiguousarray(g["kpm_mu"], dtype=np.float64).tobytes())


def test_c_consumer_compiles_as_c99_and_has_no_cpu_fallback(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "SKIP: no GPU" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_consumer_reproduces_golden_fixture(tmp_path):
    exe = build(tmp_path)
    fx = str(tmp_path / "fixture.bin")
    write_fixture(fx)
    r = subprocess.run([exe, fx], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
