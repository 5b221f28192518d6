"""The reference's three example scripts (examples/example_kpmSqw.jl, example_lanczosSqw.jl, example_time_evolution.jl), restated with the
same calls and keywords through the Python mirror (examples/*.py), run end to end on the GPU.  Each script checks its own results
(shape, positivity, fidelity against the exact propagator, conservation of total S^z) and exits non-zero otherwise."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args,expect", [
    ("kpm_sqw.py", ["16"], "dynamical_structure_factor(:kpm"),
    ("lanczos_sqw.py", [], "dynamical_structure_factor(:lanczos"),
    ("time_evolution.py", [], "worst fidelity"),
    ("time_evolution.py", ["15", "device"], "worst fidelity"),
])
def test_reference_example_runs_through_the_mirror(script, args, expect):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + args, cwd=ROOT, capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert expect in r.stdout
