"""include/alfd/dealii_adapter.hpp compiled against a mock of the deal.II
classes (deal.II is absent) and driven like immersed_laplace.cc's solve()."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DEMO = os.path.join(HERE, "adapter", "adapter_demo")


@pytest.fixture(scope="module")
def demo(built):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "adapter")])
    return DEMO


def test_adapter_compiles_and_fails_loudly_without_gpu(demo):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([demo], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr)      # alfd_create -> ALFD_E_HIP -> Error
    assert "alfd_create" in p.stderr


@pytest.mark.gpu
def test_adapter_solve_matches_golden(demo):
    p = subprocess.run([demo], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    gold = json.load(open(os.path.join(HERE, "golden", "solves.json")))["laplace2d_circle"]
    first = p.stdout.splitlines()[0]
    assert f"outer={gold['outer_iterations']} inner={gold['inner_iterations']} " in first, p.stdout
