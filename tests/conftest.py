import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "dev: recorded experiment served by libqlinear_hip_dev.so (still runs under -m gpu)")


import pytest  # noqa: E402


@pytest.fixture
def exact_dequant_policy(monkeypatch):
    """Kernel-equivalence tests of the exact-dequant arithmetic (4x4x4-MFMA rows kernel, gated / fused variants, MLP pair) compare
    against what the MODULE launches: pin the package policy to "never strict" for them (the default policy takes the
    reference's per-weight rounding for bf16, chatglm_q_amd/_lib.py::strict_for)."""
    from chatglm_q_amd import _lib
    monkeypatch.setattr(_lib, "STRICT_MODE", "off")
