import os
import sys

import pytest

# the suite reads the persisted tile table but never writes its own (tiny) geometries into it (cstp_amd.ops)
os.environ.setdefault("CSTP_TUNE_TABLE_RO", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, b):
    """max-abs-diff / max-abs-ref (logits cross zero, so not element-wise relative)."""
    import torch
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
