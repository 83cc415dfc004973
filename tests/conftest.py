"""Shared test plumbing: markers, paths, oracle loader (tests are the oracle's only importer)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_oracle():
    """Import oracle/dots_oracle.py (test infrastructure) without putting oracle/ on the product path."""
    name = "dots_oracle"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "oracle", "dots_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


# torch's bundled HIP runtime must be the one the process binds (dots_socp_amd/_lib.py: _torch_runtime_first): load and
# initialise torch before the product library, whatever test file runs first.
if has_gpu():
    import torch

    torch.cuda.init()


@pytest.fixture(scope="session")
def oracle():
    return load_oracle()
