"""pytest configuration: the `gpu` marker, and loaders for the product binding and the oracle."""
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_product():
    """Import slam-cin0051_amd/ (hyphenated directory) as module slam_cin0051_amd."""
    name = "slam_cin0051_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "slam-cin0051_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def slamx():
    return load_product()


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding

    return oracle_binding.load()


@pytest.fixture(scope="session")
def ref_data():
    return os.path.join(ROOT, "tests", "golden", "ref_data")


def load_gray(path):
    from PIL import Image

    im = Image.open(path)
    assert im.mode == "L", "fixtures are 8-bit gray PNGs (SURVEY.md section 4)"
    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))


@pytest.fixture(scope="session")
def gpu_ctx(slamx):
    return slamx.default_context()
