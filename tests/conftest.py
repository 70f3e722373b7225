import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_images():
    return np.load(os.path.join(GOLDEN, "images.npz"))


@pytest.fixture(scope="session")
def golden_traces():
    return np.load(os.path.join(GOLDEN, "traces.npz"))


@pytest.fixture(scope="session")
def scenes():
    from raytracer_amd.scene import FlatScene
    names = ["demo_onoff", "demo_path", "csg_example", "demo_path_clock30"]
    return {n: FlatScene.from_json(os.path.join(GOLDEN, f"{n}.scene.json")) for n in names}
