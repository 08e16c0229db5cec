import importlib
import os
import sys

import numpy as np
import pytest

# torch first: its wheel carries its own HIP runtime (same soname as the system's), and a
# process that loads libptx.so before torch ends up with two runtimes side by side; loaded in
# this order libptx.so binds to the runtime torch brought, and the tensors the tests hand to the
# library belong to the runtime that launches the kernels.
import torch  # noqa: F401,E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

GOLDEN = os.path.join(REPO, "tests", "golden")
SCENES = ["cornell", "example", "example_fixed", "cornell_buster", "superhot_colored", "buster_closeup"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def beq(a, b):
    """bit-for-bit equality of float32 arrays (NaN equals NaN, -0 differs from +0)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def pt():
    """the product package (directory name carries a hyphen, so import by string)"""
    return importlib.import_module("path-tracer_amd")


@pytest.fixture(scope="session")
def port():
    from oracle_lib import Port
    return Port()


@pytest.fixture(scope="session")
def sdf_dir(pt):
    return pt.SDF_DIR


class _SceneCache:
    """host scenes, oracle scenes and device scenes are expensive enough to share"""

    def __init__(self, pt, port):
        self.pt, self.port = pt, port
        self.host, self.oracle, self.device = {}, {}, {}

    def host_scene(self, name):
        if name not in self.host:
            self.host[name] = self.pt.HostScene(f"{self.pt.SDF_DIR}/{name}.sdf")
        return self.host[name]

    def oracle_scene(self, name):
        if name not in self.oracle:
            self.oracle[name] = self.port.load(f"{self.pt.SDF_DIR}/{name}.sdf")
        return self.oracle[name]

    def device_scene(self, name):
        if name not in self.device:
            self.device[name] = self.pt.DeviceScene(self.host_scene(name), 0, 1 << 30)
        return self.device[name]


@pytest.fixture(scope="session")
def scenes(pt, port):
    return _SceneCache(pt, port)
