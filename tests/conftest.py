import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from tools import gen_scenes  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")
SCENE_EXT = {"spheres_txt": "txt", "spheres_txt_dof": "txt", "spheres_txt_fisheye": "txt"}
SMALL_SCENES = ["config1", "config1_rngfree", "materials", "mesh1k", "mesh1k_gi", "spheres_txt", "spheres_txt_dof",
                "spheres_txt_fisheye", "readme_small", "obj_quirks"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scene_dir(tmp_path_factory):
    """Scene files regenerated from tools/gen_scenes.py (deterministic); nothing is read from the reference."""
    return str(tmp_path_factory.mktemp("scenes"))


_made = set()


@pytest.fixture(scope="session")
def scene_path(scene_dir):
    def get(name):
        if name not in _made:
            gen_scenes.SCENES[name](scene_dir)
            _made.add(name)
        return os.path.join(scene_dir, f"{name}.{SCENE_EXT.get(name, 'sdml')}")
    return get


@pytest.fixture(scope="session")
def golden():
    def get(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return get


def psnr(a, b, peak=1.0):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)
