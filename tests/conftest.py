import os
import sys

import pytest

# The oracle is OpenMP code; a GPU box exposes every host thread but grants a 16-CPU share, and
# oversubscribed spinning OpenMP teams are pathologically slow.
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def cornell_small():
    import clive2_amd as c2
    return c2.create_scene_from_preset("empty", 64, 48)


@pytest.fixture(scope="session")
def glass_scene():
    """Cornell box + subdivision-2 icosphere of material 5 made rough glass (alpha 0.1): exercises
    GGX sampling with alpha > 0, reflection/transmission, smooth normals, a deeper BVH."""
    import numpy as np
    import clive2_amd as c2
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    mats = get_materials()
    mats["alpha"][5] = 0.1
    v, f = icosphere(2, radius=2.0, center=(0.0, 1.0, 0.0))
    return c2.create_scene(64, 48, np.array([0, 1.5, 6]), np.array([0, 0, -1]),
                           file_specs=[dict(mesh=(v, f), material=5)], materials=mats)
