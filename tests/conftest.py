import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The tests name the kernels they mean: put_model(mj, dev) = the library's own (generic unless the model is shipped),
# put_model(..., specialise=True) = a per-model build.  The package's default ("auto": per-model builds whenever hipcc is there) would
# compile every fuzz draw.
os.environ.setdefault("AMX_SPECIALISE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The CPU oracle is test infrastructure: build it once per session (gcc, a few seconds)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


MODELS = {
    "pendulum": "models/pendulum/scene.xml",
    "quadruped": "models/quadruped/quadruped.xml",
    "biped": "models/biped/biped.xml",
}


@pytest.fixture(scope="session")
def mj_models():
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    return {k: load_mj_model_from_file(v) for k, v in MODELS.items()}


GOLDEN = os.path.join(ROOT, "tests", "golden")
