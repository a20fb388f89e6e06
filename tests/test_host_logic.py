"""Host-side API semantics (no GPU): path resolution, solver selection, dataclass-style replace, sharding maths."""
import os
from pathlib import Path

import pytest

import ambersim_amd
from ambersim_amd.dist import shard_range
from ambersim_amd.mjcf import SOL_CG, SOL_NEWTON, DisableBit, MjData
from ambersim_amd.rl.base import State
from ambersim_amd.utils._internal_utils import _check_filepath
from ambersim_amd.utils.io_utils import load_mj_model_from_file


def test_check_filepath_resolution(tmp_path, monkeypatch):
    """absolute, cwd-relative and ROOT-relative paths, str and Path (reference tests/test_model_io.py:23-46)."""
    rel = "models/pendulum/pendulum.xml"
    absolute = ambersim_amd.ROOT / rel
    for p in (str(absolute), absolute, rel, Path(rel)):
        assert os.path.samefile(_check_filepath(p), absolute)
    monkeypatch.chdir(ambersim_amd.ROOT)
    assert os.path.samefile(_check_filepath("models/pendulum/scene.xml"), ambersim_amd.ROOT / "models/pendulum/scene.xml")
    monkeypatch.chdir(tmp_path)
    with pytest.raises(ValueError):  # reference _internal_utils.py:17
        _check_filepath("models/does/not/exist.xml")


def test_load_model_options_and_errors(tmp_path):
    m = load_mj_model_from_file("models/pendulum/scene.xml")
    assert m.opt.solver == SOL_NEWTON  # the reference's default for mujoco>=3.0.1 (io_utils.py:173-174)
    m = load_mj_model_from_file("models/pendulum/scene.xml", solver="cg", iterations=7, ls_iterations=3)
    assert (m.opt.solver, m.opt.iterations, m.opt.ls_iterations) == (SOL_CG, 7, 3)
    assert load_mj_model_from_file("models/pendulum/scene.xml", solver="Newton").opt.solver == SOL_NEWTON
    with pytest.raises(ValueError):  # io_utils.py:181
        load_mj_model_from_file("models/pendulum/scene.xml", solver="pgs")
    bad = tmp_path / "model.txt"
    bad.write_text("x")
    with pytest.raises(NotImplementedError):  # io_utils.py:199
        load_mj_model_from_file(str(bad))
    urdf = tmp_path / "model.urdf"
    urdf.write_text("<robot/>")
    with pytest.raises(ValueError):  # a URDF without links (the URDF branch itself is covered by tests/test_urdf.py)
        load_mj_model_from_file(str(urdf))


def test_disable_bits_match_mujoco():
    assert (DisableBit.CONSTRAINT, DisableBit.CONTACT, DisableBit.GRAVITY, DisableBit.EULERDAMP) == (1, 16, 64, 16384)


def test_state_and_data_replace(mj_models):
    s = State(pipeline_state=None, obs=1, reward=2, done=3)
    s2 = s.replace(obs=5)
    assert (s.obs, s2.obs, s2.reward) == (1, 5, 2) and s2.metrics is s.metrics
    with pytest.raises(AttributeError):
        s.replace(nope=1)
    d = MjData(mj_models["quadruped"])
    assert len(d.qpos) == 19 and len(d.qvel) == 18 and len(d.ctrl) == 12
    o = mj_models["pendulum"].opt
    assert o.replace(timestep=0.5).timestep == 0.5 and o.timestep == 0.02


@pytest.mark.parametrize("n,world", [(4096, 8), (32768, 8), (10, 3), (5, 8), (0, 2)])
def test_shard_range_partitions(n, world):
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1


def test_full_barrett_hand_follows_the_references_flow(monkeypatch):
    """reference tests/trajopt/test_predictive_sampler.py:14-31: load bh280.xml, replace(opt=...DisableBit.CONTACT), shoot.  Host side of
    that flow (the device handle is replaced by a recorder: no GPU here): put_model accepts the full hand although none of its 2000+
    mesh-mesh pairs can run, plans it without contact, hands out its device pointer only once contact is disabled."""
    import os
    import pytest
    path = "/root/reference/ambersim/models/barrett_hand/bh280.xml"
    if not os.path.exists(path):
        pytest.skip("reference not present on this machine")
    import torch
    from ambersim_amd.mjx import _src
    from ambersim_amd.mjcf import DisableBit
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    planned = []

    class FakeHandle:
        def __init__(self, mj, device_index, specialise=False):
            planned.append(mj)
            self.refuse, self._ptr = None, 1234

        def query(self, name):
            return 0
        ptr = _src._Handle.ptr

    monkeypatch.setattr(_src, "_Handle", FakeHandle)
    mj = load_mj_model_from_file(path)
    assert mj.nunsupported_pairs > 2000
    model = _src.put_model(mj, torch.device("cpu"), specialise=False)
    assert (model.nq, model.nu) == (8, 4) and planned[-1].npair == 0 and planned[-1].opt.disableflags & DisableBit.CONTACT and model.opt.disableflags == 0
    with pytest.raises(NotImplementedError, match="collision pairs"):
        model._h.ptr
    off = model.replace(opt=model.opt.replace(timestep=0.002, iterations=1, ls_iterations=4, disableflags=DisableBit.CONTACT))
    assert off._h.ptr == 1234 and planned[-1].npair == 0 and off.opt.timestep == 0.002
