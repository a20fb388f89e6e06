"""URDF branch of the load API and force_float (reference tests/test_model_io.py:57-146)."""
import os
import re
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np
import pytest

from ambersim_amd.mjcf import DisableBit, MjData, MjModel, recompile_options
from ambersim_amd.urdf import parse_urdf, urdf_to_mjcf
from ambersim_amd.utils.conversion_utils import save_model_xml
from ambersim_amd.utils.io_utils import _modify_robot_float_base, load_mj_model_from_file

from conftest import GOLDEN, ROOT

REF_MODELS = "/root/reference/ambersim/models"
needs_reference = pytest.mark.skipif(not os.path.exists(REF_MODELS), reason="reference tree not mounted")
ARM = os.path.join(GOLDEN, "arm3.urdf")

FIELDS = ("body_parentid", "body_pos", "body_quat", "body_mass", "body_ipos", "body_inertia", "jnt_type", "jnt_axis", "jnt_pos", "jnt_range",
          "jnt_limited", "dof_damping", "dof_invweight0", "geom_type", "geom_size", "geom_pos", "geom_contype", "actuator_ctrlrange", "eq_data")


def _assert_same_model(a, b, skip=()):
    for f in FIELDS:
        if f in skip:
            continue
        x, y = np.asarray(getattr(a, f), dtype=np.float64), np.asarray(getattr(b, f), dtype=np.float64)
        assert x.shape == y.shape, f
        assert np.allclose(x, y, rtol=1e-9, atol=1e-12), f
    # quaternions up to sign
    for f in ("body_quat", "geom_quat"):
        x, y = getattr(a, f), getattr(b, f)
        assert np.allclose(np.abs(np.sum(x * y, axis=1)), 1, atol=1e-12), f


def test_unfused_urdf_equals_handwritten_mjcf(tmp_path):
    """The converter's body/joint/geom/inertial mapping against an MJCF written by hand from the same URDF numbers."""
    text = Path(ARM).read_text().replace('fusestatic="true"', 'fusestatic="false"')
    p = tmp_path / "arm3.urdf"
    p.write_text(text)
    u = load_mj_model_from_file(str(p))
    x = load_mj_model_from_file(os.path.join(GOLDEN, "arm3_unfused.xml"))
    assert (u.nq, u.nv, u.nu, u.neq, u.nbody, u.ngeom) == (4, 4, 3, 1, 7, 7)
    assert u.names_body == x.names_body and u.names_joint == x.names_joint
    assert u.names_actuator == x.names_actuator == ["shoulder_actuator", "elbow_actuator", "grip_a_actuator"]
    assert u.names_eq == x.names_eq == ["grip_b_grip_a_equality"]
    _assert_same_model(u, x)
    # principal frames can differ by an axis permutation-free sign: compare the inertia tensors they represent
    from ambersim_amd.mjcf import _quat_to_mat
    for b in range(1, u.nbody):
        Iu = _quat_to_mat(u.body_iquat[b]) @ np.diag(u.body_inertia[b]) @ _quat_to_mat(u.body_iquat[b]).T
        Ix = _quat_to_mat(x.body_iquat[b]) @ np.diag(x.body_inertia[b]) @ _quat_to_mat(x.body_iquat[b]).T
        assert np.allclose(Iu, Ix, atol=1e-12)


def test_fusestatic_preserves_dynamics():
    """Fusing the fixed pedestal into the world and the fixed tool into the forearm changes the body list, not the
    physics: same mass matrix at qpos0 and the same oracle rollout as the unfused model."""
    from ambersim_amd.mjcf import mass_matrix_qpos0
    from oracle.oracle import Oracle
    fused = load_mj_model_from_file(ARM)
    unfused = load_mj_model_from_file(os.path.join(GOLDEN, "arm3_unfused.xml"))
    assert fused.nbody == unfused.nbody - 2 and "tool" not in fused.names_body and "pedestal" not in fused.names_body
    assert fused.ngeom == unfused.ngeom  # geoms move to the parent, none is dropped
    assert fused.names_joint == unfused.names_joint
    assert abs(fused.body_mass.sum() - (unfused.body_mass.sum() - 5.0)) < 1e-12  # the world keeps no inertia
    assert np.allclose(mass_matrix_qpos0(fused)[0], mass_matrix_qpos0(unfused)[0], atol=1e-12)
    rng = np.random.default_rng(3)
    x0 = np.concatenate([[0.3, -0.4, 0.01, 0.01], 0.1 * rng.standard_normal(4)])[None]
    us = rng.uniform(-1, 1, size=(1, 50, 3)) * np.array([5.0, 3.0, 1.0])
    outs = []
    for m in (fused, unfused):
        m = recompile_options(m.replace(opt=m.opt.replace(disableflags=int(DisableBit.CONTACT))))
        outs.append(Oracle(m, np.float64).rollout(x0, us))
    assert np.allclose(outs[0], outs[1], rtol=1e-8, atol=1e-10)


def test_urdf_details(tmp_path):
    u = load_mj_model_from_file(ARM)
    assert u.jnt_limited.tolist() == [1, 1, 1, 1] and np.allclose(u.jnt_range[2], [0, 0.03])
    assert np.allclose(u.dof_damping, [0.5, 0.2, 1.0, 1.0])
    vis = [i for i, n in enumerate(u.names_geom) if n == "upper_vis"]
    assert len(vis) == 1 and u.geom_contype[vis[0]] == 0 and u.geom_conaffinity[vis[0]] == 0
    # discardvisual defaults to true, fusestatic to true, continuous joints are unlimited, effort-less joints get unlimited motors
    text = Path(ARM).read_text()
    text = re.sub(r"<mujoco>.*?</mujoco>", "", text, flags=re.S).replace('type="revolute"', 'type="continuous"').replace(' effort="40"', "")
    p = tmp_path / "arm3b.urdf"
    p.write_text(text)
    v = load_mj_model_from_file(str(p))
    assert "upper_vis" not in v.names_geom and v.nbody == u.nbody
    assert v.jnt_limited.tolist() == [0, 0, 1, 1]
    assert v.actuator_ctrllimited.tolist() == [0, 1, 1]
    # plain MuJoCo import (no transmissions / mimics patched in): what mujoco.MjModel.from_xml_path(urdf) gives
    raw = MjModel.from_xml_path(ARM)
    assert raw.nu == 0 and raw.neq == 0 and raw.nq == 4
    with pytest.raises(NotImplementedError):
        load_mj_model_from_file(str(tmp_path / "x.sdf")) if (tmp_path / "x.sdf").write_text("<sdf/>") else None


def test_force_float_cases(tmp_path):
    """reference tests/test_model_io.py:107-146: (1) first body has no joint -> freejoint added to it, (2) inline XML
    whose first body has no joint, (3) a URDF model."""
    m1 = _modify_robot_float_base(os.path.join(ROOT, "ambersim_amd/models/pendulum/pendulum.xml"))
    assert "freejoint" in "\t".join(m1.names_joint) and m1.nq == 8
    dummy = """
    <mujoco model="parent">
      <worldbody>
        <body>
          <inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/>
          <geom name="foo" type="box" pos="-0.2 0 0.3" size="0.5 0.3 0.1"/>
          <body name="child" pos="1. 2. 3." quat="1. 0. 0. 1.">
            <inertial pos="0 0 0" mass="2" diaginertia="1 1 1"/>
            <geom name="my_box" type="box" pos="0.5 0.25 1." size="0.1 0.2 0.3"/>
          </body>
        </body>
      </worldbody>
    </mujoco>
    """
    p = tmp_path / "_temp.xml"
    p.write_text(dummy)
    m2 = _modify_robot_float_base(str(p))
    assert "freejoint" in "\t".join(m2.names_joint) and m2.nq == 7
    # a body that already has a joint is left alone
    q = load_mj_model_from_file("models/quadruped/quadruped.xml", force_float=True)
    assert q.nq == 19
    unfree = load_mj_model_from_file(ARM.replace("arm3.urdf", "arm3_unfused.xml"))
    free = load_mj_model_from_file(ARM.replace("arm3.urdf", "arm3_unfused.xml"), force_float=True)
    assert len(MjData(unfree).qpos) == 4 and len(MjData(free).qpos) == 11


def test_save_model_xml_roundtrip(tmp_path):
    """reference tests/test_model_io.py:57-62: a URDF saved as XML loads again (and is the same model)."""
    out = tmp_path / "arm3_saved.xml"
    save_model_xml(ARM, output_path=str(out))
    a, b = MjModel.from_xml_path(ARM), MjModel.from_xml_path(str(out))
    _assert_same_model(a, b)


@needs_reference
def test_reference_urdfs_actuators_mimics_and_float():
    """reference tests/test_model_io.py:65-104,133-146 on the reference's own URDFs."""
    for urdf in Path(REF_MODELS).rglob("*.urdf"):
        tree = parse_urdf(urdf)
        m = load_mj_model_from_file(str(urdf))
        transmissions = tree.findall("transmission")
        assert m.nu == len(transmissions)
        assert sorted(n.replace("_actuator", "") for n in m.names_actuator) == sorted(t.find("joint").get("name") for t in transmissions)
        mimics = [j for j in tree.findall("joint") if j.find("mimic") is not None]
        assert m.neq == len(mimics)
        for j in mimics:
            assert f"{j.get('name')}_{j.find('mimic').get('joint')}_equality" in m.names_eq
    unfree = load_mj_model_from_file(REF_MODELS + "/barrett_hand/bh280.urdf", force_float=False)
    assert len(MjData(unfree).qpos) == 8
    free = load_mj_model_from_file(REF_MODELS + "/barrett_hand/bh280.urdf", force_float=True)
    assert len(MjData(free).qpos) == 15


@needs_reference
def test_reference_urdf_matches_reference_xml():
    """The reference ships bh280.xml, MuJoCo's own conversion of bh280.urdf: our conversion must give the same model."""
    u = load_mj_model_from_file(REF_MODELS + "/barrett_hand/bh280.urdf")
    x = load_mj_model_from_file(REF_MODELS + "/barrett_hand/bh280.xml")
    assert u.names_body == x.names_body and u.names_joint == x.names_joint and u.names_actuator == x.names_actuator and u.names_eq == x.names_eq
    for f in FIELDS:
        a, b = np.asarray(getattr(u, f), dtype=np.float64), np.asarray(getattr(x, f), dtype=np.float64)
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-6, atol=1e-8), f
    p = load_mj_model_from_file(REF_MODELS + "/pendulum/pendulum.urdf")
    assert (p.nq, p.nu, p.neq, p.nbody) == (1, 1, 0, 3) and np.allclose(p.jnt_range, [[-3.1416, 3.1416]])
    assert abs(p.stat.meaninertia - 0.337959) < 1e-12
