"""Host model compiler: structural pins from the reference's own tests and model files."""
import os

import numpy as np
import pytest

from ambersim_amd.mjcf import DisableBit, MjModel, recompile_options
from ambersim_amd.utils.io_utils import load_mj_model_from_file

from conftest import GOLDEN, ROOT


def test_pendulum_pins(mj_models):
    """nu=1, neq=0 (reference tests/test_model_io.py:76,98 for the pendulum); M=Iyy+m l^2; option line of pendulum.xml:3."""
    for path in ("models/pendulum/pendulum.xml", "models/pendulum/scene.xml"):
        m = load_mj_model_from_file(path)
        assert (m.nq, m.nv, m.nu, m.neq, m.nbody) == (1, 1, 1, 0, 3)
        assert m.names_actuator == ["pendulum_joint_actuator"]
        assert np.allclose(m.actuator_ctrlrange, [[-2, 2]]) and m.actuator_ctrllimited.tolist() == [1]
        assert (m.opt.timestep, m.opt.iterations, m.opt.ls_iterations) == (0.02, 5, 10)
        assert abs(m.stat.meaninertia - 0.337959) < 1e-12 and abs(m.dof_invweight0[0] - 1 / 0.337959) < 1e-9
        assert m.npair == 0  # the floor cannot touch the pendulum: contype=conaffinity=0 on its geoms
    assert load_mj_model_from_file("models/pendulum/scene.xml").ngeom == 3


def test_barrett_pins():
    """Barrett hand: nq=8 (reference tests/test_model_io.py:143), nu=4 and actuator names (:76,82), neq=4 (:98,104)."""
    m = MjModel.from_xml_path(os.path.join(GOLDEN, "bh280_nomesh.xml"))
    assert (m.nq, m.nv, m.nu, m.neq, m.nbody) == (8, 8, 4, 4, 10)
    assert m.names_actuator == [j + "_actuator" for j in ("bh_j32_joint", "bh_j11_joint", "bh_j12_joint", "bh_j22_joint")]
    assert m.names_eq[0] == "bh_j33_joint_bh_j32_joint_equality"
    assert m.jnt_limited.all() and np.allclose(m.jnt_range[:, 0], 0)  # every range starts at 0
    assert (m.opt.timestep, m.opt.iterations, m.opt.ls_iterations, m.opt.solver) == (0.002, 100, 50, 2)  # MuJoCo defaults
    assert np.allclose(m.eq_data[:, 1], [0.3442622950819672, 0.3442622950819672, 1.0, 0.3442622950819672])


@pytest.mark.skipif(not os.path.exists("/root/reference/ambersim/models"), reason="reference tree not mounted")
def test_reference_models_load():
    """Every XML the reference ships loads (reference tests/test_model_io.py:49-54, XML branch, force_float=False)."""
    base = "/root/reference/ambersim/models"
    p = load_mj_model_from_file(base + "/pendulum/scene.xml")
    assert (p.nq, p.nu) == (1, 1) and abs(p.stat.meaninertia - 0.337959) < 1e-12
    h = load_mj_model_from_file(base + "/barrett_hand/bh280.xml")
    assert (h.nq, h.nu, h.neq, h.ngeom) == (8, 4, 4, 98)
    assert h.nmesh == 48 and h.mesh_vert.shape[0] > 1000  # 44 collision + 4 visual meshes, stored as convex hulls
    assert h.nunsupported_pairs == h.npair > 2000  # mesh-mesh pairs: the host model loads; put_model plans it without contact and refuses to step it with contact on
    # geoms never change the dynamics constants here: explicit inertials win
    g = MjModel.from_xml_path(os.path.join(GOLDEN, "bh280_nomesh.xml"))
    assert np.allclose(h.body_mass, g.body_mass) and np.allclose(h.dof_invweight0, g.dof_invweight0)


def test_authored_models(mj_models):
    q, b = mj_models["quadruped"], mj_models["biped"]
    assert (q.nq, q.nv, q.nu, q.nbody, q.npair, q.ncon) == (19, 18, 12, 14, 4, 4)
    assert (b.nq, b.nv, b.nu, b.nbody, b.npair, b.ncon) == (31, 30, 24, 23, 2, 8)
    assert set(q.pair_fn.tolist()) == {0} and set(b.pair_fn.tolist()) == {2}
    assert q.jnt_limited.sum() == 12 and b.jnt_limited.sum() == 24
    for m in (q, b):
        assert (m.dof_invweight0 > 0).all() and (m.body_invweight0[1:] > 0).all() and m.nkey == 1
        assert np.allclose(np.linalg.norm(m.body_iquat, axis=1), 1)
        # free joint: translation and rotation invweights are averaged per triple
        assert np.allclose(m.dof_invweight0[:3], m.dof_invweight0[0]) and np.allclose(m.dof_invweight0[3:6], m.dof_invweight0[3])


def test_disable_contact_drops_pairs(mj_models):
    m = mj_models["quadruped"]
    m2 = recompile_options(m.replace(opt=m.opt.replace(disableflags=DisableBit.CONTACT)))
    assert m2.npair == 0 and m2.ncon == 0 and m.npair == 4


def test_compiler_features():
    xml = """
    <mujoco><compiler angle="degree"/>
      <default><joint damping="0.3"/><default class="big"><geom size="0.2"/></default></default>
      <worldbody>
        <body name="a" pos="0 0 1" euler="0 0 90" childclass="big">
          <joint name="h" type="hinge" axis="0 0 1" range="-90 90"/>
          <geom name="g1" type="sphere" mass="2"/>
          <geom name="g2" type="capsule" fromto="0 0 0 1 0 0" size="0.05" density="500"/>
        </body>
      </worldbody>
      <actuator><position name="p" joint="h" kp="10" ctrlrange="-1 1"/></actuator>
    </mujoco>"""
    m = MjModel.from_xml_string(xml)
    assert np.allclose(m.jnt_range[0], [-np.pi / 2, np.pi / 2]) and m.jnt_limited[0] == 1 and m.dof_damping[0] == 0.3
    assert np.allclose(m.geom_size[0, 0], 0.2)
    cap_mass = 500 * (np.pi * 0.05 ** 2 * 1.0 + 4 / 3 * np.pi * 0.05 ** 3)
    assert abs(m.body_mass[1] - (2 + cap_mass)) < 1e-12
    assert np.allclose(m.geom_pos[1], [0.5, 0, 0]) and abs(m.geom_size[1, 1] - 0.5) < 1e-12
    assert np.allclose(m.actuator_gainprm[0], [10, 0, 0]) and np.allclose(m.actuator_biasprm[0], [0, -10, 0]) and m.actuator_biastype[0] == 1
    assert np.allclose(m.body_quat[1], [np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])


@pytest.mark.parametrize("snippet,exc", [
    ('<option cone="elliptic"/>', NotImplementedError),
    ('<option integrator="RK4"/>', None),  # compiles on the host; rejected when put on the device
])
def test_unsupported_features_raise(snippet, exc):
    """Unsupported physics raises NotImplementedError, the type mjx.device_put raises (reference io_utils.py:228-241)."""
    xml = f"<mujoco>{snippet}</mujoco>"
    if exc is None:
        MjModel.from_xml_string(xml)
    else:
        with pytest.raises(exc):
            MjModel.from_xml_string(xml)


def test_joint_frictionloss_compiles_to_dof_fields():
    """Dry joint friction (MJCF frictionloss / URDF <dynamics friction>) is carried since r03: every dof of the
    joint gets the joint's frictionloss and its solreffriction / solimpfriction; a negative value is refused."""
    from ambersim_amd.mjcf import MjModel
    m = MjModel.from_xml_string("""<mujoco><worldbody><body><joint type="hinge" frictionloss="0.1" solreffriction="0.03 1"/><geom size="0.1"/>
        <body pos="0 0 1"><joint type="ball" frictionloss="0.2"/><geom size="0.1"/></body></body></worldbody></mujoco>""")
    assert m.dof_frictionloss.tolist() == [0.1, 0.2, 0.2, 0.2]
    assert m.dof_solref.tolist() == [[0.03, 1.0]] + [[0.02, 1.0]] * 3 and m.dof_solimp.shape == (4, 5)
    with pytest.raises(ValueError):
        MjModel.from_xml_string('<mujoco><worldbody><body><joint type="hinge" frictionloss="-1"/><geom size="0.1"/></body></worldbody></mujoco>')
