"""Known-answer tests that pin the CPU oracle without MJX (parity unpinned by the reference).

Each test is independent of the oracle's own algorithm: closed forms, conservation laws, finite
differences of quantities the oracle does not compute the same way."""
import os

import numpy as np
import pytest

from ambersim_amd.mjcf import MjModel, mass_matrix_qpos0
from oracle.oracle import Oracle

from conftest import GOLDEN

G = 9.81


def test_pendulum_closed_form(mj_models):
    """qacc=(clip(u,-2,2)-m g l sin(theta))/(Iyy+m l^2), semi-implicit Euler; from the reference's pendulum.xml:3-23
    (M=0.087959+1*0.5^2=0.337959, m g l=4.905).  theta=0 hangs down (reference examples/interactive_simulation.py:21)."""
    mj = mj_models["pendulum"]
    z = np.load(os.path.join(GOLDEN, "rollouts.npz"))
    for key in ("pendulum", "pendulum_ctrl"):
        x0, us = z[key + "_x0"], z[key + "_us"]
        xs = Oracle(mj, np.float64).rollout(x0, us)
        th, om = x0[:, 0].copy(), x0[:, 1].copy()
        for t in range(us.shape[1]):
            acc = (np.clip(us[:, t, 0], -2, 2) - 4.905 * np.sin(th)) / 0.337959
            om = om + 0.02 * acc
            th = th + 0.02 * om
            assert np.allclose(xs[:, t + 1, 0], th, rtol=0, atol=1e-11)
            assert np.allclose(xs[:, t + 1, 1], om, rtol=0, atol=1e-11)
        xs32 = Oracle(mj, np.float32).rollout(x0, us)
        assert np.abs(xs32 - xs).max() / np.abs(xs).max() < 1e-4


CHAIN = """
<mujoco>
  <compiler angle="radian"/>
  <option timestep="{dt}" gravity="0 0 {g}" iterations="20" ls_iterations="20"><flag {flags}/></option>
  <worldbody>
    <body name="a" pos="0 0 2">
      {root_joint}
      <inertial pos="0.05 0.02 -0.2" quat="0.9 0.1 0.3 0.2" mass="1.3" diaginertia="0.03 0.02 0.012"/>
      <body name="b" pos="0.1 0 -0.4" quat="0.8 0.2 0.1 0.5">
        <joint name="j1" type="hinge" axis="0 1 0" pos="0.01 0.02 0.03"/>
        <joint name="j1b" type="hinge" axis="1 0 0"/>
        <inertial pos="0 0.03 -0.15" quat="0.7 0.3 0.1 0.6" mass="0.7" diaginertia="0.01 0.008 0.004"/>
        <body name="c" pos="0 0.05 -0.3">
          <joint name="j2" type="slide" axis="0.3 0.1 1"/>
          <inertial pos="0.02 0 -0.1" mass="0.4" diaginertia="0.003 0.003 0.001"/>
        </body>
      </body>
      <body name="d" pos="-0.1 0.1 0.1">
        <joint name="j3" type="ball" pos="0 0 0.02"/>
        <inertial pos="0.1 0 0" quat="0.5 0.5 0.5 0.5" mass="0.5" diaginertia="0.006 0.004 0.002"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


def chain_model(root="free", dt=1e-4, g=-9.81, flags='contact="disable"'):
    rj = {"free": '<freejoint name="root"/>', "hinge": '<joint name="root" type="hinge" axis="0 0 1"/>'}[root]
    return MjModel.from_xml_string(CHAIN.format(dt=dt, g=g, flags=flags, root_joint=rj))


def random_state(mj, rng):
    q = mj.qpos0.copy()
    v = rng.standard_normal(mj.nv)
    for j in range(mj.njnt):
        a = mj.jnt_qposadr[j]
        t = mj.jnt_type[j]
        if t == 0:
            q[a:a + 3] += 0.2 * rng.standard_normal(3)
            qq = rng.standard_normal(4); q[a + 3:a + 7] = qq / np.linalg.norm(qq)
        elif t == 1:
            qq = rng.standard_normal(4); q[a:a + 4] = qq / np.linalg.norm(qq)
        else:
            q[a] += 0.5 * rng.standard_normal()
    return q, v


def energies(mj, o, q, v):
    o.set_state(q, v, np.zeros(mj.nu)); o.forward()
    M = o.get("qM", (mj.nv, mj.nv))
    xipos = o.get("xipos", (mj.nbody, 3))
    return 0.5 * v @ M @ v, -(mj.body_mass * (xipos @ mj.opt.gravity)).sum()


def test_mass_matrix_symmetric_pd_and_matches_host_crba():
    """CRBA in the oracle (com-frame spatial algebra) vs the host compiler's independent world-frame CRBA at qpos0."""
    mj = chain_model()
    o = Oracle(mj)
    o.set_state(mj.qpos0, np.zeros(mj.nv), np.zeros(0)); o.forward()
    M = o.get("qM", (mj.nv, mj.nv))
    assert np.allclose(M, M.T, atol=1e-13) and np.linalg.eigvalsh(M).min() > 0
    assert np.allclose(M, mass_matrix_qpos0(mj)[0], atol=1e-12)


def test_kinetic_energy_from_finite_differences():
    """0.5 v'Mv must equal sum_b 0.5 m |d xipos/dt|^2 + 0.5 w'Iw with body velocities from finite-differenced
    kinematics along the integration manifold (independent of CRBA and of cvel)."""
    mj = chain_model(dt=1e-6, g=0.0)
    rng = np.random.default_rng(5)
    for _ in range(3):
        q, v = random_state(mj, rng)
        o = Oracle(mj)
        T, _ = energies(mj, o, q, v)

        def poses(sign):
            oo = Oracle(mj)
            oo.set_state(q, sign * v, np.zeros(0), np.zeros(mj.nv))
            # one Euler step of length dt with zero acceleration: qpos moves by +-dt*v on the manifold
            m2 = chain_model(dt=1e-6, g=0.0, flags='contact="disable" gravity="disable"')
            o2 = Oracle(m2); o2.set_state(q, sign * v, np.zeros(0)); o2.step()
            qn = o2.get("qpos")
            o2.set_state(qn, np.zeros(mj.nv), np.zeros(0)); o2.forward()
            return o2.get("xipos", (mj.nbody, 3)), o2.get("xmat", (mj.nbody, 3, 3))
        # centrifugal terms make qvel change over the step, but positions move by dt*(v + O(dt)): error O(dt)
        xp, Rp = poses(+1)
        xm, Rm = poses(-1)
        o.set_state(q, v, np.zeros(0)); o.forward()
        R0 = o.get("xmat", (mj.nbody, 3, 3))
        T_fd = 0.0
        for b in range(1, mj.nbody):
            vel = (xp[b] - xm[b]) / 2e-6
            W = (Rp[b] - Rm[b]) / 2e-6 @ R0[b].T  # skew(w)
            w = np.array([W[2, 1], W[0, 2], W[1, 0]])
            iq = mj.body_iquat[b]
            from ambersim_amd.mjcf import _quat_to_mat
            Ri = R0[b] @ _quat_to_mat(iq)
            Iw = Ri @ np.diag(mj.body_inertia[b]) @ Ri.T
            T_fd += 0.5 * mj.body_mass[b] * vel @ vel + 0.5 * w @ Iw @ w
        assert abs(T - T_fd) / T < 1e-4


def test_gravity_bias_is_potential_gradient():
    """qfrc_bias(q, v=0) = dV/dq with V = -sum m g.xipos, by central differences on hinge/slide coordinates."""
    mj = chain_model(root="hinge")
    rng = np.random.default_rng(7)
    q, _ = random_state(mj, rng)
    o = Oracle(mj)
    o.set_state(q, np.zeros(mj.nv), np.zeros(0)); o.forward()
    bias = o.get("qfrc_bias")
    for j in range(mj.njnt):
        if mj.jnt_type[j] < 2:
            continue
        a, d = mj.jnt_qposadr[j], mj.jnt_dofadr[j]
        e = np.zeros(mj.nq); e[a] = 1e-6
        Vp = energies(mj, o, q + e, np.zeros(mj.nv))[1]
        Vm = energies(mj, o, q - e, np.zeros(mj.nv))[1]
        assert abs(bias[d] - (Vp - Vm) / 2e-6) < 1e-6 * max(1, abs(bias[d]))


def test_energy_conservation_free_chain():
    """No damping, no contact, no actuation: total energy drifts only at O(dt) of the semi-implicit integrator."""
    mj = chain_model(dt=2e-5)
    rng = np.random.default_rng(11)
    q, v = random_state(mj, rng)
    o = Oracle(mj)
    T0, V0 = energies(mj, o, q, v)
    xs = o.rollout(np.concatenate([q, v]), np.zeros((1, 2000, 0)))
    T1, V1 = energies(mj, o, xs[0, -1, :mj.nq], xs[0, -1, mj.nq:])
    assert abs((T1 + V1) - (T0 + V0)) / abs(T0 + abs(V0)) < 2e-3


def test_free_fall_and_momentum():
    """Free-floating tree: the COM follows the semi-implicit free-fall closed form (g enters exactly: the residual is
    identical with g=0) and the residual is the integrator's O(dt) error: it halves when dt halves."""
    errs = {}
    for dt, N in ((1e-3, 200), (5e-4, 400)):
        for g in (-9.81, 0.0):
            mj = chain_model(dt=dt, g=g)
            rng = np.random.default_rng(13)
            q, v = random_state(mj, rng)
            o = Oracle(mj)

            def com(state):
                o.set_state(state[:mj.nq], state[mj.nq:], np.zeros(0)); o.forward()
                return o.get("subtree_com", (mj.nbody, 3))[1]
            xs = o.rollout(np.concatenate([q, v]), np.zeros((1, N, 0)))[0]
            c0, c1, cN = com(xs[0]), com(xs[1]), com(xs[N])
            vcom1 = (c1 - c0) / dt  # COM velocity after the first step (already includes dt*g)
            pred = c1 + (N - 1) * dt * vcom1 + np.array([0, 0, g]) * dt * dt * (N - 1) * N / 2
            errs[(dt, g)] = np.abs(cN - pred).max()
    assert abs(errs[(1e-3, -9.81)] - errs[(1e-3, 0.0)]) < 1e-12
    assert errs[(1e-3, 0.0)] < 2e-5
    assert 1.8 < errs[(1e-3, 0.0)] / errs[(5e-4, 0.0)] < 2.2


SPHERE = """
<mujoco>
  <option timestep="0.002" iterations="50" ls_iterations="50"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 0.1"/>
    <body name="ball" pos="0 0 0.101">
      <freejoint/>
      <geom name="s" type="{gtype}" size="{size}" density="1000"/>
    </body>
  </worldbody>
</mujoco>
"""


@pytest.mark.parametrize("gtype,size,ncon", [("sphere", "0.1", 1), ("capsule", "0.1 0.2", 2), ("box", "0.1 0.1 0.1", 4)])
def test_resting_contact_force_equals_weight(gtype, size, ncon):
    """A body at rest on the plane: pyramid edge forces sum to m g, penetration stays sub-millimetre."""
    xml = SPHERE.format(gtype=gtype, size=size)
    if gtype == "capsule":
        xml = xml.replace('pos="0 0 0.101"', 'pos="0 0 0.101" quat="0.7071068 0 0.7071068 0"')
    mj = MjModel.from_xml_string(xml)
    assert mj.ncon == ncon
    o = Oracle(mj)
    x0 = np.concatenate([mj.qpos0, np.zeros(6)])
    xs = o.rollout(x0, np.zeros((1, 1500, 0)))[0]
    # at rest; the box may chatter at the 1e-3 rad/s level: with exact symmetry the reference's manifold heuristic ties
    # between a corner and the first point and can return a 3-point manifold
    assert np.abs(xs[-1, mj.nq:]).max() < (5e-3 if gtype == "box" else 1e-3)
    weight = mj.body_mass.sum() * G
    forces = []
    for t in (-2, -1):
        o.set_state(xs[t, :7], xs[t, 7:], np.zeros(0)); o.forward()
        forces.append(o.get("efc_force").sum())
    assert abs(np.mean(forces) - weight) / weight < 5e-3
    dist = o.get("con_dist")
    active = dist < 0
    assert active.sum() >= min(ncon, 3) and (dist[active] > -1e-3).all()
    frame = o.get("con_frame", (ncon, 3, 3))
    for f in frame:
        assert np.allclose(f @ f.T, np.eye(3), atol=1e-12) and np.allclose(f[0], [0, 0, 1])


def test_plane_convex_manifold():
    """Flat convex sole on the plane (the reference physics' manifold heuristic): a = first deepest
    vertex, b = farthest from a, c = farthest from line ab, d = farthest from the other two edges.  On an exactly
    symmetric rectangle step d ties between the fourth corner and a; a duplicate index is reported inactive (dist = 1).
    On an asymmetric sole all four corners are distinct."""
    mj = MjModel.from_xml_string(SPHERE.format(gtype="box", size="0.2 0.1 0.05").replace("0 0 0.101", "0 0 0.0495"))
    o = Oracle(mj)
    o.set_state(mj.qpos0, np.zeros(6), np.zeros(0)); o.forward()
    dist, pos = o.get("con_dist"), o.get("con_pos", (4, 3))
    act = dist < 0
    assert act.sum() >= 3 and np.allclose(dist[act], -0.0005, atol=1e-12) and np.all(dist[~act] == 1.0)
    corners = {(round(p[0], 6), round(p[1], 6)) for p in pos[act]}
    assert len(corners) == act.sum() and corners <= {(0.2, 0.1), (0.2, -0.1), (-0.2, 0.1), (-0.2, -0.1)}
    trapezoid = """
    <mujoco><option timestep="0.002"/>
      <asset><mesh name="sole" vertex="0.2 0.1 0  0.2 -0.1 0  -0.2 0.06 0  -0.2 -0.06 0  0.1 0.05 0.05  0.1 -0.05 0.05  -0.1 0.03 0.05  -0.1 -0.03 0.05"/></asset>
      <worldbody><geom type="plane" size="0 0 1"/>
        <body pos="0 0 -0.0005"><freejoint/><geom type="mesh" mesh="sole"/></body></worldbody></mujoco>"""
    mj = MjModel.from_xml_string(trapezoid)
    o = Oracle(mj)
    o.set_state(mj.qpos0, np.zeros(6), np.zeros(0)); o.forward()
    dist, pos = o.get("con_dist"), o.get("con_pos", (4, 3))
    assert np.allclose(dist, -0.0005, atol=1e-12)
    assert {(round(p[0], 6), round(p[1], 6)) for p in pos} == {(0.2, 0.1), (0.2, -0.1), (-0.2, 0.06), (-0.2, -0.06)}
    assert np.allclose(pos[:, 2], -0.00025, atol=1e-12)  # midway between the surfaces


LIMITED = """
<mujoco>
  <compiler angle="radian"/>
  <option timestep="0.002" iterations="30" ls_iterations="30"/>
  <worldbody>
    <body pos="0 0 1">
      <joint name="a" type="hinge" axis="0 1 0" range="-0.5 0.5"/>
      <inertial pos="0 0 -0.5" mass="1" diaginertia="0.05 0.05 0.01"/>
      <body pos="0 0 -1">
        <joint name="b" type="hinge" axis="0 1 0" damping="0.05"/>
        <inertial pos="0 0 -0.3" mass="0.5" diaginertia="0.02 0.02 0.005"/>
      </body>
    </body>
  </worldbody>
  <equality><joint joint1="b" joint2="a" polycoef="0.1 0.5 0 0 0"/></equality>
  <actuator><motor joint="a" ctrlrange="-20 20"/></actuator>
</mujoco>
"""


def test_joint_limit_and_equality_hold():
    """Driven hard into its range the hinge stops near the limit, and the polycoef equality b = 0.1 + 0.5 a tracks."""
    mj = MjModel.from_xml_string(LIMITED)
    assert mj.jnt_limited.tolist() == [1, 0] and mj.neq == 1
    o = Oracle(mj)
    xs = o.rollout(np.zeros(4), np.full((1, 3000, 1), 15.0))[0]
    a, b = xs[-1, 0], xs[-1, 1]
    assert 0.5 <= a < 0.56, a
    assert abs(b - (0.1 + 0.5 * a)) < 0.02
    assert np.abs(xs[-1, 2:]).max() < 1e-2


def test_newton_and_cg_agree_when_converged(mj_models):
    """Both solvers minimise the same convex cost: converged accelerations agree (quadruped on the ground)."""
    import copy
    base = mj_models["quadruped"]
    x = None
    accs = []
    for solver in (2, 1):
        mj = copy.copy(base)
        mj.opt = base.opt.replace(solver=solver, iterations=200, ls_iterations=50, tolerance=1e-12)
        o = Oracle(mj)
        if x is None:
            x = o.rollout(np.concatenate([mj.key_qpos[0], np.zeros(mj.nv)]), np.tile(mj.key_ctrl[0], (1, 200, 1)))[0, -1]
        o.set_state(x[:mj.nq], x[mj.nq:], mj.key_ctrl[0]); o.forward()
        assert o.get("nefc")[0] >= 16
        accs.append(o.get("qacc"))
    assert np.abs(accs[0] - accs[1]).max() < 1e-5 * max(1.0, np.abs(accs[0]).max())


def test_mesh_foot_contact_jacobian_is_the_gradient_of_the_distance():
    """The bench's headline model (r05): two stages of the oracle that share no code - the plane / convex-mesh collision (A9.5b: which
    sole vertices touch, how deep) and the constraint rows' Jacobian (A9.6: frame, lever arms, the chain of dofs behind each foot) - must
    agree through calculus: along any generalised velocity v the contact distance changes at the rate J_n v, with J_n the normal row (the
    mean of a pyramid row pair, (J_n + mu J_t) and (J_n - mu J_t)).  Central differences of the distances under q (+) eps v (quaternion
    integrated), soles pressed 21 mm into the floor with the joints off their keyframe, five random directions: agreement to 1e-8."""
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    mj = load_mj_model_from_file("models/biped/biped_stand.xml")
    o = Oracle(mj, np.float64)
    rng = np.random.default_rng(0)
    q0 = mj.key_qpos[0].copy()
    q0[2] -= 0.021
    q0[7:] += 0.02 * rng.standard_normal(mj.nq - 7)
    z = lambda n: np.zeros(n)

    def moved(q, v, eps):
        q = q.copy()
        q[:3] += eps * v[:3]
        w = v[3:6]
        ang = eps * np.linalg.norm(w)
        ax = w / np.linalg.norm(w)
        w1, x1, y1, z1 = np.cos(ang / 2), *(np.sin(ang / 2) * ax)
        w0, x0, y0, z0 = q[3:7]
        q[3:7] = [w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1, w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1, w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1]
        q[7:] += eps * v[6:]
        return q

    o.set_state(q0, z(mj.nv), z(mj.nu)); o.forward()
    d0 = o.get("con_dist").copy()
    act = np.nonzero(d0 < 0)[0]
    assert len(act) >= 3 and {c // 4 for c in act} == {0, 1}  # both feet touch
    nefc = int(o.get("nefc")[0])
    Jc = o.get("efc_J").reshape(nefc, mj.nv)[-4 * len(act):]
    Jn = 0.5 * (Jc[0::4] + Jc[1::4])
    for _ in range(5):
        v, eps, d = rng.standard_normal(mj.nv), 1e-6, []
        for sgn in (1, -1):
            o.set_state(moved(q0, v, sgn * eps), z(mj.nv), z(mj.nu)); o.forward()
            d.append(o.get("con_dist").copy())
        assert (np.sign(d[0][act]) == np.sign(d0[act])).all()  # the same corners stay selected
        fd, an = (d[0] - d[1])[act] / (2 * eps), Jn @ v
        assert np.abs(fd - an).max() < 1e-8 * max(1.0, np.abs(an).max()), (fd, an)
