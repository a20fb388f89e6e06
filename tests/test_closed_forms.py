"""Host compiler (ambersim_amd/mjcf.py) against hand-computed values of MuJoCo's documented semantics - quantities the kernel and the
oracle both take from the same descriptor, so an error here would be common-mode and invisible to every GPU-vs-oracle test
("what's weak").  Nothing below is produced by code of this repository: the expected numbers are worked out in the
docstrings."""
import numpy as np
import pytest

from ambersim_amd.mjcf import MjModel


def test_free_box_inertia_invweight_and_meaninertia():
    """Free box, mass 2, half sizes (0.1, 0.2, 0.3): I = m/3 (b^2+c^2, a^2+c^2, a^2+b^2) = (0.0866667, 0.0666667, 0.0333333).
    M = diag(m, m, m, I) at qpos0, so dof_invweight0 = (1/m x3, mean(1/I) x3) - MuJoCo averages the three translations and the
    three rotations of a free joint - with mean(1/I) = (11.538462 + 15 + 30) / 3 = 18.846154; body_invweight0 = (1/m, mean(1/I));
    stat.meaninertia = trace(M) / nv = (6 + 0.1866667) / 6 = 1.0311111."""
    m = MjModel.from_xml_string('<mujoco><worldbody><body><freejoint/><geom type="box" size="0.1 0.2 0.3" mass="2"/></body></worldbody></mujoco>')
    I = np.array([2 / 3 * (0.04 + 0.09), 2 / 3 * (0.01 + 0.09), 2 / 3 * (0.01 + 0.04)])
    assert np.allclose(m.body_mass[1], 2) and np.allclose(np.sort(m.body_inertia[1]), np.sort(I), rtol=1e-12)
    assert np.allclose(m.dof_invweight0[:3], 0.5, rtol=1e-12) and np.allclose(m.dof_invweight0[3:], np.mean(1 / I), rtol=1e-10)
    assert np.allclose(m.body_invweight0[1], [0.5, np.mean(1 / I)], rtol=1e-10)
    assert abs(m.stat.meaninertia - (6 + I.sum()) / 6) < 1e-12


def test_geom_masses_from_density():
    """Default density 1000: sphere r = 0.1 -> 4/3 pi r^3 rho = 4.1887902; capsule r = 0.05, half length 0.2 -> (pi r^2 2h + 4/3 pi
    r^3) rho = 3.6651914; cylinder r = 0.05, half height 0.1 -> pi r^2 2h rho = 1.5707963; box 0.1^3 half sizes -> 8 rho / 1000 = 8."""
    m = MjModel.from_xml_string('<mujoco><worldbody>'
                                '<body><freejoint/><geom type="sphere" size="0.1"/></body>'
                                '<body pos="1 0 0"><freejoint/><geom type="capsule" size="0.05 0.2"/></body>'
                                '<body pos="2 0 0"><freejoint/><geom type="cylinder" size="0.05 0.1" contype="0" conaffinity="0"/></body>'
                                '<body pos="3 0 0"><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body></worldbody></mujoco>')
    assert np.allclose(m.body_mass[1:], [4.18879020, 3.66519143, 1.57079633, 8.0], rtol=1e-8)
    # sphere: I = 2/5 m r^2
    assert np.allclose(m.body_inertia[1], 0.4 * 4.18879020 * 0.01, rtol=1e-8)
    # capsule (cylinder m_c = 3.14159265 + two hemispheres m_s = 0.52359878): axial 1/2 m_c r^2 + 2/5 m_s r^2 = 0.00445059,
    # transverse m_c (r^2/4 + h^2/3) + m_s (2/5 r^2 + h^2 + 3/4 h r) = 0.06924594
    assert np.allclose(np.sort(m.body_inertia[2]), [0.00445058959, 0.06924593807, 0.06924593807], rtol=1e-8)
    # cylinder: axial 1/2 m r^2 = 0.00196350, transverse m (3 r^2 + (2h)^2) / 12 = 0.00621774
    assert np.allclose(np.sort(m.body_inertia[3]), [0.00196349541, 0.00621773546, 0.00621773546], rtol=1e-8)


PAIR = """<mujoco><worldbody>
  <body pos="0 0 0.5"><freejoint/><geom name="a" type="sphere" size="0.1" friction="1.0 0.005 0.0001" solref="{ra}" solimp="0.9 0.95 0.001 0.5 2" solmix="{ma}" margin="0.01" gap="0.002" priority="{pa}"/></body>
  <body pos="0 0 0.9"><freejoint/><geom name="b" type="sphere" size="0.1" friction="0.5 0.01 0.0002" solref="{rb}" solimp="0.8 0.99 0.002 0.5 2" solmix="{mb}" margin="0.03" gap="0.005" priority="{pb}"/></body>
</worldbody></mujoco>"""


def test_contact_parameter_mixing():
    """mj_contactParam: equal priority - friction = elementwise max; solref, solimp = solmix-weighted average (weights 1 and 3 ->
    0.25 a + 0.75 b); negative (direct) solref - elementwise min instead; margin, gap = max, and the contact is detected within
    margin - gap.  Different priority - the higher-priority geom's friction, solref, solimp."""
    m = MjModel.from_xml_string(PAIR.format(ra="0.02 1", rb="0.01 0.5", ma=1, mb=3, pa=0, pb=0))
    assert m.npair == 1
    assert np.allclose(m.pair_friction[0], [1.0, 0.01, 0.0002])
    assert np.allclose(m.pair_solref[0], [0.25 * 0.02 + 0.75 * 0.01, 0.25 * 1 + 0.75 * 0.5])
    assert np.allclose(m.pair_solimp[0], 0.25 * np.array([0.9, 0.95, 0.001, 0.5, 2]) + 0.75 * np.array([0.8, 0.99, 0.002, 0.5, 2]))
    assert np.allclose(m.pair_margin[0], 0.03 - 0.005)
    m = MjModel.from_xml_string(PAIR.format(ra="-100 -5", rb="-300 -2", ma=1, mb=1, pa=0, pb=0))
    assert np.allclose(m.pair_solref[0], [-300, -5])
    m = MjModel.from_xml_string(PAIR.format(ra="0.02 1", rb="0.01 0.5", ma=1, mb=3, pa=2, pb=1))
    assert np.allclose(m.pair_friction[0], [1.0, 0.005, 0.0001]) and np.allclose(m.pair_solref[0], [0.02, 1])
    assert np.allclose(m.pair_solimp[0], [0.9, 0.95, 0.001, 0.5, 2])


def test_static_pair_filter():
    """Candidate pairs: geoms of one body never collide; parent and child bodies do not (default filterparent), but a child of
    the world does collide with world geoms; the bit test is (contype_a & conaffinity_b) || (contype_b & conaffinity_a);
    two planes never pair."""
    xml = """<mujoco><worldbody>
      <geom name="floor" type="plane" size="1 1 0.1"/>
      <geom name="wall" type="plane" size="1 1 0.1" pos="1 0 0" euler="0 90 0"/>
      <body name="A" pos="0 0 0.5"><freejoint/>
        <geom name="a1" type="sphere" size="0.1"/><geom name="a2" type="sphere" size="0.1" pos="0.3 0 0"/>
        <body name="B" pos="0 0 0.3"><joint type="hinge" axis="0 1 0"/><geom name="b1" type="sphere" size="0.1"/>
          <body name="C" pos="0 0 0.3"><joint type="hinge" axis="0 1 0"/><geom name="c1" type="sphere" size="0.1" contype="2" conaffinity="4"/></body>
        </body>
      </body>
      <body name="D" pos="1 1 0.5"><freejoint/><geom name="d1" type="sphere" size="0.1" contype="4" conaffinity="0"/></body>
    </worldbody></mujoco>"""
    m = MjModel.from_xml_string(xml)
    from ambersim_amd.utils.introspection_utils import get_geom_names
    names = get_geom_names(m)
    got = {frozenset((names[a], names[b])) for a, b in zip(m.pair_geom1, m.pair_geom2)}
    want = {frozenset(p) for p in (("floor", "a1"), ("floor", "a2"), ("floor", "b1"), ("wall", "a1"), ("wall", "a2"), ("wall", "b1"),
                                   ("c1", "d1"))}
    # a1/a2 (same body), a*/b1 and b1/c1 (parent - child) are filtered; c1 (contype 2, conaffinity 4) meets d1 (contype 4) through
    # contype_d & conaffinity_c; c1 does not meet the default geoms (1 & 4 = 0, 2 & 1 = 0), nor the planes; d1 (conaffinity 0,
    # contype 4) meets nothing with conaffinity 1; floor / wall do not pair
    assert got == want, (sorted(map(sorted, got)), sorted(map(sorted, want)))


LIMIT_HINGE = """<mujoco><compiler angle="radian"/><option gravity="0 0 0" timestep="0.002" iterations="20" ls_iterations="20"/><worldbody>
  <body><joint name="j" type="hinge" axis="0 1 0" limited="true" range="-0.5 0.5"/>
    <geom type="capsule" fromto="0 0 0 0 0 -0.4" size="0.03" mass="1.3"/></body>
</worldbody></mujoco>"""

# One hinge beyond its upper limit, no gravity, default solref (0.02, 1) and solimp (0.9, 0.95, 0.001, 0.5, 2).  The limit row has
# J = -1, pos = 0.5 - q, diagApprox = dof_invweight0 = 1/M, so R = (1 - imp)/imp / M, D = imp/(1 - imp) M, and minimising
# 1/2 M a^2 + 1/2 D (J a - aref)^2 gives a = -imp * aref whatever M is, with aref = -b (J v) - k imp pos, k = 1/(dmax^2 tc^2 dr^2)
# = 1/(0.9025 * 0.0004) = 2770.0831, b = 2/(dmax tc) = 105.26316.
#   q = 0.6,    v = 0: pos = -0.1 is past the 1 mm width, imp = dmax = 0.95, aref = 2770.0831 * 0.95 * 0.1 = 263.15789, qacc = -250
#   q = 0.6,    v = 1: aref = 105.26316 + 263.15789 = 368.42105, qacc = -350
#   q = 0.5 + 2^-11 (exact in fp32), v = 0: x = |pos|/width = 0.48828125 < midpoint, y = 2 x^2 = 0.47683716, imp = 0.9 + 0.05 y
#                      = 0.92384186, aref = 2770.0831 * 0.92384186 * 0.00048828125 = 1.2495697, qacc = -imp * aref = -1.1544048
LIMIT_CASES = [(0.6, 0.0, -250.0), (0.6, 1.0, -350.0), (0.5 + 2.0 ** -11, 0.0, -1.1544048)]


def test_limit_row_closed_form_oracle():
    """The CPU restatement's impedance, reference acceleration, regulariser and solve on the one-row case worked out above."""
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(LIMIT_HINGE)
    o = Oracle(mj, np.float64)
    for q, v, want in LIMIT_CASES:
        o.set_state(np.array([q]), np.array([v]), np.zeros(0)); o.forward()
        assert abs(o.get("qacc")[0] - want) < 2e-6 * abs(want), (q, v, o.get("qacc")[0], want)


@pytest.mark.gpu
def test_limit_row_closed_form_gpu():
    """The same three numbers from the HIP path (amx_forward): not through the oracle."""
    import torch
    from test_gpu_kat import Gpu
    mj = MjModel.from_xml_string(LIMIT_HINGE)
    g = Gpu(mj, torch.device("cuda", 0))
    for q, v, want in LIMIT_CASES:
        f = g.forward(np.array([q]), np.array([v]))
        assert abs(f["qacc"][0] - want) < 2e-5 * abs(want), (q, v, f["qacc"][0], want)
        assert f["nefc"] == 1


LIMIT_BALL = """<mujoco><compiler angle="radian"/><option gravity="0 0 0" timestep="0.002" iterations="20" ls_iterations="20"/><worldbody>
  <body><joint name="j" type="ball" limited="true" range="0 0.5"/>
    <inertial pos="0 0 0" mass="1" diaginertia="0.02 0.02 0.02"/></body>
</worldbody></mujoco>"""
LIMIT_BALL_ARM = """<mujoco><compiler angle="radian"/><option timestep="0.002" iterations="10" ls_iterations="10"/><worldbody>
  <body pos="0 0 1"><joint type="hinge" axis="0 1 0" damping="0.1"/><geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.03" mass="1"/>
    <body pos="0.3 0 0"><joint type="ball" limited="true" range="0 0.4" damping="0.02"/><geom type="capsule" fromto="0 0 0 0.05 0 -0.3" size="0.025" mass="0.6"/>
      <body pos="0.05 0 -0.3"><joint type="hinge" axis="1 0 0" limited="true" range="-0.3 0.3"/><geom type="capsule" fromto="0 0 0 0 0 -0.2" size="0.02" mass="0.3"/></body>
    </body></body>
</worldbody></mujoco>"""
# A ball joint turned by 0.6 rad about a = (1, 2, 2)/3 against a limit of 0.5 rad, isotropic inertia I about the joint, no gravity:
# one row, J = -a on the three dofs (|J| = 1), pos = 0.5 - 0.6, diagApprox = dof_invweight0 = 1/I = J M^-1 J', so the hinge case's
# algebra carries over: J qacc = imp * aref, qacc = -a * imp * aref: 250 at rest, 350 with the body-frame angular velocity 1 * a
# (J v = -1; w x I w = 0 for an isotropic inertia).  Turned by 2 pi - 0.6 about a the rotation is -0.6 about a: the angle wraps,
# the axis flips, and the limit pushes the other way round.
_BA = np.array([1.0, 2.0, 2.0]) / 3.0
LIMIT_BALL_CASES = [(0.6, 0.0, -250.0), (0.6, 1.0, -350.0), (2 * np.pi - 0.6, 0.0, 250.0)]


def _ball_q(angle):
    return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * _BA])


def test_ball_limit_row_closed_form_oracle():
    """SURVEY.md A9.6 / MJX _instantiate_limit_ball: the one-row case worked out above, and no row inside the range."""
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(LIMIT_BALL)
    assert mj.jnt_limited.tolist() == [1] and np.allclose(mj.dof_invweight0, 50.0)
    for dt, tol in ((np.float64, 2e-6), (np.float32, 2e-5)):
        o = Oracle(mj, dt)
        for ang, w, want in LIMIT_BALL_CASES:
            o.set_state(_ball_q(ang), w * _BA, np.zeros(0), np.zeros(3)); o.forward()
            assert np.abs(o.get("qacc") - want * _BA).max() < tol * abs(want), (ang, w, o.get("qacc"), want * _BA)
        o.set_state(_ball_q(0.45), np.zeros(3), np.zeros(0), np.zeros(3)); o.forward()
        assert np.abs(o.get("qacc")).max() == 0.0


@pytest.mark.gpu
def test_ball_limit_row_closed_form_gpu():
    """The same numbers from the HIP path (amx_forward), not through the oracle; then a three-link arm with a limited ball joint
    in the middle, driven into the limit, against the fp64 oracle over a rollout."""
    import torch
    from test_gpu_kat import Gpu
    from oracle.oracle import Oracle
    dev = torch.device("cuda", 0)
    g = Gpu(MjModel.from_xml_string(LIMIT_BALL), dev)
    for ang, w, want in LIMIT_BALL_CASES:
        f = g.forward(_ball_q(ang), w * _BA)
        assert np.abs(f["qacc"] - want * _BA).max() < 2e-5 * abs(want), (ang, w, f["qacc"], want * _BA)
        assert f["nefc"] == 1
    f = g.forward(_ball_q(0.45), np.zeros(3))
    assert f["nefc"] == 0 and np.abs(f["qacc"]).max() == 0.0
    mj = MjModel.from_xml_string(LIMIT_BALL_ARM)
    g = Gpu(mj, dev)
    rng = np.random.default_rng(3)
    nenv, steps = 16, 50
    q = np.tile(mj.qpos0, (nenv, 1))
    for k in range(nenv):
        a = rng.standard_normal(3); a /= np.linalg.norm(a)
        ang = rng.uniform(0.2, 0.6)                                                  # some inside the range, some beyond it
        q[k, 1:5] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * a])
        q[k, 0], q[k, 5] = rng.uniform(-0.5, 0.5), rng.uniform(-0.4, 0.4)
    v = 0.5 * rng.standard_normal((nenv, mj.nv))
    x0 = np.concatenate([q, v], axis=1).astype(np.float32)
    us = np.zeros((nenv, steps, 0), dtype=np.float32)
    xs = g.rollout(x0, us)
    ref = Oracle(mj, np.float64).rollout(x0, us)
    e1 = np.abs(xs[:, 1] - ref[:, 1]).max() / max(1.0, np.abs(ref[:, 1]).max())
    eN = np.abs(xs[:, -1] - ref[:, -1]).max() / max(1.0, np.abs(ref[:, -1]).max())
    print(f"ball-limit arm: 1-step err {e1:.2e}, {steps}-step err {eN:.2e}")
    assert e1 < 1e-5 and eN < 1e-4 and np.isfinite(xs).all()


FRICTION_HINGE = """<mujoco><option timestep="0.002" gravity="0 0 0" iterations="30" ls_iterations="30"/><worldbody>
  <body><joint name="h" type="hinge" axis="0 0 1" frictionloss="1.0"/><inertial pos="0 0 0" mass="1" diaginertia="0.05 0.05 0.05"/></body>
</worldbody><actuator><motor joint="h" gear="1"/></actuator></mujoco>"""

# Dry joint friction (SURVEY.md A9.6 "friction loss" rows; reaches the reference through URDF <dynamics friction=...>).  One hinge,
# inertia I = 0.05, no gravity, frictionloss f = 1, motor torque tau, default solreffriction (0.02, 1) / solimpfriction (0.9, ...).
# The friction row has J = 1, pos = 0, so its impedance is d(0) = 0.9, R = (1 - imp)/imp * invweight0 = (1/9)(1/I), D = 9 I = 0.45,
# aref = -b v with b = 2/(dmax tc) = 105.26316, and its force is clamped to +-f: with z = a - aref,
#   stick (|D z| < f):  minimise 1/2 I a^2 - tau a + 1/2 D z^2  ->  a = (tau + D aref) / (I + D)
#   slip:               force = -f sign(z)                      ->  a = (tau - f sign(z)) / I
#   v = 0,   tau = 0.5: a = 0.5 / 0.5 = 1            (D a = 0.45 < 1: stick)
#   v = 0,   tau = 3:   a = (3 - 1) / 0.05 = 40      (0.9 tau >= f: slip);  tau = -3: a = -40
#   v = 0.2, tau = 0:   a = -0.9 * 105.26316 * 0.2 = -18.947368   (z = 2.105 < f / D = 2.222: stick)
#   v = 0.5, tau = 0:   a = -f / I = -20                          (z = -20 + 52.63 = 32.6 >= 2.222: slip)
FRICTION_CASES = [(0.0, 0.5, 1.0), (0.0, 3.0, 40.0), (0.0, -3.0, -40.0), (0.2, 0.0, -18.947368421), (0.5, 0.0, -20.0)]


def test_friction_loss_closed_form_oracle():
    """The CPU restatement's friction-loss row: regulariser, reference acceleration, both zones of the bounded force."""
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(FRICTION_HINGE)
    assert mj.dof_frictionloss.tolist() == [1.0] and abs(mj.dof_invweight0[0] - 20.0) < 1e-12
    for dt in (np.float64, np.float32):
        o = Oracle(mj, dt)
        for v, tau, want in FRICTION_CASES:
            o.set_state(np.array([0.0]), np.array([v]), np.array([tau])); o.forward()
            assert abs(o.get("qacc")[0] - want) < (1e-9 if dt is np.float64 else 2e-5) * abs(want), (v, tau, o.get("qacc")[0], want)
            assert int(o.get("nefc")[0]) == 1
    # the flag <option><flag frictionloss="disable"/> removes the row
    off = MjModel.from_xml_string(FRICTION_HINGE.replace('iterations="30"', 'iterations="30"><flag frictionloss="disable"/></option><option'))
    o = Oracle(off, np.float64)
    o.set_state(np.array([0.0]), np.array([0.0]), np.array([0.5])); o.forward()
    assert abs(o.get("qacc")[0] - 10.0) < 1e-9 and int(o.get("nefc")[0]) == 0


@pytest.mark.gpu
def test_friction_loss_closed_form_gpu():
    """The same numbers from the HIP path (amx_forward, never through the oracle), then a step-level check against the oracle on a
    two-link arm with friction on both joints, limits and gravity (generic kernels and a per-model build)."""
    import torch
    from test_gpu_kat import Gpu
    from oracle.oracle import Oracle
    dev = torch.device("cuda", 0)
    mj = MjModel.from_xml_string(FRICTION_HINGE)
    g = Gpu(mj, dev)
    for v, tau, want in FRICTION_CASES:
        f = g.forward(np.array([0.0]), np.array([v]), np.array([tau]))
        assert abs(f["qacc"][0] - want) < 2e-5 * abs(want), (v, tau, f["qacc"][0], want)
        assert f["nefc"] == 1
    arm = MjModel.from_xml_string(FRICTION_ARM)
    assert arm.dof_frictionloss.tolist() == [0.3, 0.1]
    rng = np.random.default_rng(5)
    x0 = np.concatenate([rng.uniform(-1.0, 1.0, (16, 2)), rng.uniform(-3, 3, (16, 2))], axis=1)
    x0[:4, 2:] = 0.0  # some start at rest: the rows hold them (stick)
    us = rng.uniform(-1, 1, (16, 20, 2))
    ref = Oracle(arm, np.float64).rollout(x0, us)
    for spec in (False, True):
        from ambersim_amd import mjx
        model = mjx.put_model(arm, dev, specialise=spec)
        xs = mjx.rollout(model, torch.tensor(x0, dtype=torch.float32), torch.tensor(us, dtype=torch.float32)).cpu().numpy()
        err1 = np.abs(xs[:, 1] - ref[:, 1]).max() / max(1.0, np.abs(ref[:, 1]).max())
        err20 = np.abs(xs[:, 20] - ref[:, 20]).max() / max(1.0, np.abs(ref[:, 20]).max())
        assert err1 < 1e-5 and err20 < 1e-4, (spec, err1, err20)


FRICTION_ARM = """<mujoco><compiler angle="radian"/><option timestep="0.002" iterations="10" ls_iterations="10"/><worldbody>
  <body pos="0 0 1"><joint name="a" type="hinge" axis="0 1 0" frictionloss="0.3" damping="0.05" range="-2 2"/>
    <geom type="capsule" fromto="0 0 0 0 0 -0.4" size="0.03" mass="1" contype="0" conaffinity="0"/>
    <body pos="0 0 -0.4"><joint name="b" type="hinge" axis="0 1 0" frictionloss="0.1" armature="0.01" range="-1.5 1.5"/>
      <geom type="capsule" fromto="0 0 0 0 0 -0.3" size="0.025" mass="0.5" contype="0" conaffinity="0"/></body></body>
</worldbody><actuator><motor joint="a" gear="2" ctrlrange="-1 1"/><motor joint="b" gear="1" ctrlrange="-1 1"/></actuator></mujoco>"""


BALL_ON_PLANE = """<mujoco><option timestep="0.002" iterations="30" ls_iterations="30"/><worldbody>
  <geom type="plane" size="2 2 0.1" friction="1 0.005 0.0001"/>
  <body pos="0 0 0.1"><freejoint/><geom type="sphere" size="0.1" mass="0.7" friction="1 0.005 0.0001"/></body>
</worldbody></mujoco>"""

# A ball pressed 2^-10 m into the plane, friction 1, pyramidal cone (4 rows J_n +- mu J_t), impratio 1.  Each row's regulariser uses
# invweight = (tw + mu^2 tw) 2 mu^2 / impratio with tw = body_invweight0 (translation) of the ball + of the world = 1/m + 0, i.e.
# 4/m: D = imp/(1 - imp) m/4 per row.  With no tangential motion the four rows all read J a = a_z and share aref, the tangential
# forces cancel, and minimising 1/2 m (a_z + g)^2 + 4 * 1/2 D (a_z - aref)^2 gives a_z = (1 - imp)(-g) + imp aref:
#   pos = -2^-10: x = 0.9765625 > midpoint, y = 1 - 2 (1 - x)^2 = 0.99890137, imp = 0.9 + 0.05 y = 0.94994507,
#   v_z = 0:     aref = k imp |pos| = 2770.0831 * 0.94994507 * 0.0009765625 = 2.5697527,           a_z = 1.9500850
#   v_z = -0.25: aref = 2.5697527 + b * 0.25 = 2.5697527 + 26.3157895 = 28.8855422,                a_z = 26.9486395
BALL_CASES = [(0.0, 1.9500850), (-0.25, 26.9486395)]


def _ball_state(vz):
    q = np.array([0, 0, 0.1 - 2.0 ** -10, 1, 0, 0, 0.0]); v = np.zeros(6); v[2] = vz
    return q, v


def test_contact_row_closed_form_oracle():
    """The CPU restatement on the pyramid case above: vertical acceleration as computed by hand, nothing else moves."""
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(BALL_ON_PLANE)
    o = Oracle(mj, np.float64)
    for vz, want in BALL_CASES:
        q, v = _ball_state(vz)
        o.set_state(q, v, np.zeros(0)); o.forward()
        a = o.get("qacc")
        assert abs(a[2] - want) < 5e-6 * abs(want) and np.abs(np.delete(a, 2)).max() < 1e-6, (vz, a, want)


@pytest.mark.gpu
def test_contact_row_closed_form_gpu():
    """The same two numbers from the HIP path."""
    import torch
    from test_gpu_kat import Gpu
    mj = MjModel.from_xml_string(BALL_ON_PLANE)
    g = Gpu(mj, torch.device("cuda", 0))
    for vz, want in BALL_CASES:
        q, v = _ball_state(vz)
        f = g.forward(q, v)
        assert abs(f["qacc"][2] - want) < 5e-5 * abs(want) and np.abs(np.delete(f["qacc"], 2)).max() < 1e-3, (vz, f["qacc"], want)
        assert f["nefc"] == 4


SPRING_HINGE = """<mujoco><compiler angle="radian"/><option gravity="0 0 0" timestep="0.002"/><worldbody>
  <body><inertial pos="0 0 0" mass="1" diaginertia="0.1 0.2 0.3"/>
    <joint type="hinge" axis="0 1 0" armature="0.05" damping="0.4" stiffness="3" springref="0.1"/></body>
</worldbody></mujoco>"""
# M = I_yy + armature = 0.25.  At q = 0.3, v = 2: qfrc_passive = -0.4 * 2 - 3 (0.3 - 0.1) = -1.4, so forward gives qacc = -5.6.  The
# Euler step is implicit in the joint damping: (M + dt d) a = -1.4, a = -1.4 / 0.2508 = -5.5821372, v' = 2 + 0.002 a = 1.9888357,
# q' = 0.3 + 0.002 v' = 0.3039777.

EQUALITY = """<mujoco><compiler angle="radian"/><option gravity="0 0 0" timestep="0.002" iterations="20" ls_iterations="20"/><worldbody>
  <body><inertial pos="0 0 0" mass="1" diaginertia="0.1 0.2 0.1"/><joint name="a" type="hinge" axis="0 1 0"/></body>
  <body pos="1 0 0"><inertial pos="0 0 0" mass="1" diaginertia="0.1 0.3 0.1"/><joint name="b" type="hinge" axis="0 1 0"/></body>
</worldbody><equality><joint joint1="a" joint2="b" polycoef="0.05 0.5 0 0 0"/></equality></mujoco>"""
# q_a = 0.05 + 0.5 q_b is violated at (0.2, 0.1) by pos = 0.1 (beyond the impedance width: imp = 0.95).  J = (1, -0.5) on
# M = diag(0.2, 0.3); aref = -k imp pos = -263.15789; J M^-1 J' = 5 + 0.25/0.3 = 5.8333333; the row's diagApprox is the SUM of the two
# dofs' invweight0, 5 + 3.3333333, so R = (0.05/0.95) 8.3333333 = 0.43859649; lambda = aref / (5.8333333 + 0.43859649) = -41.958042;
# qacc = M^-1 J' lambda = (-209.79021, 69.930070).


def _spring_checks(fwd, step):
    assert abs(fwd - (-5.6)) < 1e-5 * 5.6
    assert abs(step[0] - 0.3039776715) < 2e-7 and abs(step[1] - 1.9888357257) < 2e-6


def test_passive_forces_armature_and_implicit_damping_oracle():
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(SPRING_HINGE)
    o = Oracle(mj, np.float64)
    o.set_state(np.array([0.3]), np.array([2.0]), np.zeros(0)); o.forward()
    xs = o.rollout(np.array([0.3, 2.0]), np.zeros((1, 1, 0)))[0]
    _spring_checks(o.get("qacc")[0], xs[1])


def test_joint_equality_row_closed_form_oracle():
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(EQUALITY)
    o = Oracle(mj, np.float64)
    o.set_state(np.array([0.2, 0.1]), np.zeros(2), np.zeros(0)); o.forward()
    assert np.allclose(o.get("qacc"), [-209.79020979, 69.93006993], rtol=5e-6)


@pytest.mark.gpu
def test_passive_damping_and_equality_closed_forms_gpu():
    """The two cases above from the HIP path."""
    import torch
    from test_gpu_kat import Gpu
    dev = torch.device("cuda", 0)
    g = Gpu(MjModel.from_xml_string(SPRING_HINGE), dev)
    _spring_checks(g.forward(np.array([0.3]), np.array([2.0]))["qacc"][0], g.rollout(np.array([0.3, 2.0]), np.zeros((1, 1, 0)))[0][1])
    g = Gpu(MjModel.from_xml_string(EQUALITY), dev)
    assert np.allclose(g.forward(np.array([0.2, 0.1]), np.zeros(2))["qacc"], [-209.79020979, 69.93006993], rtol=5e-5)


DOUBLE_PENDULUM = """<mujoco><compiler angle="radian"/><option timestep="0.002"/><worldbody>
  <body pos="0 0 1"><joint type="hinge" axis="0 1 0"/><inertial pos="0 0 -0.25" mass="1.2" diaginertia="0.03 0.03 0.03"/>
    <body pos="0 0 -0.5"><joint type="hinge" axis="0 1 0"/><inertial pos="0 0 -0.2" mass="0.8" diaginertia="0.02 0.02 0.02"/></body>
  </body>
</worldbody></mujoco>"""
# The textbook planar double pendulum (angles from the hanging position, second angle relative): m1 = 1.2, lc1 = 0.25, l1 = 0.5,
# I1 = 0.03, m2 = 0.8, lc2 = 0.2, I2 = 0.02, g = 9.81, at theta = (0.3, 0.5), theta' = (1.1, -0.7):
#   M11 = I1 + I2 + m1 lc1^2 + m2 (l1^2 + lc2^2 + 2 l1 lc2 cos t2) = 0.49741321, M12 = I2 + m2 (lc2^2 + l1 lc2 cos t2) = 0.12220660,
#   M22 = I2 + m2 lc2^2 = 0.052;  h = m2 l1 lc2 sin t2: Coriolis c = (-h (2 w1 w2 + w2^2), h w1^2) = (0.04027175, 0.04640839);
#   gravity G = ((m1 lc1 + m2 l1) g sin t1 + m2 lc2 g sin(t1 + t2), m2 lc2 g sin(t1 + t2)) = (3.15529938, 1.12596212);
#   qfrc_bias = c + G = (3.19557112, 1.17237051).
DP_M = np.array([[0.4974132099, 0.1222066050], [0.1222066050, 0.052]])
DP_BIAS = np.array([3.1955711247, 1.1723705124])


def test_double_pendulum_mass_matrix_and_bias_oracle():
    """Composite-rigid-body mass matrix and recursive Newton-Euler bias of the CPU restatement against the textbook formulas."""
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(DOUBLE_PENDULUM)
    o = Oracle(mj, np.float64)
    o.set_state(np.array([0.3, 0.5]), np.array([1.1, -0.7]), np.zeros(0)); o.forward()
    assert np.allclose(o.get("qM", (2, 2)), DP_M, rtol=1e-9) and np.allclose(o.get("qfrc_bias"), DP_BIAS, rtol=1e-9)


@pytest.mark.gpu
def test_double_pendulum_mass_matrix_and_bias_gpu():
    """The same from the HIP path (amx_forward's qM and qfrc_bias outputs), in fp32."""
    import torch
    from test_gpu_kat import Gpu
    f = Gpu(MjModel.from_xml_string(DOUBLE_PENDULUM), torch.device("cuda", 0)).forward(np.array([0.3, 0.5]), np.array([1.1, -0.7]))
    assert np.allclose(f["qM"], DP_M, rtol=2e-6) and np.allclose(f["qfrc_bias"], DP_BIAS, rtol=2e-6)
    assert np.allclose(f["qacc"], -np.linalg.solve(DP_M, DP_BIAS), rtol=1e-5)  # free motion: M qacc = -bias


FREE_BODY = """<mujoco><option gravity="0 0 0" timestep="0.002"/><worldbody>
  <body><freejoint/><inertial pos="0 0 0" mass="2" diaginertia="0.1 0.2 0.3"/></body>
</worldbody></mujoco>"""
# Torque-free rigid body (Euler's equations; the free joint's angular velocity is in the body frame, the body frame is the principal
# frame): w = (1, 2, 3), I w = (0.1, 0.4, 0.9), w x I w = (0.6, -0.6, 0.2), w' = -I^-1 (w x I w) = (-6, 3, -0.6666667); no linear
# acceleration.  One semi-implicit Euler step from the identity orientation: w1 = w + dt w' = (0.988, 2.006, 2.9986667), the quaternion
# turns by the angle dt |w1| about w1 / |w1|: q' = (cos(a/2), sin(a/2) w1 / |w1|) with a = 0.002 * 3.7406125 = 0.0074812250:
# q' = (0.99999300, 0.00098800, 0.00200600, 0.00299866).
FREE_QACC = np.array([0, 0, 0, -6.0, 3.0, -2.0 / 3.0])
FREE_QUAT1 = np.array([0.9999930039173, 0.0009879976960, 0.0020059953220, 0.0029986596737])


def test_free_body_euler_equations_oracle():
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(FREE_BODY)
    o = Oracle(mj, np.float64)
    q, v = np.array([0, 0, 0, 1, 0, 0, 0.0]), np.array([0, 0, 0, 1, 2, 3.0])
    o.set_state(q, v, np.zeros(0)); o.forward()
    assert np.allclose(o.get("qacc"), FREE_QACC, atol=1e-9)
    xs = o.rollout(np.concatenate([q, v]), np.zeros((1, 1, 0)))[0]
    assert np.allclose(xs[1, 3:7], FREE_QUAT1, atol=2e-9) and np.allclose(xs[1, 10:], [0.988, 2.006, 3 - 0.002 * 2 / 3], atol=1e-12)


@pytest.mark.gpu
def test_free_body_euler_equations_gpu():
    import torch
    from test_gpu_kat import Gpu
    g = Gpu(MjModel.from_xml_string(FREE_BODY), torch.device("cuda", 0))
    q, v = np.array([0, 0, 0, 1, 0, 0, 0.0]), np.array([0, 0, 0, 1, 2, 3.0])
    assert np.allclose(g.forward(q, v)["qacc"], FREE_QACC, atol=2e-5)
    xs = g.rollout(np.concatenate([q, v]), np.zeros((1, 1, 0)))[0]
    assert np.allclose(xs[1, 3:7], FREE_QUAT1, atol=2e-7) and np.allclose(xs[1, 10:], [0.988, 2.006, 3 - 0.002 * 2 / 3], atol=2e-6)


def _two_geoms(g1, p1, g2, p2, extra2=""):
    return MjModel.from_xml_string(f'<mujoco><compiler angle="degree"/><worldbody><body pos="{p1}"><freejoint/><geom {g1}/></body>'
                                   f'<body pos="{p2}" {extra2}><freejoint/><geom {g2}/></body></worldbody></mujoco>')


# Primitive pair distances by elementary geometry (contact distance = centre / axis distance minus the radii):
#   spheres r 0.1 and 0.2, centres 0.25 apart: 0.25 - 0.3 = -0.05
#   sphere r 0.1 at the origin vs capsule r 0.05, half length 0.3 along x, centred at (0.1, 0.12, 0): nearest axis point (0, 0.12, 0):
#     0.12 - 0.15 = -0.03
#   capsule r 0.05 along z at the origin vs capsule r 0.04 along x centred at (0, 0.08, 0.1), half lengths 0.3: the axes' common
#     perpendicular joins (0, 0, 0.1) and (0, 0.08, 0.1): 0.08 - 0.09 = -0.01
#   sphere r 0.1 vs sphere r 0.1, 0.5 apart (separated): +0.3
PRIM_CASES = [
    (('type="sphere" size="0.1"', "0 0 1", 'type="sphere" size="0.2"', "0.25 0 1", ""), -0.05),
    (('type="sphere" size="0.1"', "0 0 0", 'type="capsule" size="0.05 0.3"', "0.1 0.12 0", 'euler="0 90 0"'), -0.03),
    (('type="capsule" size="0.05 0.3"', "0 0 0", 'type="capsule" size="0.04 0.3"', "0 0.08 0.1", 'euler="0 90 0"'), -0.01),
    (('type="sphere" size="0.1"', "0 0 0", 'type="sphere" size="0.1"', "0.3 0.4 0", ""), 0.3),
]


def test_primitive_pair_distances_oracle():
    from oracle.oracle import Oracle
    for args, want in PRIM_CASES:
        mj = _two_geoms(*args)
        assert mj.npair == 1
        o = Oracle(mj, np.float64)
        o.set_state(mj.qpos0, np.zeros(mj.nv), np.zeros(0)); o.forward()
        assert abs(o.get("con_dist")[0] - want) < 1e-12, (args, o.get("con_dist")[0], want)


@pytest.mark.gpu
def test_primitive_pair_distances_gpu():
    import torch
    from test_gpu_kat import Gpu
    for args, want in PRIM_CASES:
        mj = _two_geoms(*args)
        d = Gpu(mj, torch.device("cuda", 0)).forward(mj.qpos0, np.zeros(mj.nv))["dist"][0]
        assert abs(d - want) < 2e-7, (args, d, want)


SLIDING_BOX = """<mujoco><option timestep="0.002" iterations="50" ls_iterations="50"/><worldbody>
  <geom type="plane" size="5 5 0.1" friction="0.5 0.005 0.0001"/>
  <body pos="0 0 0.05"><freejoint/><geom type="box" size="0.1 0.1 0.05" mass="1" friction="0.5 0.005 0.0001"/></body>
</worldbody></mujoco>"""
# Coulomb's law on the pyramid: a box sliding along a pyramid axis (first tangent = world x for a z normal) is held by the row n - mu t1,
# whose force has tangential / normal = mu exactly; the rows n +- mu t2 add normal force only.  Over a window the impulses therefore
# satisfy  -m dvx = ratio * m (g T + dvz)  with ratio <= mu = 0.5, and close to it while the box slides (it chatters on its corners, so
# the instantaneous normal force is not m g - the impulse balance is what is exact).


def _friction_impulse_ratio(rollout):
    x0 = np.array([0, 0, 0.05, 1, 0, 0, 0, 2.5, 0, 0, 0, 0, 0.0])
    xs = rollout(x0, 180)
    assert xs[180, 7] > 0.5  # still sliding at the end of the window
    return -(xs[180, 7] - xs[40, 7]) / (9.81 * 140 * 0.002 + (xs[180, 9] - xs[40, 9]))


def test_sliding_friction_impulse_ratio_oracle():
    from oracle.oracle import Oracle
    o = Oracle(MjModel.from_xml_string(SLIDING_BOX), np.float64)
    r = _friction_impulse_ratio(lambda x0, n: o.rollout(x0, np.zeros((1, n, 0)))[0])
    assert 0.48 <= r <= 0.5 + 1e-4, r


@pytest.mark.gpu
def test_sliding_friction_impulse_ratio_gpu():
    import torch
    from test_gpu_kat import Gpu
    g = Gpu(MjModel.from_xml_string(SLIDING_BOX), torch.device("cuda", 0))
    r = _friction_impulse_ratio(lambda x0, n: g.rollout(x0, np.zeros((1, n, 0)))[0])
    assert 0.48 <= r <= 0.5 + 1e-4, r

SERVO = """<mujoco><compiler angle="radian"/><option gravity="0 0 0" timestep="0.002"/><worldbody>
  <body><inertial pos="0 0 0" mass="1" diaginertia="0.1 0.25 0.1"/><joint name="j" type="hinge" axis="0 1 0"/></body>
</worldbody><actuator>{act}</actuator></mujoco>"""
# Actuation on one hinge with M = 0.25, q = 0.1, v = 0.4 (no gravity, no passive force):
#   position kp = 20, ctrl 0.5:                         force = 20 (0.5 - 0.1) = 8            qacc = 32
#   the same with ctrlrange +-0.3 (ctrl clamped to 0.3): force = 20 (0.3 - 0.1) = 4            qacc = 16
#   velocity kv = 5, ctrl 1.0:                           force = 5 (1.0 - 0.4) = 3             qacc = 12
#   motor gear 3, ctrl 0.7, forcerange +-1.5:            force = clamp(0.7) = 0.7, torque 2.1  qacc = 8.4
#   motor gear 3, ctrl 2.0, forcerange +-1.5:            force clamped to 1.5, torque 4.5      qacc = 18
SERVO_CASES = [('<position joint="j" kp="20"/>', 0.5, 32.0),
               ('<position joint="j" kp="20" ctrllimited="true" ctrlrange="-0.3 0.3"/>', 0.5, 16.0),
               ('<velocity joint="j" kv="5"/>', 1.0, 12.0),
               ('<motor joint="j" gear="3" forcelimited="true" forcerange="-1.5 1.5"/>', 0.7, 8.4),
               ('<motor joint="j" gear="3" forcelimited="true" forcerange="-1.5 1.5"/>', 2.0, 18.0)]


def test_actuation_closed_forms_oracle():
    from oracle.oracle import Oracle
    for act, u, want in SERVO_CASES:
        o = Oracle(MjModel.from_xml_string(SERVO.format(act=act)), np.float64)
        o.set_state(np.array([0.1]), np.array([0.4]), np.array([u])); o.forward()
        assert abs(o.get("qacc")[0] - want) < 1e-9, (act, o.get("qacc")[0], want)


@pytest.mark.gpu
def test_actuation_closed_forms_gpu():
    import torch
    from test_gpu_kat import Gpu
    for act, u, want in SERVO_CASES:
        f = Gpu(MjModel.from_xml_string(SERVO.format(act=act)), torch.device("cuda", 0)).forward(np.array([0.1]), np.array([0.4]), np.array([u]))
        assert abs(f["qacc"][0] - want) < 1e-5 * want, (act, f["qacc"][0], want)


# ---- connect / weld equalities -------------------------------------------------------------------------------
EQ_BOXES = """<mujoco><option timestep="0.002" iterations="50" ls_iterations="50"><flag contact="disable"/></option><worldbody>
  <body name="a" pos="0 0 1"><freejoint/><geom type="box" size="0.1 0.05 0.2" mass="1"/></body>
  <body name="b" pos="0.5 0 1"><freejoint/><geom type="box" size="0.1 0.1 0.1" mass="2"/></body>
</worldbody><equality><connect body1="a" anchor="0 0 0.2"/><weld body1="a" body2="b"/></equality></mujoco>"""

# a closed loop: a two-link arm whose tip is pinned to the world by a connect (a four-bar-like mechanism with one free dof less)
EQ_LOOP = """<mujoco><compiler angle="radian"/><option timestep="0.002" iterations="20" ls_iterations="20"/><worldbody>
  <body name="l1" pos="0 0 1"><joint name="j1" type="hinge" axis="0 1 0" damping="0.05"/><geom type="capsule" fromto="0 0 0 0.4 0 0" size="0.02" mass="0.5" contype="0" conaffinity="0"/>
    <body name="l2" pos="0.4 0 0"><joint name="j2" type="hinge" axis="0 1 0" damping="0.05"/><geom type="capsule" fromto="0 0 0 0.3 0 -0.3" size="0.02" mass="0.4" contype="0" conaffinity="0"/>
      <body name="l3" pos="0.3 0 -0.3"><joint name="j3" type="ball" damping="0.02"/><geom type="capsule" fromto="0 0 0 -0.2 0 -0.3" size="0.02" mass="0.3" contype="0" conaffinity="0"/></body></body></body>
  <body name="m" pos="1.2 0 1"><freejoint/><geom type="sphere" size="0.08" mass="0.7" contype="0" conaffinity="0"/></body>
</worldbody><equality><connect body1="l3" anchor="-0.2 0 -0.3"/><weld body1="m" body2="l1" anchor="0.1 0 0" torquescale="0.5"/></equality>
<actuator><motor joint="j1" gear="2" ctrlrange="-1 1"/></actuator></mujoco>"""


def _integrate(mj, q, v, h):
    """qpos advanced by h * v (free and ball joints on the quaternion manifold): the oracle's own integrator is not used."""
    out = q.copy()
    def qmul(a, b):
        return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2], a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])
    for j in range(mj.njnt):
        qa, da, t = mj.jnt_qposadr[j], mj.jnt_dofadr[j], mj.jnt_type[j]
        if t in (0, 1):
            if t == 0:
                out[qa:qa + 3] += h * v[da:da + 3]; qa += 3; da += 3
            w = v[da:da + 3]; n = np.linalg.norm(w)
            if n > 0:
                dq = np.concatenate([[np.cos(h * n / 2)], np.sin(h * n / 2) * w / n])
                out[qa:qa + 4] = qmul(q[qa:qa + 4], dq)
        else:
            out[qa] += h * v[da]
    return out


@pytest.mark.parametrize("xml", [EQ_BOXES, EQ_LOOP])
def test_connect_weld_rows_oracle(xml):
    """The CPU restatement's connect / weld rows against two things it did not produce: (1) the Jacobian is the time derivative of
    the residual - central differences of efc_pos along a random velocity, on the quaternion manifold; (2) the solver's qacc is
    the minimiser of the quadratic it is given - a dense numpy solve of (M + J'DJ) a = M a0 + J'D aref (all rows of an equality
    are active, so with contact off the problem is one linear system)."""
    from oracle.oracle import Oracle
    mj = MjModel.from_xml_string(xml)
    assert set(mj.eq_type.tolist()) == {0, 1}
    o = Oracle(mj, np.float64)
    rng = np.random.default_rng(3)
    q = mj.qpos0.copy()
    q = _integrate(mj, q, 0.05 * rng.standard_normal(mj.nv), 1.0)  # a nearby configuration: non-zero residuals, generic orientations
    nu = mj.nu
    def rows(qq, vv=None):
        o.set_state(qq, np.zeros(mj.nv) if vv is None else vv, np.zeros(nu)); o.forward()
        ne = int(o.get("nefc")[0])
        return ne, o.get("efc_pos")[:ne].copy(), o.get("efc_J")[: ne * mj.nv].reshape(ne, mj.nv).copy()
    ne, r0, J = rows(q)
    assert ne == 9  # 3 (connect) + 6 (weld)
    vv = rng.standard_normal(mj.nv)
    eps = 1e-6
    fd = (rows(_integrate(mj, q, vv, eps))[1] - rows(_integrate(mj, q, vv, -eps))[1]) / (2 * eps)
    assert np.abs(fd - J @ vv).max() < 1e-6 * max(1.0, np.abs(fd).max()), (fd, J @ vv)
    ne, _, J = rows(q, 0.3 * rng.standard_normal(mj.nv))
    M, D, aref, a0 = o.get("qM", (mj.nv, mj.nv)), o.get("efc_D")[:ne], o.get("efc_aref")[:ne], o.get("qacc_smooth")
    a = np.linalg.solve(M + J.T @ (D[:, None] * J), M @ a0 + J.T @ (D * aref))
    assert np.abs(a - o.get("qacc")).max() < 1e-7 * max(1.0, np.abs(a).max())
    # the constraint works: released near the configuration it pulls back - after 200 steps the residual is far below where it started
    x0 = np.concatenate([q, np.zeros(mj.nv)])[None]
    xs = o.rollout(x0, np.zeros((1, 200, nu)))
    assert np.isfinite(xs).all()
    rN = rows(xs[0, -1, : mj.nq])[1]
    assert np.abs(rN[:3]).max() < 0.2 * max(np.abs(r0[:3]).max(), 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("xml", [EQ_BOXES, EQ_LOOP, EQ_BOXES.replace('<flag contact="disable"/>', "")])
def test_connect_weld_rows_gpu(xml):
    """The HIP path on the same models (and on the two boxes with their box - box contact on) from random nearby states, generic
    kernels and a per-model build: the number of active rows agrees, one step is within 1e-5 of the fp64 oracle - or within 3x
    the fp32 CPU oracle's own distance to it: a violated equality makes the cost large (k imp pos ~ 30 m/s^2 per mm), fp32
    resolves its improvement only to ~1e-7 of that, and the solver's improvement test stops both fp32 implementations at a
    gradient that is visible at 1e-5 (measured: tools/dbg/dbg_eq.py) - and ten steps within 5e-5, or 3x the fp32 oracle's
    own ten-step distance (the loop model, cond(H) = 4.5e5)."""
    import torch
    from ambersim_amd import mjx
    from oracle.oracle import Oracle
    from test_gpu_kat import Gpu
    mj = MjModel.from_xml_string(xml)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4)
    n = 16
    x0 = np.zeros((n, mj.nq + mj.nv))
    for e in range(n):
        x0[e, : mj.nq] = _integrate(mj, mj.qpos0.copy(), 0.01 * rng.standard_normal(mj.nv), 1.0)
        x0[e, mj.nq:] = 0.075 * rng.standard_normal(mj.nv)
    us = rng.uniform(-1, 1, (n, 10, mj.nu))
    o = Oracle(mj, np.float64)
    ref = o.rollout(x0, us)
    ref32 = Oracle(mj, np.float32).rollout(x0, us)
    floor1 = np.abs(ref32[:, 1] - ref[:, 1]).max() / max(1.0, np.abs(ref[:, 1]).max())
    floor10 = np.abs(ref32[:, 10] - ref[:, 10]).max() / max(1.0, np.abs(ref[:, 10]).max())
    g = Gpu(mj, dev)
    o32 = Oracle(mj, np.float32)
    for oo in (o, o32):
        oo.set_state(x0[0, : mj.nq], x0[0, mj.nq:], us[0, 0]); oo.forward()
    f = g.forward(x0[0, : mj.nq], x0[0, mj.nq:], us[0, 0])
    assert f["nefc"] == int(o.get("nefc")[0])
    # qacc: 2e-5, or 3x the fp32 CPU oracle's own error where the Hessian is ill-conditioned (the loop model's light ball-jointed link:
    # cond(H) = 4.5e5; both fp32 solvers stop at the same cost and gradient, tools/dbg/dbg_eq2.py)
    scale = max(1.0, np.abs(o.get("qacc")).max())
    assert np.abs(f["qacc"] - o.get("qacc")).max() < max(2e-5 * scale, 3 * np.abs(o32.get("qacc") - o.get("qacc")).max())
    for spec in (False, True):
        model = mjx.put_model(mj, dev, specialise=spec)
        xs = mjx.rollout(model, torch.tensor(x0, dtype=torch.float32), torch.tensor(us, dtype=torch.float32)).cpu().numpy()
        e1 = np.abs(xs[:, 1] - ref[:, 1]).max() / max(1.0, np.abs(ref[:, 1]).max())
        e10 = np.abs(xs[:, 10] - ref[:, 10]).max() / max(1.0, np.abs(ref[:, 10]).max())
        assert e1 < max(1e-5, 3 * floor1) and e1 < 1e-4 and e10 < max(5e-5, 3 * floor10) and e10 < 1e-3, (spec, e1, floor1, e10, floor10)


def test_foot_mesh_mass_com_and_inertia_by_independent_integration():
    """The headline model's feet are convex meshes whose mass, centre of mass and principal inertia the model compiler derives from the density
    (ambersim_amd/mjcf.py: its own hull and its own polyhedral integration) - values the oracle, the kernel and every fixture then share.  Pinned
    here by an independent route: qhull's hull of the same twelve vertices (scipy), integrated tetrahedron by tetrahedron from an interior point
    in numpy (volume, first and second moments in closed form per tetrahedron), density 700."""
    from scipy.spatial import ConvexHull
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    mj = load_mj_model_from_file("models/biped/biped_stand.xml")
    mesh = mj.names_mesh.index("foot") if hasattr(mj.names_mesh, "index") else 0
    V = np.asarray(mj.mesh_vert[mj.mesh_vertadr[mesh]: mj.mesh_vertadr[mesh] + mj.mesh_vertnum[mesh]], dtype=float)
    assert V.shape == (12, 3)
    h = ConvexHull(V)
    assert len(h.vertices) == 12  # every vertex of the file is a hull vertex (the planner's tables count on it)
    c0 = V[h.vertices].mean(axis=0)
    vol, first, second = 0.0, np.zeros(3), np.zeros((3, 3))
    for tri in h.simplices:
        a, b, c = V[tri] - c0
        v = abs(np.dot(a, np.cross(b, c))) / 6.0
        s = a + b + c
        vol += v
        first += v * s / 4.0
        second += v / 20.0 * (np.outer(a, a) + np.outer(b, b) + np.outer(c, c) + np.outer(s, s))
    com = first / vol
    second -= vol * np.outer(com, com)
    inertia = 700.0 * (np.trace(second) * np.eye(3) - second)
    assert abs(h.volume - vol) < 1e-12
    feet = [mj.names_body.index(n) for n in ("l_foot", "r_foot")]
    for b in feet:
        assert abs(mj.body_mass[b] - 700.0 * vol) < 1e-9
        assert np.allclose(mj.body_ipos[b], com + c0, atol=1e-9)
        assert np.allclose(np.sort(mj.body_inertia[b]), np.sort(np.linalg.eigvalsh(inertia)), rtol=1e-8, atol=1e-12)
