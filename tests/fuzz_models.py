"""Random small MJCF models for the fuzz parity tests (seeded, deterministic): random kinematic trees with free / ball /
hinge / slide joints, joint limits (ball joints included), damping, armature, springs, motors and position actuators, an optional joint
equality, sphere / capsule geoms over a plane, random solver options.  Only features of the supported table are drawn.
`convex=True` (r04) draws the geoms from spheres, capsules, boxes and small random mesh hulls instead, so that every convex pair
function occurs: plane-convex, convex-convex, sphere-convex and capsule-convex."""
import numpy as np

from ambersim_amd.mjcf import MjModel


def _v(rng, lo, hi, n=3):
    return " ".join(f"{x:.4f}" for x in rng.uniform(lo, hi, n))


def random_mjcf(seed: int, convex: bool = False) -> str:
    rng = np.random.default_rng(seed)
    rng_c = np.random.default_rng(seed + 200003)  # the convex family's geoms (a generator of its own: the plain family keeps its stream)
    meshes = []
    rng_b = np.random.default_rng(seed + 100003)  # ball-joint limits (r03)
    nbody = int(rng.integers(2, 9))
    solver = rng.choice(["Newton", "CG"], p=[0.8, 0.2])
    iters = int(rng.integers(1, 5)) if solver == "Newton" else int(rng.integers(4, 9))
    dt = float(rng.choice([0.002, 0.004, 0.01]))
    out = [f'<mujoco model="fuzz{seed}">',
           f'  <option timestep="{dt}" iterations="{iters}" ls_iterations="{int(rng.integers(3, 9))}" solver="{solver}" gravity="0 0 -9.81"/>',
           '  <compiler angle="radian"/>', '  <worldbody>', '    <geom name="floor" type="plane" size="5 5 0.1"/>']
    parents = [-1] + [int(rng.integers(0, i)) for i in range(1, nbody)]
    if rng.random() < 0.3 and nbody > 3:
        parents[nbody - 1] = -1  # a second tree
    children = {i: [j for j in range(nbody) if parents[j] == i] for i in range(-1, nbody)}
    scalar_joints, ngeom_budget = [], int(rng.integers(0, 4))
    if convex:
        ngeom_budget = int(rng_c.integers(2, 5))
    jid = [0]

    def emit(b, ind):
        root = parents[b] == -1
        pos = _v(rng, -0.2, 0.2) if not root else f"{rng.uniform(-0.5, 0.5):.3f} {rng.uniform(-0.5, 0.5):.3f} {rng.uniform(0.4, 1.0):.3f}"
        if convex and root:  # low over the floor: the geoms start in or near contact with it and with each other
            pos = " ".join(pos.split()[:2]) + f" {rng_c.uniform(0.03, 0.25):.3f}"
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        out.append(f'{ind}<body name="b{b}" pos="{pos}" quat="{" ".join(f"{x:.5f}" for x in q)}">')
        m = rng.uniform(0.2, 2.0)
        d = np.sort(rng.uniform(0.002, 0.02, 3))[::-1] * m
        d[0] = min(d[0], 0.95 * (d[1] + d[2]))
        iq = rng.standard_normal(4)
        iq /= np.linalg.norm(iq)
        out.append(f'{ind}  <inertial pos="{_v(rng, -0.05, 0.05)}" quat="{" ".join(f"{x:.5f}" for x in iq)}" mass="{m:.4f}" diaginertia="{d[0]:.6f} {d[1]:.6f} {d[2]:.6f}"/>')
        kind = rng.choice(["free", "fixed", "joints"], p=[0.5, 0.1, 0.4]) if root else "joints"
        if kind == "free":
            out.append(f'{ind}  <freejoint name="j{jid[0]}"/>')
            jid[0] += 1
        elif kind == "joints":
            for _ in range(int(rng.integers(1, 3))):
                t = rng.choice(["hinge", "slide", "ball"], p=[0.6, 0.25, 0.15])
                at = f'name="j{jid[0]}" type="{t}" pos="{_v(rng, -0.03, 0.03)}"'
                if t != "ball":
                    at += f' axis="{_v(rng, -1, 1)}" damping="{rng.uniform(0, 0.5):.3f}" armature="{rng.uniform(0, 0.02):.4f}"'
                    if rng.random() < 0.6:
                        lo, hi = (-rng.uniform(0.2, 1.5), rng.uniform(0.2, 1.5)) if t == "hinge" else (-rng.uniform(0.02, 0.1), rng.uniform(0.02, 0.1))
                        at += f' range="{lo:.3f} {hi:.3f}"'
                    if rng.random() < 0.3:
                        at += f' stiffness="{rng.uniform(0, 20):.2f}" springref="{rng.uniform(-0.1, 0.1):.3f}"'
                    scalar_joints.append(f"j{jid[0]}")
                else:
                    at += f' damping="{rng.uniform(0, 0.2):.3f}"'
                    if rng_b.random() < 0.5:  # (a generator of its own: the draws of the older features keep their stream)
                        at += f' range="0 {rng_b.uniform(0.3, 1.2):.3f}"'
                out.append(f"{ind}  <joint {at}/>")
                jid[0] += 1
        nonlocal ngeom_budget
        if convex and ngeom_budget > 0 and rng_c.random() < 0.8:
            ngeom_budget -= 1
            kind_g = rng_c.choice(["sphere", "capsule", "box", "mesh"], p=[0.2, 0.25, 0.35, 0.2])
            at = f'name="g{b}" pos="{_v(rng_c, -0.05, 0.05)}" quat="{" ".join(f"{x:.4f}" for x in rng_c.standard_normal(4))}" friction="{rng_c.uniform(0.3, 1.2):.2f}"'
            if kind_g == "sphere":
                at += f' type="sphere" size="{rng_c.uniform(0.04, 0.09):.3f}"'
            elif kind_g == "capsule":
                at += f' type="capsule" size="{rng_c.uniform(0.03, 0.05):.3f} {rng_c.uniform(0.05, 0.12):.3f}"'
            elif kind_g == "box":
                at += f' type="box" size="{_v(rng_c, 0.04, 0.1)}"'
            else:
                pts = rng_c.standard_normal((int(rng_c.integers(5, 13)), 3))
                pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng_c.uniform(0.05, 0.1, (len(pts), 1))
                meshes.append(f'<mesh name="m{b}" vertex="{" ".join(f"{x:.5f}" for x in pts.ravel())}"/>')
                at += f' type="mesh" mesh="m{b}"'
            out.append(f"{ind}  <geom {at}/>")
        elif not convex and ngeom_budget > 0 and rng.random() < 0.7:
            ngeom_budget -= 1
            if rng.random() < 0.5:
                out.append(f'{ind}  <geom name="g{b}" type="sphere" size="{rng.uniform(0.03, 0.08):.3f}" pos="{_v(rng, -0.05, 0.05)}" friction="{rng.uniform(0.3, 1.2):.2f}"/>')
            else:
                out.append(f'{ind}  <geom name="g{b}" type="capsule" size="{rng.uniform(0.02, 0.05):.3f} {rng.uniform(0.04, 0.12):.3f}" pos="{_v(rng, -0.05, 0.05)}" '
                           f'quat="{" ".join(f"{x:.4f}" for x in rng.standard_normal(4))}" friction="{rng.uniform(0.3, 1.2):.2f}"/>')
        for c in children[b]:
            emit(c, ind + "  ")
        out.append(f"{ind}</body>")

    for r in children[-1]:
        emit(r, "    ")
    out.append("  </worldbody>")
    if scalar_joints:
        out.append("  <actuator>")
        for j in scalar_joints:
            r = rng.random()
            if r < 0.4:
                out.append(f'    <motor name="a_{j}" joint="{j}" gear="{rng.uniform(0.5, 3):.2f}" ctrlrange="-1 1"/>')
            elif r < 0.6:
                out.append(f'    <position name="a_{j}" joint="{j}" kp="{rng.uniform(2, 20):.2f}" ctrlrange="-0.5 0.5"/>')
        if out[-1] == "  <actuator>":
            out.pop()
        else:
            out.append("  </actuator>")
    if len(scalar_joints) >= 2 and rng.random() < 0.4:
        a, b = rng.choice(len(scalar_joints), 2, replace=False)
        out.append(f'  <equality><joint name="eq0" joint1="{scalar_joints[a]}" joint2="{scalar_joints[b]}" polycoef="0 {rng.uniform(-1, 1):.3f} {rng.uniform(-0.2, 0.2):.3f} 0 0"/></equality>')
    out.append("</mujoco>")
    if meshes:
        out.insert(3, "  <asset>" + "".join(meshes) + "</asset>")
    return "\n".join(out)


def random_model(seed: int, convex: bool = False):
    """Host model for `seed`, or None when the draw exceeds the device capacity (caller skips it)."""
    m = MjModel.from_xml_string(random_mjcf(seed, convex))
    nlim = int(np.sum(m.jnt_limited))
    if m.nv > 32 or m.neq + nlim + 4 * m.ncon > 64 or getattr(m, "nunsupported_pairs", 0) or m.ncon > (16 if convex else 63):
        return None
    return m


def random_inputs(m, seed: int, nenv: int, nsteps: int):
    rng = np.random.default_rng(1000 + seed)
    q = np.tile(m.qpos0, (nenv, 1))
    for j in range(m.njnt):
        a, t = m.jnt_qposadr[j], m.jnt_type[j]
        if t == 0:  # free: keep the position, random orientation
            quat = rng.standard_normal((nenv, 4))
            q[:, a + 3:a + 7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
        elif t == 1:
            quat = rng.standard_normal((nenv, 4))
            q[:, a:a + 4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
        else:
            q[:, a] += rng.uniform(-0.3, 0.3, nenv) * (1.0 if t == 3 else 0.1)
    v = 0.5 * rng.standard_normal((nenv, m.nv))
    x0 = np.concatenate([q, v], axis=1)
    lo, hi = (m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]) if m.nu else (np.zeros(0), np.zeros(0))
    us = rng.uniform(lo, hi, (nenv, nsteps, m.nu)) if m.nu else np.zeros((nenv, nsteps, 0))
    return x0, us


def random_dense_mjcf(seed: int) -> str:
    """The contact-rich family (r05): what the planner's dense form is for - a Newton model with 9 .. 16 sphere contacts against the floor,
    joint limits, no equality / ball / friction-loss rows; a floating base or a fixed pedestal with a leading hinge chain, an optional spine
    below it, two to five limbs of two to six hinges.  At most 30 dofs."""
    rng = np.random.default_rng(seed + 300007)
    dt = float(rng.choice([0.002, 0.004]))
    out = [f'<mujoco model="dense{seed}">',
           f'  <option timestep="{dt}" iterations="{int(rng.integers(3, 9))}" ls_iterations="{int(rng.integers(4, 11))}" solver="Newton" gravity="0 0 -9.81"/>',
           '  <compiler angle="radian" autolimits="true"/>',
           '  <default><joint type="hinge" damping="0.3" armature="0.01"/><geom contype="0" conaffinity="0" density="600"/>'
           '<default class="c"><geom type="sphere" contype="1" conaffinity="0"/></default></default>',
           '  <worldbody>', '    <geom name="floor" type="plane" size="0 0 0.1" contype="0" conaffinity="1"/>']
    free = rng.random() < 0.7
    budget = 30 - (6 if free else 0)
    joints, spheres = [], [0]
    nsph = int(rng.integers(9, 17))
    jid = [0]

    def hinge(ind):
        ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
        lo, hi = -rng.uniform(0.3, 1.5), rng.uniform(0.3, 1.5)
        name = f"j{jid[0]}"; jid[0] += 1
        joints.append(name)
        out.append(f'{ind}<joint name="{name}" axis="{ax[0]:.4f} {ax[1]:.4f} {ax[2]:.4f}" range="{lo:.3f} {hi:.3f}" damping="{rng.uniform(0.05, 0.6):.3f}"/>')

    def sphere(ind, p):
        if spheres[0] < nsph and rng.random() < p:
            spheres[0] += 1
            out.append(f'{ind}<geom class="c" size="{rng.uniform(0.03, 0.07):.3f}" pos="{_v(rng, -0.04, 0.04)}" friction="{rng.uniform(0.4, 1.1):.2f}"/>')

    def link(ind, length):
        out.append(f'{ind}<geom type="capsule" fromto="0 0 0 0 0 {-length:.3f}" size="{rng.uniform(0.02, 0.04):.3f}"/>')

    hz = rng.uniform(0.5, 0.9)
    out.append(f'    <body name="root" pos="0 0 {hz if free else 0.2 + 0.3 * (hz - 0.5) / 0.4:.3f}">')  # a pedestal low enough for the limbs to reach the floor
    ind = "      "
    if free:
        out.append(f'{ind}<freejoint name="rootj"/>')
    else:
        hinge(ind); budget -= 1
    out.append(f'{ind}<geom type="box" size="0.1 0.08 0.06"/>')
    sphere(ind, 0.8)
    depth = 1
    nspine = int(rng.integers(0, 4)) if free else int(rng.integers(1, 3))
    for s_ in range(nspine):  # a chain every limb may hang from
        if budget < 8: break
        out.append(f'{ind}<body name="spine{s_}" pos="0 0 {-0.08 if s_ else -0.07:.3f}">')
        ind += "  "; depth += 1
        hinge(ind); budget -= 1
        link(ind, 0.08)
        sphere(ind, 0.5)
    nlimb = int(rng.integers(2, 6))
    for l_ in range(nlimb):
        nd = int(rng.integers(2, 7))
        if budget - nd < 0: break
        budget -= nd
        close = 0
        a = rng.uniform(0, 2 * np.pi)
        first = True
        while nd > 0:
            out.append(f'{ind}{"  " * close}<body name="l{l_}_{close}" pos="{0.12 * np.cos(a) if first else 0:.3f} {0.12 * np.sin(a) if first else 0:.3f} {-0.02 if first else -0.16:.3f}">')
            close += 1
            first = False
            k = 2 if (nd >= 2 and rng.random() < 0.3) else 1
            for _ in range(k): hinge(ind + "  " * close)
            nd -= k
            link(ind + "  " * close, 0.16)
            sphere(ind + "  " * close, 0.9)
        while spheres[0] < nsph and rng.random() < 0.5:
            sphere(ind + "  " * close, 1.0)
        for c in range(close, 0, -1): out.append(f'{ind}{"  " * (c - 1)}</body>')
    for d_ in range(depth - 1, 0, -1): out.append("      " + "  " * (d_ - 1) + "</body>")
    out.append("    </body>")
    out.append("  </worldbody>")
    out.append("  <actuator>")
    for j in joints:
        if rng.random() < 0.5: out.append(f'    <motor joint="{j}" gear="{rng.uniform(0.5, 3):.2f}" ctrllimited="true" ctrlrange="-1 1"/>')
        else: out.append(f'    <position joint="{j}" kp="{rng.uniform(5, 30):.1f}" ctrllimited="true" ctrlrange="-0.5 0.5"/>')
    out.append("  </actuator>")
    out.append("</mujoco>")
    return "\n".join(out)


def random_dense_model(seed: int):
    """Host model of the contact-rich family, or None when the draw does not need the dense form's capacity (at most 8 contacts) or exceeds it."""
    m = MjModel.from_xml_string(random_dense_mjcf(seed))
    if m.nv > 30 or m.ncon <= 8 or m.ncon > 16 or m.nbody > 32:
        return None
    return m


def random_dense_inputs(m, oracle, seed: int, nenv: int, nsteps: int):
    """States of the contact-rich family: random joint angles inside their ranges, random base attitude, the lowest sphere 0 .. 35 mm in the floor
    (floating base) - so that several contacts of either row slot are active at the first step."""
    rng = np.random.default_rng(2000 + seed)
    q = np.tile(m.qpos0, (nenv, 1))
    free = m.jnt_type[0] == 0
    for k in range(nenv):
        for j in range(m.njnt):
            if m.jnt_type[j] == 3: q[k, m.jnt_qposadr[j]] = 0.6 * rng.uniform(m.jnt_range[j, 0], m.jnt_range[j, 1])
        if free:
            quat = np.array([1.0, 0, 0, 0]) + 0.3 * rng.standard_normal(4)
            q[k, 3:7] = quat / np.linalg.norm(quat)
            q[k, 2] = 2.0
            oracle.set_state(q[k], np.zeros(m.nv), np.zeros(m.nu)); oracle.forward()
            q[k, 2] -= oracle.get("con_dist").min() + rng.uniform(0.0, 0.035)
    v = 0.5 * rng.standard_normal((nenv, m.nv))
    us = rng.uniform(m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1], (nenv, nsteps, m.nu))
    return np.concatenate([q, v], axis=1), us
