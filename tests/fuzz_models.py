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
