"""BASELINE.json configs[3] (C4): 1024 rollouts x 64 knots, forward-difference gradient of the terminal cost wrt ctrl.

Definition: SURVEY.md §8(d) "C4 inputs" — rollout 0 nominal, rollout k perturbs one control coordinate by
eps = 1e-3 * ctrlrange, g_k = (J_k - J_0)/eps, J = 0.5 (x_N - xg)' Qf (x_N - xg) (reference ambersim/trajopt/cost.py:62-85
with Q = R = 0, on the output of shoot, ambersim/trajopt/shooting.py:22-48).  The reference has no FD code of its own.

CPU part: the oracle's FD restatement against the tangent-linear recursion of the reference pendulum's closed form.
GPU part (through the C ABI, amx_fd_grad): J_k against the fp64 oracle, g against the oracle's own fp32 and fp64 FD."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def c4_case(name):
    """Seeded C4 inputs: shared x0, nominal controls well inside the ctrl range, goal = initial pose at rest."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    N = 64
    if name == "bh280":
        mj = make_golden.bh280_fixture_model()
        rng = np.random.default_rng(41)
        x0 = np.concatenate([np.array([0.4, 0.6, 0.2, 0.4, 0.7, 0.25, 0.8, 0.3])[:mj.nq], np.zeros(mj.nv)])
        us = 2.0 * rng.standard_normal((N, mj.nu))
        xg = np.concatenate([np.full(mj.nq, 0.5), np.zeros(mj.nv)])
    else:
        # biped_flight: the biped dropped from 0.45 m above its standing height - no contact within the 64 steps, smooth
        # dynamics, tight tolerances.  biped_stance: the standing keyframe through its first foot impact - MJX's 4-point
        # manifold selection on a flat foot is discontinuous in the state (a 1e-9 perturbation of the fp64 oracle's input moves
        # the state by 0.1 at the impact step), so only order-of-magnitude statements hold there.
        base = name.split("_")[0]
        mj = load_mj_model_from_file({"biped": "models/biped/biped.xml", "quadruped": "models/quadruped/quadruped.xml"}[base])
        rng = np.random.default_rng(42)
        x0 = np.concatenate([mj.key_qpos[0], np.zeros(mj.nv)])
        x0[7:mj.nq] += 0.03 * rng.standard_normal(mj.nq - 7)
        x0[mj.nq:] += 0.05 * rng.standard_normal(mj.nv)
        if name.endswith("_flight"):
            x0[2] += 0.45
        lo, hi = mj.actuator_ctrlrange[:, 0], mj.actuator_ctrlrange[:, 1]
        us = 0.5 * (lo + hi) + 0.25 * (hi - lo) * rng.uniform(-1, 1, (N, mj.nu))
        xg = x0.copy()
        xg[2] += 0.05  # goal: the initial pose at rest, slightly higher
    nx = mj.nq + mj.nv
    Qf = np.diag(np.concatenate([np.full(mj.nq, 10.0), np.full(mj.nv, 0.1)]))
    Qf[0, 1] = Qf[1, 0] = 0.5  # not diagonal: the full quadratic form is exercised
    eps = 1e-3 * (mj.actuator_ctrlrange[:, 1] - mj.actuator_ctrlrange[:, 0])
    return mj, x0, us, Qf, xg, eps


def test_oracle_fd_matches_pendulum_tangent():
    """d(terminal cost)/d(u_t) of the reference pendulum (models/pendulum/pendulum.xml:3-23: M = 0.337959, m g l = 4.905,
    dt = 0.02, ctrl inside its +-2 clamp) by the tangent-linear recursion of its semi-implicit Euler closed form; the fp64
    oracle's forward differences converge to it at O(eps)."""
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    from oracle.oracle import Oracle, fd_grad_terminal_cost
    mj = load_mj_model_from_file("models/pendulum/pendulum.xml")
    N, dt, M, mgl = 40, 0.02, 0.337959, 4.905
    rng = np.random.default_rng(5)
    us = rng.uniform(-1.5, 1.5, (N, 1))
    x0 = np.array([0.7, -0.3])
    xg = np.array([np.pi, 0.0])
    Qf = np.array([[3.0, 0.4], [0.4, 0.5]])
    th, om = [x0[0]], [x0[1]]
    for t in range(N):
        om.append(om[-1] + dt * (us[t, 0] - mgl * np.sin(th[-1])) / M)
        th.append(th[-1] + dt * om[-1])
    lam = Qf @ (np.array([th[-1], om[-1]]) - xg)  # dJ/dx_N
    g = np.zeros(N)
    for t in range(N - 1, -1, -1):
        # x_{t+1} = f(x_t, u_t): om' = om + dt (u - mgl sin th)/M, th' = th + dt om'
        dom_du = dt / M
        g[t] = lam[1] * dom_du + lam[0] * dt * dom_du
        a = -dt * mgl * np.cos(th[t]) / M  # d om'/d th
        A = np.array([[1 + dt * a, dt], [a, 1.0]])  # d(th', om')/d(th, om)
        lam = A.T @ lam
    o = Oracle(mj, np.float64)
    errs = []
    for e in (1e-3, 1e-5):
        J, gfd, xf = fd_grad_terminal_cost(o, x0, us, Qf, xg, np.array([e]), N + 1)
        assert abs(xf[0, 0] - th[-1]) < 1e-6 and abs(xf[0, 1] - om[-1]) < 1e-6  # the oracle's pendulum is the closed form (to the inertia's 6 digits)
        errs.append(np.abs(gfd - g).max() / np.abs(g).max())
    assert errs[1] < 2e-5 and errs[1] < errs[0], errs  # limited by the 6-digit inertia of the closed form, not by eps


def test_oracle_fd_ignores_coordinates_beyond_the_horizon():
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    from oracle.oracle import Oracle, fd_grad_terminal_cost
    mj = load_mj_model_from_file("models/pendulum/pendulum.xml")
    us = np.zeros((3, 1))
    J, g, xf = fd_grad_terminal_cost(Oracle(mj, np.float64), np.array([0.2, 0.0]), us, np.eye(2), np.zeros(2), np.array([1e-3]), 6, coord0=1)
    assert g[0] != 0 and g[1] != 0 and np.all(g[2:] == 0) and np.all(J[3:] == J[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name,tolJ", [("biped_flight", 1e-5), ("bh280", 1e-5), ("biped_stance", None)])
def test_c4_fd_gradient_vs_oracle(name, tolJ):
    """C4 at its full size (1024 rollouts x 64 knots) through amx_fd_grad.
    J_k: relative to the fp64 oracle <= tolJ after the 64-step horizon in fp32: 1e-5 for the biped in flight and for the Barrett
    hand (measured 1.3e-6 and 3.1e-6, on or below the fp32 CPU oracle's own 1.5e-6 / 4.0e-6; r02 stood at 7e-5 on the hand with a
    2e-4 bound - stiff equality rows on 1e-6 kg m^2 finger inertias, and a line search that dropped steps whose improvement fell
    below the rounding of the full cost: DESIGN.md section 4e).  g: forward differences of an fp32 rollout carry
    rounding noise of order ulp(J)/eps, so fp64 is the wrong yardstick for individual entries; the norm-wise distance of
    the GPU gradient to the fp64 FD must be within 2x the fp32 oracle's own distance (+1e-3), cosine > 0.999.
    biped_stance runs through a foot impact (discontinuous manifold selection, see c4_case): finite, deterministic, and
    the same order of error as the fp32 CPU oracle - nothing tighter is meaningful there."""
    from ambersim_amd import mjx
    from ambersim_amd.trajopt.fd import fd_grad_terminal_cost
    from oracle.oracle import Oracle, fd_grad_terminal_cost as fd_oracle
    mj, x0, us, Qf, xg, eps = c4_case(name)
    dev = torch.device("cuda", 0)
    model = mjx.put_model(mj, dev)
    S = 1024
    args = (model, torch.tensor(x0), torch.tensor(us), torch.tensor(Qf), torch.tensor(xg), torch.tensor(eps))
    r = fd_grad_terminal_cost(*args, nsamples=S)
    torch.cuda.synchronize()
    J, g = r.J.cpu().numpy().astype(np.float64), r.grad.cpu().numpy().astype(np.float64)
    J64, g64, xf64 = fd_oracle(Oracle(mj, np.float64), x0, us, Qf, xg, eps, S)
    J32, g32, xf32 = fd_oracle(Oracle(mj, np.float32), x0, us, Qf, xg, eps, S)
    assert np.isfinite(J).all() and np.isfinite(g).all()
    errJ = np.abs(J - J64).max() / np.abs(J64).max()
    floorJ = np.abs(J32 - J64).max() / np.abs(J64).max()
    errx = np.abs(r.xfinal.cpu().numpy() - xf64).max() / max(1.0, np.abs(xf64).max())
    n64 = np.linalg.norm(g64)
    errg, floorg = np.linalg.norm(g - g64) / n64, np.linalg.norm(g32.astype(np.float64) - g64) / n64
    cos = float(g @ g64 / (np.linalg.norm(g) * n64))
    print(f"\nC4 {name}: J0={J64[0]:.6g} |g|={n64:.4g}  J rel err gpu {errJ:.2e} (fp32 oracle {floorJ:.2e})  xN rel err {errx:.2e}  "
          f"g normwise err gpu {errg:.2e} (fp32 oracle {floorg:.2e})  cos(g_gpu, g_fp64) {cos:.6f}")
    ncoord = min(S - 1, us.size)
    assert np.all(g[ncoord:] == 0)
    r2 = fd_grad_terminal_cost(*args, nsamples=S)
    assert torch.equal(r.J, r2.J) and torch.equal(r.grad, r2.grad)  # deterministic
    if tolJ is None:
        assert errJ < max(5 * floorJ, 0.5), (errJ, floorJ)
        return
    assert errJ < tolJ, (errJ, floorJ)
    assert errg < 2 * floorg + 1e-3, (errg, floorg)
    assert cos > 0.999


@pytest.mark.gpu
def test_wide_plan_is_opt_in_and_agrees(monkeypatch):
    """The biped also has a WIDE plan (the same model on a 64-lane group, one environment per wavefront, static kernels
    Spec_biped_w64) meant for batches that cannot give every SIMD a two-environment wave (C4: 1024 rollouts = 512 waves on 1024 SIMDs).
    It measured slower than the regular plan (profiles/r04_c4_fd_grad.txt), so it is opt-in: AMX_WIDE_SMALL_BATCH=1 at model creation
    builds its device image (r05: not otherwise), and the same variable, read per call, selects it; by default every batch size runs the
    same kernels.  Here: the plan exists when asked for, the switch selects it (through a foot impact the two lane layouts round
    differently), and the two agree to rounding after one step and stay finite."""
    from ambersim_amd import mjx
    mj, x0, us, Qf, xg, eps = c4_case("biped_stance")
    dev = torch.device("cuda", 0)
    assert mjx.put_model(mj, dev)._h.query("wide_plan") == 0
    monkeypatch.setenv("AMX_WIDE_SMALL_BATCH", "1")
    model = mjx.put_model(mj, dev)
    monkeypatch.delenv("AMX_WIDE_SMALL_BATCH")
    assert model._h.query("wide_plan") == 1 and model._h.query("lanes_per_env") == 32
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    usp = t(us).unsqueeze(0).repeat(256, 1, 1) + 0.01 * torch.randn((256,) + us.shape, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
    regular = mjx.rollout(model, t(x0), usp)
    monkeypatch.setenv("AMX_WIDE_SMALL_BATCH", "1")
    wide = mjx.rollout(model, t(x0), usp)
    monkeypatch.delenv("AMX_WIDE_SMALL_BATCH")
    again = mjx.rollout(model, t(x0), usp)
    assert torch.equal(regular, again) and bool(torch.isfinite(wide).all())
    assert not torch.equal(regular[:, -1], wide[:, -1])  # another kernel: the switch took effect
    err = (regular[:, 1] - wide[:, 1]).abs().max(dim=1).values / regular[:, 1].abs().max(dim=1).values.clamp(min=1.0)
    assert float(err.max()) < 1e-5, float(err.max())


@pytest.mark.gpu
def test_c4_api_consistency():
    """amx_fd_grad is exactly perturb -> rollout -> terminal cost: its J equals amx_quadcost on trajectories of the
    explicitly perturbed controls bit for bit; coord0 windows tile the full gradient; bad shapes raise ValueError."""
    from ambersim_amd import mjx
    from ambersim_amd.trajopt.cost import StaticGoalQuadraticCost
    from ambersim_amd.trajopt.fd import fd_grad_all, fd_grad_terminal_cost
    mj, x0, us, Qf, xg, eps = c4_case("quadruped")
    us = us[:8]
    dev = torch.device("cuda", 0)
    model = mjx.put_model(mj, dev)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    S = 40
    r = fd_grad_terminal_cost(model, t(x0), t(us), t(Qf), t(xg), t(eps), nsamples=S, coord0=5)
    usp = t(us).unsqueeze(0).repeat(S, 1, 1)
    flat = usp.view(S, -1)
    for k in range(1, S):
        flat[k, 5 + k - 1] += t(eps)[(5 + k - 1) % mj.nu]
    xs = mjx.rollout(model, t(x0), usp)
    assert torch.equal(xs[:, -1], r.xfinal)
    nx = mj.nq + mj.nv
    cost = StaticGoalQuadraticCost(torch.zeros(nx, nx), t(Qf), torch.zeros(mj.nu, mj.nu), t(xg))
    Jq, _ = cost.cost(xs, usp)
    assert torch.allclose(Jq, r.J, rtol=2e-6, atol=0)  # same terminal form; the tiled kernel sums in another order
    assert torch.equal((r.J[1:] - r.J[0]) / t(eps)[(5 + torch.arange(S - 1, device=dev)) % mj.nu], r.grad)
    full = fd_grad_all(model, t(x0), t(us), t(Qf), t(xg), t(eps), nsamples=33)
    assert full.shape == (8, mj.nu)
    assert torch.equal(full.reshape(-1)[5:5 + S - 1], r.grad)
    with pytest.raises(ValueError):
        fd_grad_terminal_cost(model, t(x0)[:-1], t(us), t(Qf), t(xg), t(eps))
    with pytest.raises(ValueError):
        fd_grad_terminal_cost(model, t(x0), t(us), t(Qf), t(xg), torch.zeros(mj.nu))
