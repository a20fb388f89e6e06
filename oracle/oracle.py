"""ctypes wrapper of the CPU oracle (oracle/amx_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; nothing under ambersim_amd/ imports this module."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from ambersim_amd._desc import AmxModelDesc, build_desc

_DIR = Path(__file__).resolve().parent
_LIBS = {}


def build():
    subprocess.run(["make", "-s", "-C", str(_DIR)], check=True)


def _lib(dtype):
    key = dtype if dtype in ("native", "count") else np.dtype(dtype).name
    if key not in _LIBS:
        name = {"float64": "libamxo_f64.so", "float32": "libamxo_f32.so", "native": "libamxo_f32_native.so", "count": "libamxo_count.so"}[key]
        path = _DIR / name
        if not path.exists():
            build()
        lib = C.CDLL(str(path))
        lib.amxo_model_create.restype = C.c_void_p
        lib.amxo_model_create.argtypes = [C.POINTER(AmxModelDesc)]
        lib.amxo_model_destroy.argtypes = [C.c_void_p]
        lib.amxo_data_create.restype = C.c_void_p
        lib.amxo_data_create.argtypes = [C.c_void_p]
        lib.amxo_data_destroy.argtypes = [C.c_void_p]
        lib.amxo_forward.argtypes = [C.c_void_p, C.c_void_p]
        lib.amxo_step.argtypes = [C.c_void_p, C.c_void_p]
        lib.amxo_set_state.argtypes = [C.c_void_p, C.c_void_p] + [C.c_void_p] * 4
        lib.amxo_get.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        lib.amxo_get.restype = C.c_int
        lib.amxo_model_int.argtypes = [C.c_void_p, C.c_char_p]
        lib.amxo_rollout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.amxo_max_threads.restype = C.c_int
        lib.amxo_model_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.amxo_ls_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.amxo_ls_stats_flush.argtypes = [C.c_void_p, C.c_void_p]
        _LIBS[key] = lib
    return _LIBS[key]


class Oracle:
    """One model + one scratch data block in the chosen precision."""

    def __init__(self, mj_model, dtype=np.float64, variant=None, ls_rule=0, ls_stats=False):
        """variant: None (portable -O2 build), "native" (fp32, -O3 -march=native, `make -C oracle native` on this host) or
        "count" (flop census build: fp64 arithmetic through a counting scalar).
        ls_rule: 0 = MJX's line-search formulation (the statement of record); 1 = the HIP kernel's acceptance rule (the step's
        decrease with the constant terms differenced per row; that decrease is the solver's improvement); 3 = that rule with the
        kernel's arithmetic around it (reciprocal trial points, multiplied-out stopping tests).  ls_stats: count, per line search,
        what either acceptance rule decides (ls_stats())."""
        self.dtype = np.dtype(np.float32 if variant == "native" else (np.float64 if variant == "count" else dtype))
        self.lib = _lib(variant or dtype)
        self.variant = variant
        self.mj = mj_model
        if getattr(mj_model, "nunsupported_pairs", 0):
            raise NotImplementedError("model has collision pairs outside the implemented pair functions")
        self._desc = build_desc(mj_model)
        self.m = self.lib.amxo_model_create(C.byref(self._desc))
        self.d = self.lib.amxo_data_create(self.m)
        self.nq, self.nv, self.nu = mj_model.nq, mj_model.nv, mj_model.nu
        self.ncon = self.lib.amxo_model_int(self.m, b"ncon")
        if ls_rule:
            self.lib.amxo_model_set_int(self.m, b"ls_rule", int(ls_rule))
        if ls_stats:
            self.lib.amxo_model_set_int(self.m, b"ls_stats", 1)
        self.nefc_max = self.lib.amxo_model_int(self.m, b"nefc_max")

    def __del__(self):
        try:
            self.lib.amxo_data_destroy(self.d)
            self.lib.amxo_model_destroy(self.m)
        except Exception:
            pass

    def _p(self, a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=self.dtype)
        self._tmp.append(a)
        return a.ctypes.data

    def set_state(self, qpos=None, qvel=None, ctrl=None, qacc_warmstart=None):
        self._tmp = []
        self.lib.amxo_set_state(self.m, self.d, self._p(qpos), self._p(qvel), self._p(ctrl), self._p(qacc_warmstart))

    def forward(self):
        self.lib.amxo_forward(self.m, self.d)

    def step(self):
        self.lib.amxo_step(self.m, self.d)

    def get(self, name, shape=None):
        buf = np.zeros(1 << 16, dtype=self.dtype)
        n = self.lib.amxo_get(self.m, self.d, name.encode(), buf.ctypes.data, buf.size)
        if n < 0:
            raise KeyError(name)
        out = buf[:n].copy()
        return out.reshape(shape) if shape is not None else out

    def rollout(self, x0, us, nthreads=0, keep_traj=True):
        """x0: [nx] or [nenv,nx]; us: [nenv,N,nu] -> xs [nenv,N+1,nx]."""
        us = np.ascontiguousarray(us, dtype=self.dtype)
        nenv, N = us.shape[0], us.shape[1]
        x0 = np.ascontiguousarray(x0, dtype=self.dtype)
        batched = int(x0.ndim == 2)
        xs = np.zeros((nenv, N + 1, self.nq + self.nv), dtype=self.dtype)
        self.lib.amxo_rollout(self.m, nenv, N, x0.ctypes.data, batched, us.ctypes.data, xs.ctypes.data, int(nthreads))
        return xs

    def ls_stats(self):
        """Counters since construction (ls_stats=True), both acceptance rules evaluated at the end of every line search:
        line searches, steps accepted by MJX's rule, steps accepted by the kernel's rule, accept decisions that differ, Newton
        iterations, solver calls (= env-steps + forwards), line searches where both accept but pick different bracket ends,
        solver calls with at least one differing accept decision, stopping tests (`improvement < tolerance * scale`) that come
        out differently for `prev_cost - cost` and for the line search's own decrease; end_gap_max: the largest relative gap between the two
        ends' decreases among those differing picks (ties, when it is rounding-sized); accept_differ_decrease_max: the largest
        step decrease relative to cost(0) among the differing accept decisions (what the full-cost comparison could not resolve)."""
        self.lib.amxo_ls_stats_flush(self.m, self.d)
        out = (C.c_longlong * 12)()
        gap = (C.c_double * 3)()
        self.lib.amxo_ls_stats(self.m, out, gap)
        k = ("linesearches", "accept_mjx_rule", "accept_kernel_rule", "accept_differ", "newton_iterations", "solves", "end_pick_differ", "solves_affected", "stop_differ")
        r = dict(zip(k, list(out)))
        r["end_gap_max"] = gap[0]
        r["stop_differ_gap_max"] = gap[2]  # largest |(prev_cost - cost) - decrease| / |cost| among the differing stopping tests
        r["accept_differ_decrease_max"] = gap[1]  # largest |decrease| / |cost(0)| among the differing accept decisions
        return r

    def max_threads(self):
        return self.lib.amxo_max_threads()

    def flops_per_step(self, qpos, qvel, ctrl, nsteps=1):
        """Flop census (variant="count" only): arithmetic operations on REALs executed by `nsteps` amxo_step calls from the
        given state, per step.  Single-threaded by construction (the counter is thread-local)."""
        assert self.variant == "count"
        self.lib.amxo_flops_read_and_reset.restype = C.c_double
        self.set_state(qpos, qvel, ctrl, np.zeros(self.nv))
        self.forward()
        self.lib.amxo_flops_read_and_reset()
        for _ in range(nsteps):
            self.step()
        return self.lib.amxo_flops_read_and_reset() / nsteps


def quadcost(xs, us, Q, Qf, R, xg):
    """Explicit restatement of StaticGoalQuadraticCost.cost (reference ambersim/trajopt/cost.py:62-85) in the dtype of xs,
    batched over a leading axis: 0.5*[sum_{t<N}(x_t-xg)'Q(x_t-xg) + (x_N-xg)'Qf(x_N-xg) + sum_t u_t'R u_t]."""
    xs = np.asarray(xs)
    dt = xs.dtype
    e = xs - np.asarray(xg, dtype=dt)
    val = np.einsum("...ti,ij,...tj->...", e[..., :-1, :], np.asarray(Q, dtype=dt), e[..., :-1, :])
    val = val + np.einsum("...i,ij,...j->...", e[..., -1, :], np.asarray(Qf, dtype=dt), e[..., -1, :])
    if us is not None and np.asarray(us).size:
        u = np.asarray(us, dtype=dt)
        val = val + np.einsum("...ti,ij,...tj->...", u, np.asarray(R, dtype=dt), u)
    return (0.5 * val).astype(dt)


def fd_grad_terminal_cost(oracle: "Oracle", x0, us_nominal, Qf, xg, eps, nsamples, coord0=0, nthreads=0):
    """CPU counterpart of amx_fd_grad (BASELINE.json configs[3]): rollout 0 nominal, rollout k adds
    eps[i] to flat control coordinate c = coord0+k-1 (t = c // nu, i = c % nu); J = terminal cost (cost.py:62-85 with
    Q = R = 0), g[k-1] = (J[k]-J[0])/eps[i].  Everything in the oracle's dtype.  Returns (J, g, xfinal)."""
    dt = oracle.dtype
    us_nominal = np.asarray(us_nominal, dtype=dt)
    N, nu = us_nominal.shape
    eps = np.asarray(eps, dtype=dt)
    us = np.tile(us_nominal[None], (nsamples, 1, 1))
    flat = us.reshape(nsamples, N * nu)
    for k in range(1, nsamples):
        c = coord0 + k - 1
        if c < N * nu:
            flat[k, c] = flat[k, c] + eps[c % nu]
    xs = oracle.rollout(np.asarray(x0, dtype=dt), us, nthreads=nthreads)
    xf = xs[:, -1]
    e = xf - np.asarray(xg, dtype=dt)
    J = (0.5 * np.einsum("ki,ij,kj->k", e, np.asarray(Qf, dtype=dt), e)).astype(dt)
    g = np.zeros(nsamples - 1, dtype=dt)
    for k in range(1, nsamples):
        c = coord0 + k - 1
        if c < N * nu:
            g[k - 1] = (J[k] - J[0]) / eps[c % nu]
    return J, g, xf
