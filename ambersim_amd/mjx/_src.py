from __future__ import annotations

import copy
import ctypes as C
import os
import dataclasses
from typing import Any, Optional

import numpy as np
import torch

from .. import _lib
from .._desc import build_desc
from ..mjcf import DisableBit, MjData, MjModel, recompile_options


class SolverType:
    PGS, CG, NEWTON = 0, 1, 2


def _dev(device=None) -> torch.device:
    if device is not None:
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        return device
    if not torch.cuda.is_available():
        raise RuntimeError("ambersim_amd.mjx needs a GPU (MI355X): no HIP device is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream_ptr(device: torch.device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


@dataclasses.dataclass(frozen=True)
class Option:
    """Counterpart of mjx.Option; `.replace` as used at reference tests/trajopt/test_predictive_sampler.py:22-31."""
    timestep: float
    iterations: int
    ls_iterations: int
    tolerance: float
    ls_tolerance: float
    impratio: float
    gravity: Any
    integrator: int
    solver: int
    disableflags: int
    cone: int = 0

    def replace(self, **kw) -> "Option":
        return dataclasses.replace(self, **kw)


class _Handle:
    """Owns one amx_model* (freed with the last Model that shares it) and the library build it lives in: the default
    libamx.so, or - `specialise=True` - a build of csrc/ compiled for exactly this model's launch plan (ambersim_amd/_spec.py)."""

    def __init__(self, mj: MjModel, device_index: int, specialise: bool = False):
        self.desc = build_desc(mj)
        self._ptr = C.c_void_p()
        self.refuse = None  # set by Model: why this handle must not be stepped (see Model.__init__)
        self.L = _lib.lib()
        needs_build = False
        try:
            _lib.check(self.L.amx_model_create(C.byref(self.desc), device_index, C.byref(self._ptr)), self.L)
        except NotImplementedError as e:
            # a model only per-model kernels can run (two contact rows per lane): no generic handle exists; build unless told not to
            if "needs per-model kernels" not in str(e) or specialise is False:
                raise
            needs_build = True
        if needs_build or (specialise and not self.query("specialised")):
            # specialise: True = the caller asked for a per-model build (errors propagate); "auto" = the default: build one when hipcc
            # is on this machine, say so in one line, and run the generic kernels (about half the throughput) when it is not
            import logging
            from .._spec import build_specialised_library, specialised_library_path
            log = logging.getLogger("ambersim_amd")
            try:
                if specialise == "auto":
                    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
                    prebuilt = specialised_library_path(mj)[0].exists()  # e.g. by `python -m ambersim_amd.prebuild model.xml` on a build host
                    if not os.path.exists(hipcc) and not prebuilt:
                        raise FileNotFoundError(f"{hipcc} not found (and no prebuilt library for this model under {specialised_library_path(mj)[0].parent}: "
                                                f"`python -m ambersim_amd.prebuild <model>` makes one where hipcc is installed)")
                    if not prebuilt:
                        log.warning("no kernels are compiled for this model's launch plan: building them with hipcc (about ten seconds, once; "
                                    "cached under %s; AMX_SPECIALISE=0 keeps the generic kernels)", specialised_library_path(mj)[0].parent)
                L2 = _lib.load(build_specialised_library(mj))
                ptr2 = C.c_void_p()
                _lib.check(L2.amx_model_create(C.byref(self.desc), device_index, C.byref(ptr2)), L2)
                if self._ptr:
                    self.L.amx_model_destroy(self._ptr)
                self.L, self._ptr = L2, ptr2
            except Exception as e:
                if needs_build and isinstance(e, (FileNotFoundError, RuntimeError)) and "hipcc" in str(e):
                    # the ABI's documented answer to "this model cannot run here": NotImplementedError, with the original reason
                    raise NotImplementedError(f"this model needs per-model kernels (two contact rows per lane) and they cannot be built on this "
                                              f"machine: {e}") from e
                if specialise != "auto" or needs_build:
                    raise
                log.warning("per-model kernels unavailable (%s: %s): this model runs on the generic kernels, at about half the throughput of a "
                            "per-model build", type(e).__name__, str(e).splitlines()[0][:160] if str(e) else "")

    def query(self, name: str) -> int:
        out = C.c_int()
        _lib.check(self.L.amx_model_query(self._ptr, name.encode(), C.byref(out)), self.L)
        return out.value

    @property
    def ptr(self):
        """The amx_model* every compute entry point passes to the library.  A model whose contact the engine cannot run (see
        Model.__init__) is refused here, at its first use, with the exception type `mjx.device_put` raises for unsupported features."""
        if self.refuse:
            raise NotImplementedError(self.refuse)
        return self._ptr

    def __del__(self):
        try:
            if self._ptr:
                self.L.amx_model_destroy(self._ptr)
        except Exception:
            pass


class Model:
    """Device model.  Immutable; `replace(opt=...)` re-uploads the constant block."""

    def __init__(self, mj: MjModel, device: torch.device, opt: Optional[Option] = None, specialise: bool = False):
        self.mj_model = mj
        self.specialise = specialise
        self.device = device
        o = mj.opt
        self.opt = opt if opt is not None else Option(
            timestep=float(o.timestep), iterations=int(o.iterations), ls_iterations=int(o.ls_iterations),
            tolerance=float(o.tolerance), ls_tolerance=float(o.ls_tolerance), impratio=float(o.impratio),
            gravity=tuple(float(g) for g in o.gravity), integrator=int(o.integrator), solver=int(o.solver),
            disableflags=int(o.disableflags))
        self.nq, self.nv, self.nu, self.nbody, self.njnt, self.ngeom, self.neq = mj.nq, mj.nv, mj.nu, mj.nbody, mj.njnt, mj.ngeom, mj.neq
        self.actuator_ctrlrange = torch.as_tensor(np.asarray(mj.actuator_ctrlrange, dtype=np.float32).reshape(-1, 2), device=device)
        self.qpos0 = torch.as_tensor(np.asarray(mj.qpos0, dtype=np.float32), device=device)
        # Collision pairs outside the implemented pair functions (the reference's full Barrett hand: 2000+ mesh-mesh pairs of hulls with
        # hundreds of vertices).  MJX puts such a model on the device and the reference then switches its contact off
        # (tests/trajopt/test_predictive_sampler.py:14-31: load, `model.replace(opt=...disableflags=DisableBit.CONTACT)`, shoot), so the
        # model must survive `put_model` for that flow to work: it is planned without its contact, and every entry point that would step
        # it WITH contact raises NotImplementedError (reference io_utils.py:228-241's exception type) - `replace(opt=...)` with contact
        # disabled gives a model that runs.
        refuse = None
        if getattr(mj, "nunsupported_pairs", 0) and not (int(self.opt.disableflags) & (DisableBit.CONTACT | DisableBit.CONSTRAINT)):
            refuse = (f"{mj.nunsupported_pairs} collision pairs are outside the implemented pair functions (condim != 3, a cylinder or ellipsoid, a "
                      "convex hull beyond 64 vertices / 128 faces / 192 edges - `maxhullvert` caps a mesh's hull): this model cannot be stepped "
                      "with contact enabled; model.replace(opt=model.opt.replace(disableflags=DisableBit.CONTACT)), restrict contype / "
                      "conaffinity or simplify the meshes")
            import logging
            logging.getLogger("ambersim_amd").warning("put_model: %s", refuse)
            plan_mj = copy.copy(mj)
            plan_mj.opt = mj.opt.replace(disableflags=int(mj.opt.disableflags) | DisableBit.CONTACT)
            mj_for_plan = recompile_options(plan_mj)
        else:
            mj_for_plan = mj
        self._h = _Handle(mj_for_plan, device.index or 0, specialise)
        self._h.refuse = refuse
        self.ncon = self._h.query("ncon")
        self.nefc_max = self._h.query("nefc_max")

    # parameter arrays `replace` accepts beside `opt` (the reference itself only replaces `opt`, tests/trajopt/test_predictive_sampler.py:22-31;
    # these are what domain randomisation over an mjx.Model touches): the array goes into a copy of the host model as it is - like
    # mjx.Model.replace, nothing derived from it (invweight0, pair friction of explicit <pair>s) is recomputed, except the collision
    # pair table, which is rebuilt from the geoms - and the constant block is uploaded again.  Same shape, no structural change.
    REPLACEABLE = ("body_mass", "body_inertia", "body_pos", "body_quat", "body_ipos", "body_iquat", "body_invweight0", "jnt_pos", "jnt_axis",
                   "jnt_range", "jnt_margin", "jnt_stiffness", "jnt_solref", "jnt_solimp", "qpos0", "qpos_spring", "dof_armature", "dof_damping",
                   "dof_invweight0", "dof_frictionloss", "dof_solref", "dof_solimp", "geom_pos", "geom_quat", "geom_size", "geom_friction",
                   "geom_solref", "geom_solimp", "geom_margin", "actuator_gear", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange",
                   "actuator_forcerange", "eq_data", "eq_solref", "eq_solimp")

    def replace(self, **kw) -> "Model":
        other = set(kw) - {"opt"}
        if other - set(self.REPLACEABLE):
            raise NotImplementedError(f"Model.replace supports opt= and the parameter arrays {', '.join(self.REPLACEABLE)} (got {sorted(other - set(self.REPLACEABLE))})")
        opt: Option = kw.get("opt", self.opt)
        mj = copy.copy(self.mj_model)
        for k in sorted(other):
            old = np.asarray(getattr(self.mj_model, k))
            v = kw[k]
            v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            if v.shape != old.shape:
                raise ValueError(f"Model.replace({k}=...): shape {v.shape}, the model's is {old.shape}")
            setattr(mj, k, v.astype(old.dtype, copy=True))
        mj.opt = self.mj_model.opt.replace(
            timestep=float(opt.timestep), iterations=int(opt.iterations), ls_iterations=int(opt.ls_iterations),
            tolerance=float(opt.tolerance), ls_tolerance=float(opt.ls_tolerance), impratio=float(opt.impratio),
            gravity=np.asarray(opt.gravity, dtype=float), integrator=int(opt.integrator), solver=int(opt.solver),
            disableflags=int(opt.disableflags))
        mj = recompile_options(mj)
        return Model(mj, self.device, opt, self.specialise)


_DATA_FIELDS = ("qpos", "qvel", "ctrl", "qacc", "qacc_warmstart", "time", "qacc_smooth", "qfrc_bias", "qfrc_constraint",
                "xpos", "xquat")


class Data:
    """Physics state + derived quantities (subset of mjx.Data the reference reads: qpos, qvel, ctrl,
    reference ambersim/rl/pendulum/swingup.py:64-76, ambersim/trajopt/shooting.py:42)."""
    __slots__ = _DATA_FIELDS

    def __init__(self, **kw):
        for f in _DATA_FIELDS:
            setattr(self, f, kw.get(f))

    def replace(self, **kw) -> "Data":
        out = Data(**{f: getattr(self, f) for f in _DATA_FIELDS})
        for k, v in kw.items():
            if k not in _DATA_FIELDS:
                raise AttributeError(f"Data has no field '{k}'")
            ref = getattr(self, k)
            if ref is not None and not torch.is_tensor(v):
                v = torch.as_tensor(v, dtype=torch.float32, device=ref.device)
            elif torch.is_tensor(v) and ref is not None:
                v = v.to(device=ref.device, dtype=torch.float32)
            setattr(out, k, v)
        return out

    @property
    def batched(self) -> bool:
        return self.qpos.dim() == 2


def put_model(mj: MjModel, device=None, specialise: Optional[bool] = None) -> Model:
    """Host model -> device model.  `specialise=True`: if the library holds no kernels compiled for this model's launch plan
    (only the shipped models have them), compile csrc/ for it with hipcc - about ten seconds, cached by plan - and use that build:
    static loop bounds, immediate LDS offsets, tree-sparse eliminations, as for the shipped models.  Default (None): do so whenever
    hipcc is available, fall back to the generic kernels with one log line when it is not; False: always the library's own kernels."""
    if mj.opt.integrator != 0:
        raise NotImplementedError("only the Euler integrator is implemented")
    if specialise is None:
        # default (r04): "auto" - a model the library ships no kernels for gets a per-model build when hipcc is on this machine
        # (the generic kernels run at about half the throughput: bench.py, "C3 generic kernels"); AMX_SPECIALISE=0 / 1 forces never / always
        env = os.environ.get("AMX_SPECIALISE", "auto")
        specialise = "auto" if env in ("", "auto") else env != "0"
    return Model(mj, _dev(device), specialise=specialise)


def make_data(m: Model, nenv: Optional[int] = None) -> Data:
    """mjx.make_data: zero state at qpos0 (reference ambersim/utils/io_utils.py:226).  `nenv` adds a batch axis."""
    dev = m.device
    shape = (lambda n: (n,)) if nenv is None else (lambda n: (nenv, n))
    z = lambda n: torch.zeros(shape(n), dtype=torch.float32, device=dev)
    qpos = m.qpos0.clone() if nenv is None else m.qpos0.expand(nenv, -1).clone()
    time = torch.zeros((), dtype=torch.float32, device=dev) if nenv is None else torch.zeros(nenv, dtype=torch.float32, device=dev)
    return Data(qpos=qpos, qvel=z(m.nv), ctrl=z(m.nu), qacc=z(m.nv), qacc_warmstart=z(m.nv), time=time,
                qacc_smooth=z(m.nv), qfrc_bias=z(m.nv), qfrc_constraint=z(m.nv),
                xpos=torch.zeros(shape(m.nbody) + (3,), dtype=torch.float32, device=dev),
                xquat=torch.zeros(shape(m.nbody) + (4,), dtype=torch.float32, device=dev))


def _host_data(d: MjData, dev: torch.device, nbody: int) -> Data:
    f = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)
    nv = int(np.asarray(d.qvel).shape[-1])
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    return Data(qpos=f(d.qpos), qvel=f(d.qvel), ctrl=f(d.ctrl), qacc=f(d.qacc), qacc_warmstart=f(d.qacc_warmstart),
                time=torch.tensor(float(d.time), dtype=torch.float32, device=dev), qacc_smooth=z(nv), qfrc_bias=z(nv),
                qfrc_constraint=z(nv), xpos=z(nbody, 3), xquat=z(nbody, 4))


def put_data(m: Model, d: MjData) -> Data:
    return _host_data(d, m.device, m.nbody)


def device_put(x, device=None):
    """mjx.device_put: a host MjModel -> Model (reference ambersim/rl/base.py:52, ambersim/utils/io_utils.py:225); a host
    MjData -> Data on the device (reference ambersim/rl/base.py:83, inside MjxEnv.pipeline_init)."""
    if isinstance(x, MjModel):
        return put_model(x, device)
    if isinstance(x, MjData):
        mj = getattr(x, "_model", None)
        if mj is None:
            raise ValueError("device_put(MjData): this MjData was not created from an MjModel (MjData(model))")
        return _host_data(x, _dev(device), mj.nbody)
    raise NotImplementedError(f"device_put accepts a host MjModel or MjData, got {type(x).__name__}")


def _c(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if not t.is_contiguous() else t


def _prep(m: Model, d: Data, names):
    """Validates the Data leaves a kernel will read against the model and returns them as contiguous fp32 [n, width]
    tensors on the model's device.  The batch size is the leading axis of qpos; a leaf without a batch axis (e.g. the
    zero `ctrl` the reference's pipeline_init sets, ambersim/rl/base.py:84) is broadcast to it.  Anything else - wrong
    trailing size, another batch size, another device - is a ValueError here instead of an out-of-bounds access there."""
    width = {"qpos": m.nq, "qvel": m.nv, "ctrl": m.nu, "qacc_warmstart": m.nv}
    qpos = d.qpos
    if not torch.is_tensor(qpos) or qpos.dim() not in (1, 2):
        raise ValueError("Data.qpos must be a tensor of shape [nq] or [nenv, nq]")
    batched = qpos.dim() == 2
    n = qpos.shape[0] if batched else 1
    out = {}
    for name in names:
        t = getattr(d, name)
        if not torch.is_tensor(t):
            raise ValueError(f"Data.{name} must be a tensor")
        if t.device != m.device:
            raise ValueError(f"Data.{name} lives on {t.device}, the model on {m.device}")
        if t.dim() not in (1, 2) or t.shape[-1] != width[name]:
            raise ValueError(f"Data.{name} has shape {tuple(t.shape)}, expected [..., {width[name]}]")
        if t.dim() == 2 and t.shape[0] != n:
            raise ValueError(f"Data.{name} has batch {t.shape[0]}, qpos has {n}")
        t = t.to(torch.float32)
        if t.dim() == 1:
            t = t.unsqueeze(0).expand(n, -1)
        out[name] = _c(t)
    return batched, n, out


def forward(m: Model, d: Data) -> Data:
    """mjx.forward (reference ambersim/rl/base.py:85, ambersim/trajopt/shooting.py:36): derived quantities, no integration."""
    b, n, t = _prep(m, d, ("qpos", "qvel", "ctrl", "qacc_warmstart"))
    qpos, qvel, ctrl, warm = t["qpos"], t["qvel"], t["ctrl"], t["qacc_warmstart"]
    dev = m.device
    new = lambda *s: torch.empty((n,) + s, dtype=torch.float32, device=dev)
    qacc, qs, fb, fc, xpos, xquat = new(m.nv), new(m.nv), new(m.nv), new(m.nv), new(m.nbody, 3), new(m.nbody, 4)
    out = _lib.AmxForwardOut(qacc.data_ptr(), qs.data_ptr(), fb.data_ptr(), fc.data_ptr(), None, xpos.data_ptr(), xquat.data_ptr(), None, None)
    with torch.cuda.device(dev):
        _lib.check(m._h.L.amx_forward(m._h.ptr, n, qpos.data_ptr(), qvel.data_ptr(), ctrl.data_ptr(), warm.data_ptr(),
                                          C.byref(out), _stream_ptr(dev)), m._h.L)
    s = (lambda t: t) if b else (lambda t: t[0])
    # the solver's result becomes the next warm start, as in the reference physics
    warm_out = qacc if m.nefc_max > 0 else warm
    # every leaf of a batched Data carries the batch axis (as under jax.vmap), `time` included: a later step() returns [n] and the
    # structure of the record must not change between calls (graph capture of the env step keeps persistent buffers per leaf)
    time = d.time
    if b and torch.is_tensor(time) and time.dim() == 0:
        time = time.to(device=dev, dtype=torch.float32).expand(n).clone()
    return d.replace(qacc=s(qacc), qacc_smooth=s(qs), qfrc_bias=s(fb), qfrc_constraint=s(fc), xpos=s(xpos), xquat=s(xquat),
                     qacc_warmstart=s(warm_out), qvel=s(qvel), ctrl=s(ctrl), time=time)


def constraint_activity(m: Model, qpos: torch.Tensor, qvel: torch.Tensor, ctrl: torch.Tensor):
    """What the constraint set of a batch of states looks like: one amx_forward launch, returning
    (contact_dist [n, ncon] - the signed distance of every contact slot, >= 0 or the inactive marker when the slot holds no
    contact - and nefc [n] - rows that entered the solve, i.e. D > 0: violated limits, equality rows, the pyramid rows of
    contacts inside their margin).  bench.py reports these beside the throughput (the kernel skips the matrix-core work of rows
    that are idle in both environments of a wave, so cost depends on them)."""
    d = Data(qpos=qpos, qvel=qvel, ctrl=ctrl, qacc=None, qacc_warmstart=torch.zeros_like(qvel), time=None, qacc_smooth=None, qfrc_bias=None,
             qfrc_constraint=None, xpos=None, xquat=None)
    _, n, t = _prep(m, d, ("qpos", "qvel", "ctrl", "qacc_warmstart"))
    dev = m.device
    dist = torch.empty((n, max(1, m.ncon)), dtype=torch.float32, device=dev)
    nefc = torch.empty((n,), dtype=torch.int32, device=dev)
    out = _lib.AmxForwardOut(None, None, None, None, None, None, None, dist.data_ptr(), nefc.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(m._h.L.amx_forward(m._h.ptr, n, t["qpos"].data_ptr(), t["qvel"].data_ptr(), t["ctrl"].data_ptr(), t["qacc_warmstart"].data_ptr(),
                                          C.byref(out), _stream_ptr(dev)), m._h.L)
    return dist[:, :m.ncon], nefc


def step(m: Model, d: Data, nsub: int = 1) -> Data:
    """mjx.step, `nsub` times with ctrl held (reference ambersim/rl/base.py:88-96).  Functional: returns new Data."""
    b, n, t = _prep(m, d, ("qpos", "qvel", "ctrl", "qacc_warmstart"))
    ctrl = t["ctrl"]
    time = d.time if torch.is_tensor(d.time) else torch.tensor(float(d.time or 0.0))
    time = time.to(device=m.device, dtype=torch.float32).reshape(-1)
    if time.numel() not in (1, n):
        raise ValueError(f"Data.time has {time.numel()} entries, the batch has {n}")
    time = time.expand(n).contiguous()
    # out of place (amx_step_to): fresh outputs, no copies of the inputs in front of the launch
    qpos, qvel, warm, tout = (torch.empty_like(t["qpos"]), torch.empty_like(t["qvel"]), torch.empty_like(t["qacc_warmstart"]),
                              torch.empty_like(time))
    with torch.cuda.device(m.device):
        _lib.check(m._h.L.amx_step_to(m._h.ptr, n, int(nsub), t["qpos"].data_ptr(), t["qvel"].data_ptr(), ctrl.data_ptr(),
                                          t["qacc_warmstart"].data_ptr(), time.data_ptr(), qpos.data_ptr(), qvel.data_ptr(), warm.data_ptr(),
                                          tout.data_ptr(), _stream_ptr(m.device)), m._h.L)
    time = tout
    s = (lambda t: t) if b else (lambda t: t[0])
    return d.replace(qpos=s(qpos), qvel=s(qvel), qacc_warmstart=s(warm), qacc=s(warm) if m.nefc_max > 0 else d.qacc, time=s(time),
                     ctrl=s(ctrl))


def make_env_tail(obs_terms, reward_terms, episode_length: int) -> "_lib.AmxEnvTailDesc":
    """The descriptor of a fused env.step (include/amx.h: amx_env_tail_desc).  obs_terms: [(kind, index)] with kind in
    {"qpos", "cos", "sin", "qvel"}; reward_terms: [(kind, index, target, weight)] with kind in {"wrapped_angle", "qpos", "qvel",
    "ctrl"}: reward = -sum weight * term^2."""
    ok = {"qpos": 0, "cos": 1, "sin": 2, "qvel": 3}
    rk = {"wrapped_angle": 0, "qpos": 1, "qvel": 2, "ctrl": 3}
    if len(obs_terms) > _lib.TAIL_MAX_OBS or len(reward_terms) > _lib.TAIL_MAX_REW:
        raise ValueError(f"a fused env tail holds at most {_lib.TAIL_MAX_OBS} observation and {_lib.TAIL_MAX_REW} reward terms")
    d = _lib.AmxEnvTailDesc()
    d.nobs, d.nrew, d.episode_length = len(obs_terms), len(reward_terms), int(episode_length)
    for k, (kind, i) in enumerate(obs_terms):
        d.obs_kind[k], d.obs_index[k] = ok[kind], int(i)
    for k, (kind, i, target, weight) in enumerate(reward_terms):
        d.rew_kind[k], d.rew_index[k], d.rew_target[k], d.rew_weight[k] = rk[kind], int(i), float(target), float(weight)
    return d


def env_step(m: Model, d: Data, action: torch.Tensor, nsub: int, tail, first: Data, first_obs: torch.Tensor, counters: dict):
    """One wrapped env.step as ONE call of the C ABI (amx_env_step: the physics launch + one elementwise launch): the counterpart of
    the reference's jitted env.step under brax's Episode / AutoReset wrappers.  `d`: batched state; `first` / `first_obs`: the
    state and observation every env was first reset to; `counters`: {"steps", "episode_return", "last_episode_return", "episodes"}
    [n] tensors and optionally "step" (the env's own 0-dim counter).  Returns (Data, out) with out = {obs, terminal_obs, reward,
    done, truncation, steps, episode_return, last_episode_return, episodes[, step]} - all fresh tensors, inputs untouched."""
    b, n, t = _prep(m, d, ("qpos", "qvel", "qacc_warmstart"))
    if not b:
        raise ValueError("env_step: the state must carry the batch axis")
    dev = m.device
    def f32(x, shape):  # the common case - already fp32, on the device, contiguous, full shape - costs three comparisons
        if x.dtype is torch.float32 and x.device == dev and tuple(x.shape) == shape and x.is_contiguous():
            return x
        return _c(x.to(device=dev, dtype=torch.float32).expand(shape))
    act = f32(action, (n, m.nu))
    time = f32(d.time if torch.is_tensor(d.time) else torch.tensor(float(d.time or 0.0)), (n,))
    fq, fv, fw = f32(first.qpos, (n, m.nq)), f32(first.qvel, (n, m.nv)), f32(first.qacc_warmstart, (n, m.nv))
    fc, ft = f32(first.ctrl, (n, m.nu)), f32(first.time if torch.is_tensor(first.time) else torch.tensor(0.0), (n,))
    nobs = tail.nobs
    fo = f32(first_obs, (n, nobs))
    cin = {k: f32(counters[k], (n,)) for k in ("steps", "episode_return", "last_episode_return", "episodes")}
    # one allocation for every output, carved into contiguous views: state fields [n, w], per-env scalars [n]
    fields2d = (("qpos_out", m.nq), ("qvel_out", m.nv), ("ctrl_out", m.nu), ("warm_out", m.nv), ("qacc_out", m.nv), ("obs_out", nobs), ("terminal_obs_out", nobs))
    scalars = ("time_out", "reward_out", "done_out", "truncation_out", "steps_out", "episode_return_out", "last_episode_return_out", "episodes_out")
    buf = torch.empty(((sum(w for _, w in fields2d) + len(scalars)) * n + 1,), dtype=torch.float32, device=dev)
    o, off = {}, 0
    for name, w in fields2d:
        o[name] = buf[off:off + n * w].view(n, w)
        off += n * w
    for name in scalars:
        o[name] = buf[off:off + n]
        off += n
    step_in = counters.get("step")
    step_out = buf[off:off + 1].view(()) if step_in is not None else None
    io = _lib.AmxEnvTailIO()
    vals = dict(first_qpos=fq, first_qvel=fv, first_ctrl=fc, first_warm=fw, first_time=ft, first_obs=fo, step_counter_in=step_in, step_counter_out=step_out, **cin, **o)
    for name in _lib.TAIL_IO_FIELDS:
        v = vals.get(name)
        setattr(io, name, None if v is None else v.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(m._h.L.amx_env_step(m._h.ptr, n, int(nsub), t["qpos"].data_ptr(), t["qvel"].data_ptr(), act.data_ptr(), t["qacc_warmstart"].data_ptr(),
                                           time.data_ptr(), C.byref(tail), C.byref(io), _stream_ptr(dev)), m._h.L)
    data = Data(qpos=o["qpos_out"], qvel=o["qvel_out"], ctrl=o["ctrl_out"], qacc_warmstart=o["warm_out"], time=o["time_out"],
                qacc=o["qacc_out"] if m.nefc_max > 0 else d.qacc, qacc_smooth=d.qacc_smooth, qfrc_bias=d.qfrc_bias, qfrc_constraint=d.qfrc_constraint,
                xpos=d.xpos, xquat=d.xquat)  # (the derived leaves are stale pass-throughs, as after mjx.step)
    out = {"obs": o["obs_out"], "terminal_obs": o["terminal_obs_out"], "reward": o["reward_out"], "done": o["done_out"], "truncation": o["truncation_out"],
           "steps": o["steps_out"], "episode_return": o["episode_return_out"], "last_episode_return": o["last_episode_return_out"], "episodes": o["episodes_out"]}
    if step_out is not None:
        out["step"] = step_out
    return data, out


def rollout(m: Model, x0: torch.Tensor, us: torch.Tensor, return_traj: bool = True):
    """Fused shoot (reference ambersim/trajopt/shooting.py:22-48) for a batch: x0 [nx] or [S,nx], us [S,N,nu]
    -> xs [S,N+1,nx] (or the final states [S,nx] when return_traj is False)."""
    dev = m.device
    nx = m.nq + m.nv
    if us.dim() != 3 or x0.dim() not in (1, 2):
        raise ValueError(f"rollout: x0 must be [nx] or [S,nx] and us [S,N,nu], got {tuple(x0.shape)} and {tuple(us.shape)}")
    us = _c(us.to(device=dev, dtype=torch.float32))
    x0 = _c(x0.to(device=dev, dtype=torch.float32))
    S, N = us.shape[0], us.shape[1]
    if x0.dim() not in (1, 2) or us.shape[2] != m.nu or x0.shape[-1] != nx or (x0.dim() == 2 and x0.shape[0] != S):
        raise ValueError(f"rollout: shapes x0 {tuple(x0.shape)}, us {tuple(us.shape)} do not match nx={nx}, nu={m.nu}")
    xs = torch.empty((S, N + 1, nx), dtype=torch.float32, device=dev) if return_traj else None
    xf = None if return_traj else torch.empty((S, nx), dtype=torch.float32, device=dev)
    if S == 0:  # an empty batch is a valid input: nothing to launch
        return xs if return_traj else xf
    with torch.cuda.device(dev):
        _lib.check(m._h.L.amx_rollout(m._h.ptr, S, N, x0.data_ptr(), int(x0.dim() == 2), us.data_ptr(),
                                          xs.data_ptr() if xs is not None else None, xf.data_ptr() if xf is not None else None,
                                          _stream_ptr(dev)))
    return xs if return_traj else xf
