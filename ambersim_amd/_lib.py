"""ctypes binding of libamx.so (C ABI: include/amx.h).  No CPU fallback: if the HIP library is
missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

from ._desc import AmxModelDesc

_CSRC = Path(__file__).resolve().parent / "csrc"
_LIB_PATH = Path(os.environ.get("AMX_LIBRARY", _CSRC / "libamx.so"))  # AMX_LIBRARY: diagnostic builds only
_lib = None

AMX_OK, AMX_EINVAL, AMX_EUNSUPPORTED, AMX_ENOMEM, AMX_EHIP, AMX_ECAPACITY = 0, -1, -2, -3, -4, -5


class AmxForwardOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in [
        "qacc_dev", "qacc_smooth_dev", "qfrc_bias_dev", "qfrc_constraint_dev", "qM_dev", "xpos_dev", "xquat_dev",
        "contact_dist_dev", "nefc_dev"]]


TAIL_MAX_OBS, TAIL_MAX_REW = 16, 8


class AmxEnvTailDesc(C.Structure):
    """amx_env_tail_desc (include/amx.h): observation / reward terms and the episode length of a fused env.step."""
    _fields_ = [("nobs", C.c_int), ("obs_kind", C.c_int * TAIL_MAX_OBS), ("obs_index", C.c_int * TAIL_MAX_OBS),
                ("nrew", C.c_int), ("rew_kind", C.c_int * TAIL_MAX_REW), ("rew_index", C.c_int * TAIL_MAX_REW),
                ("rew_target", C.c_float * TAIL_MAX_REW), ("rew_weight", C.c_float * TAIL_MAX_REW), ("episode_length", C.c_int)]


TAIL_IO_FIELDS = ("first_qpos", "first_qvel", "first_ctrl", "first_warm", "first_time", "first_obs",
                  "steps", "episode_return", "last_episode_return", "episodes", "step_counter_in",
                  "qpos_out", "qvel_out", "ctrl_out", "warm_out", "time_out", "qacc_out",
                  "obs_out", "terminal_obs_out", "reward_out", "done_out", "truncation_out",
                  "steps_out", "episode_return_out", "last_episode_return_out", "episodes_out", "step_counter_out")


class AmxEnvTailIO(C.Structure):
    """amx_env_tail_io (include/amx.h): device pointers of a fused env.step, in the header's order."""
    _fields_ = [(n, C.c_void_p) for n in TAIL_IO_FIELDS]


EXPORTS = ["amx_env_step", "amx_version", "amx_last_error", "amx_model_create", "amx_model_destroy", "amx_model_query",
           "amx_forward", "amx_step", "amx_step_to", "amx_rollout", "amx_quadcost", "amx_fd_workspace_bytes", "amx_fd_grad"]


def build(force: bool = False) -> Path:
    """Compiles csrc/ into libamx.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-s", "-C", str(_CSRC), "clean"], check=True)
    subprocess.run(["make", "-s", "-C", str(_CSRC), "libamx.so"], check=True)
    return _LIB_PATH


def _load_hip_runtime():
    """libamx.so is linked with -no-hip-rt: it binds to the HIP runtime of the host process.  Under PyTorch that is
    the libamdhip64 inside the torch wheel (two runtimes in one process do not share devices), else the system one."""
    cands = []
    try:
        import torch
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    for c in cands:
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    raise RuntimeError("no HIP runtime (libamdhip64) found for libamx.so")


def load(path) -> C.CDLL:
    """Loads one build of the library (the default libamx.so, or a per-model build of ambersim_amd._spec) and declares its ABI."""
    _load_hip_runtime()
    L = C.CDLL(str(path))
    L.amx_version.restype = C.c_int
    L.amx_last_error.restype = C.c_char_p
    L.amx_model_create.argtypes = [C.POINTER(AmxModelDesc), C.c_int, C.POINTER(C.c_void_p)]
    L.amx_model_destroy.argtypes = [C.c_void_p]
    L.amx_model_query.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
    L.amx_forward.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.POINTER(AmxForwardOut), C.c_void_p]
    L.amx_step.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p]
    L.amx_step_to.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_void_p]
    L.amx_env_step.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.POINTER(AmxEnvTailDesc), C.POINTER(AmxEnvTailIO), C.c_void_p]
    L.amx_rollout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.amx_quadcost.argtypes = [C.c_int] * 4 + [C.c_void_p] * 7 + [C.c_void_p]
    L.amx_fd_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    L.amx_fd_grad.argtypes = ([C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5 +
                              [C.c_void_p, C.c_size_t, C.c_void_p])
    for name in EXPORTS:
        getattr(L, name)
    return L


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "ambersim_amd has no CPU fallback for the physics path.")
        _lib = load(_LIB_PATH)
    return _lib


_EXC = {AMX_EINVAL: ValueError, AMX_EUNSUPPORTED: NotImplementedError, AMX_ENOMEM: MemoryError,
        AMX_EHIP: RuntimeError, AMX_ECAPACITY: NotImplementedError}


def check(code: int, L: C.CDLL = None):
    """Maps a C status to the exception type the reference's callers expect
    (NotImplementedError for unsupported features: reference ambersim/utils/io_utils.py:228-241)."""
    if code != AMX_OK:
        msg = (L or lib()).amx_last_error().decode(errors="replace")
        raise _EXC.get(code, RuntimeError)(f"amx error {code}: {msg}")
