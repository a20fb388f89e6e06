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
    key = np.dtype(dtype).name
    if key not in _LIBS:
        name = {"float64": "libamxo_f64.so", "float32": "libamxo_f32.so"}[key]
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
        _LIBS[key] = lib
    return _LIBS[key]


class Oracle:
    """One model + one scratch data block in the chosen precision."""

    def __init__(self, mj_model, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.lib = _lib(dtype)
        self.mj = mj_model
        if getattr(mj_model, "nunsupported_pairs", 0):
            raise NotImplementedError("model has collision pairs outside the implemented pair functions")
        self._desc = build_desc(mj_model)
        self.m = self.lib.amxo_model_create(C.byref(self._desc))
        self.d = self.lib.amxo_data_create(self.m)
        self.nq, self.nv, self.nu = mj_model.nq, mj_model.nv, mj_model.nu
        self.ncon = self.lib.amxo_model_int(self.m, b"ncon")
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

    def max_threads(self):
        return self.lib.amxo_max_threads()
