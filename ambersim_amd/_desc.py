"""ctypes mirror of `amx_model_desc` (include/amx.h) and its builder from a host MjModel."""
from __future__ import annotations

import ctypes as C

import numpy as np

_I = C.POINTER(C.c_int)
_D = C.POINTER(C.c_double)

_INT_FIELDS_1 = ["body_parentid", "body_rootid", "body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum"]
_DBL_FIELDS_1 = ["body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia", "body_invweight0"]
_INT_FIELDS_2 = ["jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited"]
_DBL_FIELDS_2 = ["jnt_pos", "jnt_axis", "jnt_range", "jnt_margin", "jnt_stiffness", "jnt_solref", "jnt_solimp", "qpos0", "qpos_spring"]
_INT_FIELDS_3 = ["dof_bodyid", "dof_jntid", "dof_parentid"]
_DBL_FIELDS_3 = ["dof_armature", "dof_damping", "dof_invweight0"]
_INT_FIELDS_4 = ["geom_type", "geom_bodyid"]
_DBL_FIELDS_4 = ["geom_pos", "geom_quat", "geom_size"]
_INT_FIELDS_5 = ["pair_geom1", "pair_geom2", "pair_fn", "pair_vertadr", "pair_vertnum"]
_DBL_FIELDS_5 = ["pair_friction", "pair_solref", "pair_solimp", "pair_margin", "convex_vert"]
_INT_FIELDS_5B = ["pair_hull1", "pair_hull2", "hull_vertadr", "hull_vertnum", "hull_faceadr", "hull_facenum", "hull_edgeadr", "hull_edgenum"]
_DBL_FIELDS_5B = ["hull_vert", "hull_face_normal"]
_INT_FIELDS_5C = ["hull_face_vertadr", "hull_face_vertnum", "hull_facevert", "hull_edge"]
_INT_FIELDS_6 = ["actuator_trnid", "actuator_biastype", "actuator_ctrllimited", "actuator_forcelimited"]
_DBL_FIELDS_6 = ["actuator_gear", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange", "actuator_forcerange"]
_INT_FIELDS_7 = ["eq_obj1id", "eq_obj2id", "eq_active"]
_DBL_FIELDS_7 = ["eq_data", "eq_solref", "eq_solimp"]
_INT_FIELDS_8 = ["eq_type"]  # version 3, appended (eq_data is [neq, 11] since then)
_DBL_FIELDS_8 = ["dof_frictionloss", "dof_solref", "dof_solimp"]


def _ptr_fields():
    out = []
    for ints, dbls in [(_INT_FIELDS_1, _DBL_FIELDS_1), (_INT_FIELDS_2, _DBL_FIELDS_2), (_INT_FIELDS_3, _DBL_FIELDS_3),
                       (_INT_FIELDS_4, _DBL_FIELDS_4), (_INT_FIELDS_5, _DBL_FIELDS_5), (_INT_FIELDS_5B, _DBL_FIELDS_5B), (_INT_FIELDS_5C, []),
                       (_INT_FIELDS_6, _DBL_FIELDS_6),
                       (_INT_FIELDS_7, _DBL_FIELDS_7), (_INT_FIELDS_8, _DBL_FIELDS_8)]:
        out += [(n, _I) for n in ints] + [(n, _D) for n in dbls]
    return out


class AmxModelDesc(C.Structure):
    _fields_ = (
        [("version", C.c_int)]
        + [(n, C.c_int) for n in ["nq", "nv", "nu", "nbody", "njnt", "ngeom", "neq", "npair", "nconvexvert",
                                  "nhull", "nhullvert", "nhullface", "nhullfacevert", "nhulledge"]]
        + [("timestep", C.c_double), ("gravity", C.c_double * 3), ("tolerance", C.c_double),
           ("ls_tolerance", C.c_double), ("impratio", C.c_double), ("meaninertia", C.c_double)]
        + [(n, C.c_int) for n in ["iterations", "ls_iterations", "solver", "integrator", "disableflags"]]
        + _ptr_fields()
    )


AMX_VERSION = 3


def build_desc(m) -> AmxModelDesc:
    """Packs a host MjModel (ambersim_amd.mjcf.MjModel) into an `amx_model_desc`.

    The returned struct keeps the numpy buffers alive in `_keep`."""
    d = AmxModelDesc()
    d.version = AMX_VERSION
    d.nq, d.nv, d.nu, d.nbody, d.njnt, d.ngeom, d.neq, d.npair = m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.neq, m.npair
    d.nconvexvert = int(m.convex_vert.shape[0])
    d.nhull, d.nhullvert, d.nhullface = int(len(m.hull_vertadr)), int(m.hull_vert.shape[0]), int(len(m.hull_face_vertadr))
    d.nhullfacevert, d.nhulledge = int(len(m.hull_facevert)), int(np.asarray(m.hull_edge).reshape(-1, 2).shape[0])
    o = m.opt
    d.timestep, d.tolerance, d.ls_tolerance, d.impratio = float(o.timestep), float(o.tolerance), float(o.ls_tolerance), float(o.impratio)
    d.gravity = (C.c_double * 3)(*[float(x) for x in o.gravity])
    d.meaninertia = float(m.stat.meaninertia)
    d.iterations, d.ls_iterations, d.solver = int(o.iterations), int(o.ls_iterations), int(o.solver)
    d.integrator, d.disableflags = int(o.integrator), int(o.disableflags)
    keep = []
    for name, typ in _ptr_fields():
        arr = getattr(m, name)
        if typ is _I:
            a = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
        else:
            a = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
        if a.size == 0:
            a = np.zeros(1, dtype=a.dtype)
        keep.append(a)
        setattr(d, name, a.ctypes.data_as(typ))
    d._keep = keep
    return d
