"""MJCF subset loader + model compiler (host side, numpy float64).

The reference obtains its host model from the MuJoCo C compiler
(`mujoco.MjModel.from_xml_path`, reference ambersim/utils/io_utils.py:203-206).  MuJoCo is
not installable here, so this module compiles the MJCF subset the hot path needs into an
`MjModel` with MuJoCo's field names.  Everything here is setup-time host code; the per-step
physics lives in csrc/ (HIP).

Supported: <include>, <compiler angle meshdir autolimits eulerseq>, <option> (+<flag>),
<default> classes, <asset><mesh file|vertex scale>, body tree with <inertial>, <joint>,
<freejoint>, <geom> (plane sphere capsule box cylinder ellipsoid mesh), <actuator>
motor/position/velocity/general on joints, <equality><joint polycoef>, <keyframe>,
<contact><exclude>.  Anything else that would change the dynamics raises NotImplementedError
(the same exception type `mjx.device_put` raises for unsupported features, reference
ambersim/utils/io_utils.py:228-241).
"""
from __future__ import annotations

import copy
import os
import struct
import xml.etree.ElementTree as ET
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

# --- MuJoCo enums (values follow mujoco's mjtJoint / mjtGeom / mjtSolver / mjtDisableBit) ---
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
GEOM_NAMES = {"plane": 0, "hfield": 1, "sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6, "mesh": 7}
SOL_PGS, SOL_CG, SOL_NEWTON = 0, 1, 2
INT_EULER, INT_RK4, INT_IMPLICIT, INT_IMPLICITFAST = 0, 1, 2, 3
EQ_CONNECT, EQ_WELD, EQ_JOINT, EQ_TENDON = 0, 1, 2, 3
GAIN_FIXED, BIAS_NONE, BIAS_AFFINE = 0, 0, 1

MINVAL = 1e-15


class DisableBit:
    """mjtDisableBit values (mujoco.mjx._src.types.DisableBit, used by the reference at
    tests/trajopt/test_predictive_sampler.py:7,29)."""

    CONSTRAINT = 1 << 0
    EQUALITY = 1 << 1
    FRICTIONLOSS = 1 << 2
    LIMIT = 1 << 3
    CONTACT = 1 << 4
    PASSIVE = 1 << 5
    GRAVITY = 1 << 6
    CLAMPCTRL = 1 << 7
    WARMSTART = 1 << 8
    FILTERPARENT = 1 << 9
    ACTUATION = 1 << 10
    REFSAFE = 1 << 11
    SENSOR = 1 << 12
    MIDPHASE = 1 << 13
    EULERDAMP = 1 << 14


_FLAG_BITS = {
    "constraint": DisableBit.CONSTRAINT, "equality": DisableBit.EQUALITY,
    "frictionloss": DisableBit.FRICTIONLOSS, "limit": DisableBit.LIMIT,
    "contact": DisableBit.CONTACT, "passive": DisableBit.PASSIVE,
    "gravity": DisableBit.GRAVITY, "clampctrl": DisableBit.CLAMPCTRL,
    "warmstart": DisableBit.WARMSTART, "filterparent": DisableBit.FILTERPARENT,
    "actuation": DisableBit.ACTUATION, "refsafe": DisableBit.REFSAFE,
    "sensor": DisableBit.SENSOR, "midphase": DisableBit.MIDPHASE,
    "eulerdamp": DisableBit.EULERDAMP,
}


# ----------------------------------------------------------------------------- math helpers
def _quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def _quat_conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def _quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def _mat_to_quat(m):
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0."""
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    q = q / np.linalg.norm(q)
    return q if q[0] >= 0 else -q


def _rot(q, v):
    return _quat_to_mat(q) @ np.asarray(v, dtype=float)


def _axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=float)
    n = np.linalg.norm(axis)
    if n < MINVAL:
        return np.array([1.0, 0, 0, 0])
    axis = axis / n
    s = np.sin(angle * 0.5)
    return np.array([np.cos(angle * 0.5), axis[0] * s, axis[1] * s, axis[2] * s])


def _zaxis_quat(z):
    """Minimal rotation taking (0,0,1) to z (MuJoCo's mjuu_z2quat)."""
    z = np.asarray(z, dtype=float)
    z = z / np.linalg.norm(z)
    axis = np.cross([0, 0, 1.0], z)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if z[2] > 0 else np.array([0.0, 1.0, 0, 0])
    ang = np.arctan2(s, z[2])
    return _axis_angle_quat(axis / s, ang)


def _eig3_frame(I):
    """Principal moments + quaternion of the principal frame of a symmetric 3x3 inertia."""
    w, v = np.linalg.eigh(I)
    order = np.argsort(-w)  # MuJoCo sorts descending
    w, v = w[order], v[:, order]
    if np.linalg.det(v) < 0:
        v[:, 2] = -v[:, 2]
    return w, _mat_to_quat(v)


# ----------------------------------------------------------------------------- containers
class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def replace(self, **kw):
        out = copy.copy(self)
        for k, v in kw.items():
            if not hasattr(out, k):
                raise AttributeError(k)
            setattr(out, k, v)
        return out

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in self.__dict__.items())})"


class MjOption(_Namespace):
    """Counterpart of mujoco.MjOption (fields the hot path reads)."""


class MjStatistic(_Namespace):
    pass


class MjModel(_Namespace):
    """Host model with MuJoCo's field names (numpy float64 / int32 arrays).

    `from_xml_path` mirrors `mujoco.MjModel.from_xml_path` (reference
    ambersim/utils/io_utils.py:206)."""

    @staticmethod
    def from_xml_path(path) -> "MjModel":
        return compile_mjcf(str(path))

    @staticmethod
    def from_xml_string(xml: str, model_dir: str = ".") -> "MjModel":
        root = ET.fromstring(xml)
        _resolve_includes(root, Path(model_dir))
        return _Compiler(root, Path(model_dir)).compile()

    def body_name2id(self, name):
        return self.names_body.index(name)

    def joint_name2id(self, name):
        return self.names_joint.index(name)


class MjData(_Namespace):
    """Host data (reference ambersim/rl/base.py:51 `mujoco.MjData(mj_model)`)."""

    def __init__(self, model: Optional[MjModel] = None, **kw):
        if model is not None:
            kw = dict(
                qpos=model.qpos0.copy(), qvel=np.zeros(model.nv), ctrl=np.zeros(model.nu),
                qacc=np.zeros(model.nv), qacc_warmstart=np.zeros(model.nv), time=0.0,
            )
        super().__init__(**kw)
        self._model = model  # mujoco.MjData knows its model; mjx.device_put(MjData) needs its sizes


# ----------------------------------------------------------------------------- parsing
def _floats(s, n=None):
    v = np.array([float(x) for x in s.replace(",", " ").split()], dtype=float)
    if n is not None and v.size != n:
        raise ValueError(f"expected {n} numbers, got {v.size}: {s!r}")
    return v


def _resolve_includes(root: ET.Element, base: Path):
    """Splices <include file=...> children in place (MuJoCo include semantics)."""
    changed = True
    while changed:
        changed = False
        for parent in list(root.iter()):
            for i, child in enumerate(list(parent)):
                if child.tag == "include":
                    inc = ET.parse(base / child.get("file")).getroot()
                    idx = list(parent).index(child)
                    parent.remove(child)
                    for k, sub in enumerate(list(inc)):
                        parent.insert(idx + k, sub)
                    changed = True


def _merge_sections(root: ET.Element, tag: str) -> List[ET.Element]:
    return [e for e in root if e.tag == tag]


_ACTUATOR_TAGS = ("motor", "position", "velocity", "general")
_DEFAULT_TAGS = ("joint", "geom", "mesh", "equality", "general") + _ACTUATOR_TAGS[:3]


class _Defaults:
    """Default-class tree: class name -> {element tag -> attribute dict}."""

    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}

    def load(self, elem: ET.Element, parent: Optional[str] = None):
        name = elem.get("class", "main" if parent is None else None)
        if name is None:
            raise ValueError("nested <default> needs a class name")
        base = copy.deepcopy(self.classes.get(parent, {})) if parent else copy.deepcopy(self.classes.get("main", {}))
        if name == "main" and parent is None:
            base = copy.deepcopy(self.classes["main"])
        for child in elem:
            if child.tag == "default":
                continue
            base.setdefault(child.tag, {}).update(child.attrib)
        self.classes[name] = base
        for child in elem:
            if child.tag == "default":
                self.load(child, name)

    def attrs(self, tag: str, cls: Optional[str], own: Dict[str, str]) -> Dict[str, str]:
        out = dict(self.classes.get(cls or "main", self.classes["main"]).get(tag, {}))
        out.update(own)
        out.pop("class", None)
        return out


def _load_obj_vertices(path: Path) -> np.ndarray:
    verts = []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append([float(p[1]), float(p[2]), float(p[3])])
    return np.array(verts, dtype=float)


def _load_stl_vertices(path: Path) -> np.ndarray:
    with open(path, "rb") as f:
        data = f.read()
    ntri = struct.unpack_from("<I", data, 80)[0]
    if 84 + 50 * ntri != len(data):
        raise NotImplementedError("only binary STL is supported")
    tri = np.frombuffer(data, dtype=np.dtype([("n", "<3f4"), ("v", "<9f4"), ("a", "<u2")]), count=ntri, offset=84)
    return np.unique(tri["v"].reshape(-1, 3).astype(float), axis=0)


def _convex_hull(verts: np.ndarray):
    """Hull vertices and outward triangles (scipy Qhull; MuJoCo uses its bundled qhull)."""
    from scipy.spatial import ConvexHull

    hull = ConvexHull(verts)
    used = np.unique(hull.simplices)
    remap = -np.ones(len(verts), dtype=int)
    remap[used] = np.arange(len(used))
    hv = verts[used]
    faces = remap[hull.simplices]
    c = hv.mean(axis=0)
    for k, f in enumerate(faces):  # orient outward
        n = np.cross(hv[f[1]] - hv[f[0]], hv[f[2]] - hv[f[0]])
        if np.dot(n, hv[f[0]] - c) < 0:
            faces[k] = f[::-1]
    return hv, faces


def _farthest_subset(hv: np.ndarray, n: int) -> np.ndarray:
    """At most n of the hull vertices, chosen greedily: the two extremes of the longest axis, then repeatedly the vertex
    farthest from those already chosen (deterministic; an inner approximation of the hull, like qhull's early termination
    that MuJoCo's `maxhullvert` uses)."""
    if len(hv) <= n:
        return hv
    ax = int(np.argmax(hv.max(0) - hv.min(0)))
    chosen = [int(np.argmin(hv[:, ax])), int(np.argmax(hv[:, ax]))]
    d = np.minimum(np.linalg.norm(hv - hv[chosen[0]], axis=1), np.linalg.norm(hv - hv[chosen[1]], axis=1))
    while len(chosen) < n:
        k = int(np.argmax(d))
        chosen.append(k)
        d = np.minimum(d, np.linalg.norm(hv - hv[k], axis=1))
    return hv[sorted(chosen)]


def _polyhedron_mass_props(v: np.ndarray, faces: np.ndarray):
    """Volume, centre of mass and unit-density inertia about the COM of a closed triangle mesh."""
    vol, com = 0.0, np.zeros(3)
    P = np.zeros((3, 3))  # second moment integral x x^T dV about origin
    canon = np.array([[2, 1, 1], [1, 2, 1], [1, 1, 2]]) / 120.0
    for f in faces:
        a, b, c = v[f[0]], v[f[1]], v[f[2]]
        A = np.stack([a, b, c], axis=1)
        d = np.linalg.det(A)
        vol += d / 6.0
        com += d / 24.0 * (a + b + c)
        P += d * (A @ canon @ A.T)
    com = com / vol
    P = P - vol * np.outer(com, com)
    I = np.trace(P) * np.eye(3) - P
    return vol, com, I


# ----------------------------------------------------------------------------- compiler
class _Compiler:
    def __init__(self, root: ET.Element, model_dir: Path):
        self.root = root
        self.dir = model_dir
        self.defaults = _Defaults()
        comp = {}
        for c in _merge_sections(root, "compiler"):
            comp.update(c.attrib)
        self.degree = comp.get("angle", "degree") == "degree"
        self.meshdir = comp.get("meshdir", comp.get("assetdir", ""))
        self.autolimits = comp.get("autolimits", "true") == "true"
        self.eulerseq = comp.get("eulerseq", "xyz")
        self.boundmass = float(comp.get("boundmass", 0))
        self.boundinertia = float(comp.get("boundinertia", 0))
        self.inertiafromgeom = comp.get("inertiafromgeom", "auto")
        self.balanceinertia = comp.get("balanceinertia", "false") == "true"
        if comp.get("coordinate", "local") != "local":
            raise NotImplementedError("compiler coordinate='global'")
        for d in _merge_sections(root, "default"):
            self.defaults.load(d)

    # -- orientation / frames
    def _angle(self, a):
        return np.deg2rad(a) if self.degree else a

    def _orientation(self, at: Dict[str, str]) -> np.ndarray:
        if "quat" in at:
            q = _floats(at["quat"], 4)
            return q / np.linalg.norm(q)
        if "axisangle" in at:
            v = _floats(at["axisangle"], 4)
            return _axis_angle_quat(v[:3], self._angle(v[3]))
        if "euler" in at:
            e = self._angle(_floats(at["euler"], 3))
            q = np.array([1.0, 0, 0, 0])
            for ch, ang in zip(self.eulerseq, e):
                ax = {"x": [1, 0, 0], "y": [0, 1, 0], "z": [0, 0, 1]}[ch.lower()]
                r = _axis_angle_quat(ax, ang)
                q = _quat_mul(q, r) if ch.islower() else _quat_mul(r, q)
            return q
        if "xyaxes" in at:
            v = _floats(at["xyaxes"], 6)
            x = v[:3] / np.linalg.norm(v[:3])
            y = v[3:] - x * np.dot(x, v[3:])
            y = y / np.linalg.norm(y)
            return _mat_to_quat(np.stack([x, y, np.cross(x, y)], axis=1))
        if "zaxis" in at:
            return _zaxis_quat(_floats(at["zaxis"], 3))
        return np.array([1.0, 0, 0, 0])

    # -- assets
    def _load_meshes(self):
        self.meshes = {}
        for asset in _merge_sections(self.root, "asset"):
            for m in asset:
                if m.tag != "mesh":
                    continue
                at = self.defaults.attrs("mesh", m.get("class"), m.attrib)
                if "vertex" in at:
                    verts = _floats(at["vertex"]).reshape(-1, 3)
                    name = at.get("name")
                else:
                    fpath = self.dir / self.meshdir / at["file"]
                    name = at.get("name", Path(at["file"]).stem)
                    ext = fpath.suffix.lower()
                    if ext == ".obj":
                        verts = _load_obj_vertices(fpath)
                    elif ext == ".stl":
                        verts = _load_stl_vertices(fpath)
                    else:
                        raise NotImplementedError(f"mesh format {ext}")
                scale = _floats(at.get("scale", "1 1 1"), 3)
                verts = verts * scale
                hv, faces = _convex_hull(verts)
                maxhull = int(at.get("maxhullvert", "-1"))  # MuJoCo's mesh attribute: cap on the hull's vertex count
                if maxhull != -1:
                    if maxhull < 4:
                        raise ValueError("maxhullvert must be -1 or at least 4")
                    hv, faces = _convex_hull(_farthest_subset(hv, maxhull))
                self.meshes[name] = dict(vert=hv, face=faces)

    # -- geoms
    def _geom(self, e: ET.Element, childclass: Optional[str]):
        at = self.defaults.attrs("geom", e.get("class", childclass), e.attrib)
        gtype = GEOM_NAMES[at.get("type", "sphere")]
        if "mesh" in at and "type" not in at:
            gtype = GEOM_MESH
        if gtype in (GEOM_HFIELD,):
            raise NotImplementedError("hfield geoms")
        size = np.zeros(3)
        sz = _floats(at.get("size", "0 0 0"))
        size[: min(3, sz.size)] = sz[:3]
        pos = _floats(at.get("pos", "0 0 0"), 3)
        quat = self._orientation(at)
        if "fromto" in at:
            ft = _floats(at["fromto"], 6)
            a, b = ft[:3], ft[3:]
            pos = 0.5 * (a + b)
            quat = _zaxis_quat(b - a)
            half = 0.5 * np.linalg.norm(b - a)
            if gtype in (GEOM_CAPSULE, GEOM_CYLINDER):
                size[1] = half
            elif gtype in (GEOM_BOX, GEOM_ELLIPSOID):
                size[2] = half
            else:
                raise ValueError("fromto on unsupported geom type")
        mesh = None
        if gtype == GEOM_MESH:
            mesh = at["mesh"]
            if mesh not in self.meshes:
                raise ValueError(f"unknown mesh {mesh}")
        fr = _floats(at.get("friction", "1 0.005 0.0001"))
        friction = np.array([1.0, 0.005, 0.0001])
        friction[: fr.size] = fr
        g = dict(
            name=at.get("name", ""), type=gtype, size=size, pos=pos, quat=quat, mesh=mesh,
            contype=int(at.get("contype", 1)), conaffinity=int(at.get("conaffinity", 1)),
            condim=int(at.get("condim", 3)), priority=int(at.get("priority", 0)),
            friction=friction, solmix=float(at.get("solmix", 1)),
            solref=_floats(at.get("solref", "0.02 1"), 2),
            solimp=self._solimp(at.get("solimp")),
            margin=float(at.get("margin", 0)), gap=float(at.get("gap", 0)),
            density=float(at.get("density", 1000)), mass=float(at["mass"]) if "mass" in at else None,
            group=int(at.get("group", 0)),
        )
        return g

    @staticmethod
    def _solimp(s):
        v = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
        if s is not None:
            u = _floats(s)
            v[: u.size] = u
        return v

    def _geom_mass_props(self, g):
        """(mass, com in geom frame, inertia matrix about com in geom frame)."""
        t, s = g["type"], g["size"]
        com = np.zeros(3)
        if t == GEOM_SPHERE:
            vol = 4.0 / 3.0 * np.pi * s[0] ** 3
            I = np.eye(3) * 0.4 * s[0] ** 2
        elif t == GEOM_CAPSULE:
            r, h = s[0], s[1]
            vcyl, vsph = np.pi * r * r * 2 * h, 4.0 / 3.0 * np.pi * r ** 3
            vol = vcyl + vsph
            # unit-density moments, then normalise by volume
            ixx = vcyl * (r * r / 4 + h * h / 3) + vsph * (2 * r * r / 5 + h * h + 3 * h * r / 4)
            izz = vcyl * r * r / 2 + vsph * 2 * r * r / 5
            I = np.diag([ixx, ixx, izz]) / vol
        elif t == GEOM_CYLINDER:
            r, h = s[0], s[1]
            vol = np.pi * r * r * 2 * h
            I = np.diag([(3 * r * r + 4 * h * h) / 12, (3 * r * r + 4 * h * h) / 12, r * r / 2])
        elif t == GEOM_BOX:
            vol = 8 * s[0] * s[1] * s[2]
            I = np.diag([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2]) / 3.0
        elif t == GEOM_ELLIPSOID:
            vol = 4.0 / 3.0 * np.pi * s[0] * s[1] * s[2]
            I = np.diag([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2]) / 5.0
        elif t == GEOM_MESH:
            m = self.meshes[g["mesh"]]
            vol, com, Iv = _polyhedron_mass_props(m["vert"], m["face"])
            I = Iv / vol
        else:  # plane: no mass
            return 0.0, com, np.zeros((3, 3))
        mass = g["mass"] if g["mass"] is not None else g["density"] * vol
        return mass, com, I * mass

    # -- bodies
    def _walk_body(self, e: ET.Element, parent: int, childclass: Optional[str]):
        childclass = e.get("childclass", childclass)
        bid = len(self.bodies)
        body = dict(
            name=e.get("name", f"body{bid}"), parent=parent,
            pos=_floats(e.get("pos", "0 0 0"), 3), quat=self._orientation(e.attrib),
            joints=[], geoms=[], inertial=None,
        )
        self.bodies.append(body)
        for c in e:
            if c.tag == "inertial":
                at = c.attrib
                I = dict(pos=_floats(at.get("pos", "0 0 0"), 3), quat=self._orientation(at), mass=float(at["mass"]))
                if "fullinertia" in at:
                    f = _floats(at["fullinertia"], 6)
                    M = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                    w, q = _eig3_frame(M)
                    I["diag"], I["quat"] = w, _quat_mul(I["quat"], q)
                else:
                    I["diag"] = _floats(at["diaginertia"], 3)
                d = I["diag"]
                if self.balanceinertia and (d[0] + d[1] < d[2] or d[0] + d[2] < d[1] or d[1] + d[2] < d[0]):
                    I["diag"] = np.full(3, d.mean())  # MuJoCo: triangle inequality violated -> all three set to the mean
                body["inertial"] = I
            elif c.tag in ("joint", "freejoint"):
                body["joints"].append(self._joint(c, childclass))
            elif c.tag == "geom":
                body["geoms"].append(self._geom(c, childclass))
        for c in e:
            if c.tag == "body":
                self._walk_body(c, bid, childclass)

    def _joint(self, e: ET.Element, childclass):
        if e.tag == "freejoint":
            return dict(name=e.get("name", ""), type=JNT_FREE, pos=np.zeros(3), axis=np.array([0, 0, 1.0]),
                        limited=False, range=np.zeros(2), damping=0.0, armature=0.0, stiffness=0.0,
                        ref=0.0, springref=0.0, margin=0.0, frictionloss=0.0,
                        solref=np.array([0.02, 1.0]), solimp=self._solimp(None),
                        solref_fri=np.array([0.02, 1.0]), solimp_fri=self._solimp(None))
        at = self.defaults.attrs("joint", e.get("class", childclass), e.attrib)
        jt = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}[at.get("type", "hinge")]
        rng = _floats(at.get("range", "0 0"), 2)
        lim = at.get("limited", "auto")
        if lim == "auto":
            limited = self.autolimits and "range" in at and not (rng[0] == 0 and rng[1] == 0)
            if not self.autolimits and "range" in at:
                limited = False
        else:
            limited = lim == "true"
        if jt in (JNT_HINGE, JNT_BALL):
            rng = self._angle(rng)
        axis = _floats(at.get("axis", "0 0 1"), 3)
        axis = axis / max(np.linalg.norm(axis), MINVAL)
        ref = float(at.get("ref", 0))
        springref = float(at.get("springref", 0))
        if jt == JNT_HINGE:
            ref, springref = self._angle(ref), self._angle(springref)
        frictionloss = float(at.get("frictionloss", 0))
        if frictionloss < 0:
            raise ValueError("joint frictionloss must be non-negative")
        if jt == JNT_BALL and limited and rng[0] != 0:
            raise ValueError("ball joint range must start at 0 (the limit is on the rotation angle)")
        return dict(
            name=at.get("name", ""), type=jt, pos=_floats(at.get("pos", "0 0 0"), 3), axis=axis,
            limited=limited, range=rng, damping=float(at.get("damping", 0)), armature=float(at.get("armature", 0)),
            stiffness=float(at.get("stiffness", 0)), ref=ref, springref=springref,
            margin=float(at.get("margin", 0)), frictionloss=frictionloss,
            solref=_floats(at.get("solreflimit", "0.02 1"), 2), solimp=self._solimp(at.get("solimplimit")),
            solref_fri=_floats(at.get("solreffriction", "0.02 1"), 2), solimp_fri=self._solimp(at.get("solimpfriction")),
        )

    # -- main
    def compile(self) -> MjModel:
        root = self.root
        self._load_meshes()
        self.bodies: List[dict] = []
        world = dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), joints=[], geoms=[], inertial=None)
        self.bodies.append(world)
        for wb in _merge_sections(root, "worldbody"):
            for c in wb:
                if c.tag == "geom":
                    world["geoms"].append(self._geom(c, None))
                elif c.tag == "body":
                    self._walk_body(c, 0, None)
        nbody = len(self.bodies)

        # MuJoCo numbers bodies depth-first in document order: _walk_body already did.
        m = MjModel()
        m.nbody = nbody
        m.names_body = [b["name"] for b in self.bodies]
        m.body_parentid = np.array([b["parent"] for b in self.bodies], dtype=np.int32)
        m.body_pos = np.stack([b["pos"] for b in self.bodies])
        m.body_quat = np.stack([b["quat"] for b in self.bodies])

        # joints / dofs
        jnt, names_joint = [], []
        body_jntadr, body_jntnum, body_dofadr, body_dofnum = [], [], [], []
        qadr, dadr = 0, 0
        jnt_qposadr, jnt_dofadr, jnt_bodyid = [], [], []
        dof_bodyid, dof_jntid, dof_parentid, dof_armature, dof_damping = [], [], [], [], []
        qpos0, qpos_spring = [], []
        last_dof_of_body = {0: -1}
        for b, body in enumerate(self.bodies):
            body_jntadr.append(len(jnt) if body["joints"] else -1)
            body_jntnum.append(len(body["joints"]))
            body_dofadr.append(dadr if body["joints"] else -1)
            parent_last = last_dof_of_body[body["parent"]] if b > 0 else -1
            prev = parent_last
            nd0 = dadr
            for j in body["joints"]:
                if j["type"] == JNT_FREE and (body["parent"] != 0 or len(body["joints"]) != 1):
                    raise ValueError("free joint must be the only joint of a top-level body")
                nq_j, nv_j = {JNT_FREE: (7, 6), JNT_BALL: (4, 3), JNT_SLIDE: (1, 1), JNT_HINGE: (1, 1)}[j["type"]]
                jid = len(jnt)
                jnt.append(j)
                names_joint.append(j["name"])
                jnt_qposadr.append(qadr)
                jnt_dofadr.append(dadr)
                jnt_bodyid.append(b)
                if j["type"] == JNT_FREE:
                    q0 = np.concatenate([body["pos"], body["quat"]])
                    qpos0 += list(q0)
                    qpos_spring += list(q0)
                elif j["type"] == JNT_BALL:
                    qpos0 += [1.0, 0, 0, 0]
                    qpos_spring += [1.0, 0, 0, 0]
                else:
                    qpos0.append(j["ref"])
                    qpos_spring.append(j["springref"])
                for k in range(nv_j):
                    dof_bodyid.append(b)
                    dof_jntid.append(jid)
                    dof_parentid.append(prev)
                    dof_armature.append(j["armature"])
                    dof_damping.append(j["damping"])
                    prev = dadr
                    dadr += 1
                qadr += nq_j
            body_dofnum.append(dadr - nd0)
            last_dof_of_body[b] = prev
        m.nq, m.nv, m.njnt = qadr, dadr, len(jnt)
        m.names_joint = names_joint
        m.body_jntadr = np.array(body_jntadr, dtype=np.int32)
        m.body_jntnum = np.array(body_jntnum, dtype=np.int32)
        m.body_dofadr = np.array(body_dofadr, dtype=np.int32)
        m.body_dofnum = np.array(body_dofnum, dtype=np.int32)
        I32 = lambda x: np.array(x, dtype=np.int32).reshape(-1)
        F = lambda x, shape: np.array(x, dtype=float).reshape(shape)
        m.jnt_type = I32([j["type"] for j in jnt])
        m.jnt_qposadr = I32(jnt_qposadr)
        m.jnt_dofadr = I32(jnt_dofadr)
        m.jnt_bodyid = I32(jnt_bodyid)
        m.jnt_pos = F([j["pos"] for j in jnt], (-1, 3))
        m.jnt_axis = F([j["axis"] for j in jnt], (-1, 3))
        m.jnt_limited = I32([int(j["limited"]) for j in jnt])
        m.jnt_range = F([j["range"] for j in jnt], (-1, 2))
        m.jnt_margin = F([j["margin"] for j in jnt], (-1,))
        m.jnt_stiffness = F([j["stiffness"] for j in jnt], (-1,))
        m.jnt_solref = F([j["solref"] for j in jnt], (-1, 2))
        m.jnt_solimp = F([j["solimp"] for j in jnt], (-1, 5))
        m.dof_bodyid, m.dof_jntid, m.dof_parentid = I32(dof_bodyid), I32(dof_jntid), I32(dof_parentid)
        m.dof_armature, m.dof_damping = F(dof_armature, (-1,)), F(dof_damping, (-1,))
        # dry friction: every dof of a joint carries the joint's frictionloss and its solreffriction / solimpfriction (MuJoCo's
        # dof_frictionloss, dof_solref, dof_solimp)
        m.dof_frictionloss = F([jnt[j]["frictionloss"] for j in dof_jntid], (-1,))
        m.dof_solref = F([jnt[j]["solref_fri"] for j in dof_jntid], (-1, 2))
        m.dof_solimp = F([jnt[j]["solimp_fri"] for j in dof_jntid], (-1, 5))
        m.qpos0, m.qpos_spring = F(qpos0, (-1,)), F(qpos_spring, (-1,))

        # root / weld ids
        rootid = np.zeros(nbody, dtype=np.int32)
        weldid = np.zeros(nbody, dtype=np.int32)
        for b in range(1, nbody):
            p = m.body_parentid[b]
            rootid[b] = b if p == 0 else rootid[p]
            weldid[b] = b if body_jntnum[b] > 0 else weldid[p]
        m.body_rootid, m.body_weldid = rootid, weldid

        # geoms
        geoms, geom_body = [], []
        for b, body in enumerate(self.bodies):
            for g in body["geoms"]:
                geoms.append(g)
                geom_body.append(b)
        m.ngeom = len(geoms)
        m.names_geom = [g["name"] for g in geoms]
        m.geom_type = I32([g["type"] for g in geoms])
        m.geom_bodyid = I32(geom_body)
        m.geom_contype = I32([g["contype"] for g in geoms])
        m.geom_conaffinity = I32([g["conaffinity"] for g in geoms])
        m.geom_condim = I32([g["condim"] for g in geoms])
        m.geom_priority = I32([g["priority"] for g in geoms])
        m.geom_pos = F([g["pos"] for g in geoms], (-1, 3))
        m.geom_quat = F([g["quat"] for g in geoms], (-1, 4))
        m.geom_size = F([g["size"] for g in geoms], (-1, 3))
        m.geom_friction = F([g["friction"] for g in geoms], (-1, 3))
        m.geom_solmix = F([g["solmix"] for g in geoms], (-1,))
        m.geom_solref = F([g["solref"] for g in geoms], (-1, 2))
        m.geom_solimp = F([g["solimp"] for g in geoms], (-1, 5))
        m.geom_margin = F([g["margin"] for g in geoms], (-1,))
        m.geom_gap = F([g["gap"] for g in geoms], (-1,))
        # meshes: convex-hull vertex pool, in each referencing geom's frame
        mesh_names = sorted({g["mesh"] for g in geoms if g["mesh"] is not None})
        vertadr, vertnum, pool = [], [], []
        for name in mesh_names:
            v = self.meshes[name]["vert"]
            vertadr.append(sum(vertnum))
            vertnum.append(len(v))
            pool.append(v)
        m.nmesh = len(mesh_names)
        m.names_mesh = mesh_names
        m.mesh_vertadr, m.mesh_vertnum = I32(vertadr), I32(vertnum)
        m.mesh_vert = np.concatenate(pool, axis=0) if pool else np.zeros((0, 3))
        m.mesh_faces = [np.asarray(self.meshes[name]["face"], dtype=int) for name in mesh_names]  # hull triangles, vertex ids local to the mesh
        m.geom_dataid = I32([mesh_names.index(g["mesh"]) if g["mesh"] is not None else -1 for g in geoms])

        # body inertial frames
        ipos, iquat, mass, inertia = np.zeros((nbody, 3)), np.tile([1.0, 0, 0, 0], (nbody, 1)), np.zeros(nbody), np.zeros((nbody, 3))
        for b, body in enumerate(self.bodies):
            if b == 0:
                continue
            use_geoms = self.inertiafromgeom == "true" or (self.inertiafromgeom == "auto" and body["inertial"] is None)
            if not use_geoms:
                if body["inertial"] is None:
                    continue
                I = body["inertial"]
                ipos[b], iquat[b], mass[b], inertia[b] = I["pos"], I["quat"], I["mass"], I["diag"]
                continue
            tot, com, parts = 0.0, np.zeros(3), []
            for g in body["geoms"]:
                if g["type"] == GEOM_PLANE:
                    continue
                gm, gcom, gI = self._geom_mass_props(g)
                if gm <= 0:
                    continue
                R = _quat_to_mat(g["quat"])
                c = g["pos"] + R @ gcom
                parts.append((gm, c, R @ gI @ R.T))
                tot += gm
                com += gm * c
            if tot <= 0:
                continue
            com /= tot
            Ib = np.zeros((3, 3))
            for gm, c, gI in parts:
                d = c - com
                Ib += gI + gm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            w, q = _eig3_frame(Ib)
            ipos[b], iquat[b], mass[b], inertia[b] = com, q, tot, w
        mass = np.maximum(mass, self.boundmass) * (np.arange(nbody) > 0)
        inertia = np.maximum(inertia, self.boundinertia) * (np.arange(nbody) > 0)[:, None]
        for b in range(1, nbody):
            if body_jntnum[b] > 0 and (mass[b] < MINVAL or np.any(inertia[b] < MINVAL)):
                raise ValueError(f"moving body '{m.names_body[b]}' needs mass and inertia")
        m.body_ipos, m.body_iquat, m.body_mass, m.body_inertia = ipos, iquat, mass, inertia
        sub = mass.copy()
        for b in range(nbody - 1, 0, -1):
            sub[m.body_parentid[b]] += sub[b]
        m.body_subtreemass = sub

        self._compile_actuators(m)
        self._compile_equality(m)
        self._compile_option(m)
        self._compile_excludes(m)
        self._compile_keyframes(m)
        _set_const(m)
        _make_collision_pairs(m)
        return m

    def _compile_actuators(self, m: MjModel):
        acts = []
        for sec in _merge_sections(self.root, "actuator"):
            for e in sec:
                if e.tag not in _ACTUATOR_TAGS:
                    raise NotImplementedError(f"actuator <{e.tag}>")
                at = self.defaults.attrs(e.tag, e.get("class"), e.attrib)
                if e.tag != "general":
                    at = {**self.defaults.attrs("general", e.get("class"), {}), **at}
                if "joint" not in at:
                    raise NotImplementedError("only joint transmissions are supported")
                jid = m.names_joint.index(at["joint"])
                if m.jnt_type[jid] not in (JNT_HINGE, JNT_SLIDE):
                    raise NotImplementedError("actuators on free/ball joints")
                gain, bias, biastype = np.zeros(3), np.zeros(3), BIAS_NONE
                gain[0] = 1.0
                if e.tag == "position":
                    kp = float(at.get("kp", 1))
                    kv = float(at.get("kv", 0))
                    gain[0], bias[:] , biastype = kp, [0, -kp, -kv], BIAS_AFFINE
                elif e.tag == "velocity":
                    kv = float(at.get("kv", 1))
                    gain[0], bias[:], biastype = kv, [0, 0, -kv], BIAS_AFFINE
                elif e.tag == "general":
                    if at.get("gaintype", "fixed") != "fixed" or at.get("dyntype", "none") != "none":
                        raise NotImplementedError("general actuator gaintype/dyntype")
                    g = _floats(at.get("gainprm", "1"))
                    gain[: min(3, g.size)] = g[:3]
                    bt = at.get("biastype", "none")
                    if bt == "affine":
                        bv = _floats(at.get("biasprm", "0 0 0"))
                        bias[: min(3, bv.size)] = bv[:3]
                        biastype = BIAS_AFFINE
                    elif bt != "none":
                        raise NotImplementedError("biastype " + bt)
                cr = _floats(at.get("ctrlrange", "0 0"), 2)
                cl = at.get("ctrllimited", "auto")
                ctrllimited = (self.autolimits and "ctrlrange" in at) if cl == "auto" else cl == "true"
                fr = _floats(at.get("forcerange", "0 0"), 2)
                fl = at.get("forcelimited", "auto")
                forcelimited = (self.autolimits and "forcerange" in at) if fl == "auto" else fl == "true"
                gear = _floats(at.get("gear", "1"))[0]
                acts.append(dict(name=at.get("name", ""), jid=jid, gear=gear, gain=gain, bias=bias, biastype=biastype,
                                 ctrllimited=ctrllimited, ctrlrange=cr, forcelimited=forcelimited, forcerange=fr))
        m.nu = len(acts)
        m.names_actuator = [a["name"] for a in acts]
        m.actuator_trnid = np.array([a["jid"] for a in acts], dtype=np.int32).reshape(-1)
        m.actuator_gear = np.array([a["gear"] for a in acts], dtype=float).reshape(-1)
        m.actuator_gainprm = np.array([a["gain"] for a in acts], dtype=float).reshape(-1, 3)
        m.actuator_biasprm = np.array([a["bias"] for a in acts], dtype=float).reshape(-1, 3)
        m.actuator_biastype = np.array([a["biastype"] for a in acts], dtype=np.int32).reshape(-1)
        m.actuator_ctrllimited = np.array([int(a["ctrllimited"]) for a in acts], dtype=np.int32).reshape(-1)
        m.actuator_ctrlrange = np.array([a["ctrlrange"] for a in acts], dtype=float).reshape(-1, 2)
        m.actuator_forcelimited = np.array([int(a["forcelimited"]) for a in acts], dtype=np.int32).reshape(-1)
        m.actuator_forcerange = np.array([a["forcerange"] for a in acts], dtype=float).reshape(-1, 2)

    def _compile_equality(self, m: MjModel):
        """<equality>: joint (polycoef), connect and weld.  eq_data has MuJoCo's 11 columns; what they hold here:
          joint   [0:5] polycoef
          connect [0:3] anchor in body1's frame, [3:6] the same point in body2's frame at qpos0
          weld    [0:3] the weld point in body2's frame (`anchor`), [3:6] the same point in body1's frame, [6:10] relpose quaternion
                  (orientation of body2 in body1's frame), [10] torquescale
        obj ids are joint ids for joint equalities and body ids for connect / weld (body2 omitted = the world)."""
        xpos0, xquat0 = _kinematics_qpos0(m)
        eqs = []
        for sec in _merge_sections(self.root, "equality"):
            for e in sec:
                at = self.defaults.attrs("equality", e.get("class"), e.attrib)
                common = dict(name=at.get("name", ""), active=at.get("active", "true") == "true",
                              solref=_floats(at.get("solref", "0.02 1"), 2), solimp=self._solimp(at.get("solimp")))
                data = np.zeros(11)
                if e.tag == "joint":
                    j1 = m.names_joint.index(at["joint1"])
                    j2 = m.names_joint.index(at["joint2"]) if "joint2" in at else -1
                    for j in (j1, j2):
                        if j >= 0 and m.jnt_type[j] not in (JNT_HINGE, JNT_SLIDE):
                            raise ValueError("joint equality needs scalar joints")
                    data[:5] = [0.0, 1.0, 0, 0, 0]
                    if "polycoef" in at:
                        u = _floats(at["polycoef"])
                        data[: u.size] = u
                    eqs.append(dict(type=EQ_JOINT, o1=j1, o2=j2, data=data, **common))
                elif e.tag in ("connect", "weld"):
                    if "site1" in at or "site2" in at:
                        raise NotImplementedError(f"equality <{e.tag}> between sites (give body1 / body2)")
                    b1 = m.names_body.index(at["body1"])
                    b2 = m.names_body.index(at["body2"]) if "body2" in at else 0
                    inv = lambda b, p: _rot(_quat_conj(xquat0[b]), p - xpos0[b])  # world point -> body frame at qpos0
                    if e.tag == "connect":
                        a1 = _floats(at["anchor"], 3)
                        data[0:3] = a1
                        data[3:6] = inv(b2, xpos0[b1] + _rot(xquat0[b1], a1))
                        eqs.append(dict(type=EQ_CONNECT, o1=b1, o2=b2, data=data, **common))
                    else:
                        a2 = _floats(at.get("anchor", "0 0 0"), 3)
                        rel = _floats(at.get("relpose", "0 1 0 0 0 0 0"), 7)
                        if np.all(rel[3:] == 0):  # MuJoCo: an all-zero quaternion means "the relative pose at qpos0"
                            rq = _quat_mul(_quat_conj(xquat0[b1]), xquat0[b2])
                            rp = inv(b1, xpos0[b2])
                        else:
                            rq = rel[3:] / np.linalg.norm(rel[3:])
                            rp = rel[:3]
                        data[0:3] = a2
                        data[3:6] = rp + _rot(rq, a2)  # the weld point in body1's frame when the relative pose is relpose
                        data[6:10] = rq
                        data[10] = float(at.get("torquescale", 1.0))
                        eqs.append(dict(type=EQ_WELD, o1=b1, o2=b2, data=data, **common))
                else:
                    raise NotImplementedError(f"equality <{e.tag}>")
        m.neq = len(eqs)
        m.names_eq = [q["name"] for q in eqs]
        m.eq_type = np.array([q["type"] for q in eqs], dtype=np.int32).reshape(-1)
        m.eq_obj1id = np.array([q["o1"] for q in eqs], dtype=np.int32).reshape(-1)
        m.eq_obj2id = np.array([q["o2"] for q in eqs], dtype=np.int32).reshape(-1)
        m.eq_active = np.array([int(q["active"]) for q in eqs], dtype=np.int32).reshape(-1)
        m.eq_data = np.array([q["data"] for q in eqs], dtype=float).reshape(-1, 11)
        m.eq_solref = np.array([q["solref"] for q in eqs], dtype=float).reshape(-1, 2)
        m.eq_solimp = np.array([q["solimp"] for q in eqs], dtype=float).reshape(-1, 5)

    def _compile_option(self, m: MjModel):
        at, flags = {}, {}
        for o in _merge_sections(self.root, "option"):
            at.update(o.attrib)
            for f in o:
                if f.tag == "flag":
                    flags.update(f.attrib)
        disable = 0
        for k, v in flags.items():
            if k in _FLAG_BITS:
                if v == "disable":
                    disable |= _FLAG_BITS[k]
            elif v == "enable":
                raise NotImplementedError(f"enable flag {k}")
        solver = {"pgs": SOL_PGS, "cg": SOL_CG, "newton": SOL_NEWTON}[at.get("solver", "Newton").lower()]
        integ = {"euler": INT_EULER, "rk4": INT_RK4, "implicit": INT_IMPLICIT, "implicitfast": INT_IMPLICITFAST}[at.get("integrator", "Euler").lower()]
        if at.get("cone", "pyramidal") != "pyramidal":
            raise NotImplementedError("elliptic friction cones (reference ambersim/utils/io_utils.py:235)")
        m.opt = MjOption(
            timestep=float(at.get("timestep", 0.002)), gravity=_floats(at.get("gravity", "0 0 -9.81"), 3),
            tolerance=float(at.get("tolerance", 1e-8)), ls_tolerance=float(at.get("ls_tolerance", 0.01)),
            impratio=float(at.get("impratio", 1)), iterations=int(at.get("iterations", 100)),
            ls_iterations=int(at.get("ls_iterations", 50)), solver=solver, integrator=integ,
            disableflags=disable, cone=0,
        )

    def _compile_excludes(self, m: MjModel):
        ex = []
        for sec in _merge_sections(self.root, "contact"):
            for e in sec:
                if e.tag == "exclude":
                    ex.append((m.names_body.index(e.get("body1")), m.names_body.index(e.get("body2"))))
                elif e.tag == "pair":
                    raise NotImplementedError("explicit contact pairs")
        m.exclude_pairs = ex

    def _compile_keyframes(self, m: MjModel):
        keys = []
        for sec in _merge_sections(self.root, "keyframe"):
            for e in sec:
                if e.tag == "key":
                    keys.append(dict(
                        name=e.get("name", ""),
                        qpos=_floats(e.get("qpos"), m.nq) if e.get("qpos") else m.qpos0.copy(),
                        qvel=_floats(e.get("qvel"), m.nv) if e.get("qvel") else np.zeros(m.nv),
                        ctrl=_floats(e.get("ctrl"), m.nu) if e.get("ctrl") else np.zeros(m.nu)))
        m.nkey = len(keys)
        m.names_key = [k["name"] for k in keys]
        m.key_qpos = np.array([k["qpos"] for k in keys], dtype=float).reshape(len(keys), m.nq)
        m.key_qvel = np.array([k["qvel"] for k in keys], dtype=float).reshape(len(keys), m.nv)
        m.key_ctrl = np.array([k["ctrl"] for k in keys], dtype=float).reshape(len(keys), m.nu)


# ----------------------------------------------------------------------------- derived constants
def _kinematics_qpos0(m: MjModel):
    """World poses at qpos0 (used only to derive compile-time constants)."""
    nb = m.nbody
    xpos, xquat = np.zeros((nb, 3)), np.tile([1.0, 0, 0, 0], (nb, 1))
    for b in range(1, nb):
        p = m.body_parentid[b]
        # at qpos0 every joint is at its reference, so the body sits at its MJCF pose
        xpos[b] = xpos[p] + _rot(xquat[p], m.body_pos[b])
        xquat[b] = _quat_mul(xquat[p], m.body_quat[b])
    return xpos, xquat


def mass_matrix_qpos0(m: MjModel):
    """Dense M(qpos0), plus per-dof world axes/anchors; composite-rigid-body on the host."""
    xpos, xquat = _kinematics_qpos0(m)
    nb, nv = m.nbody, m.nv
    xipos = np.array([xpos[b] + _rot(xquat[b], m.body_ipos[b]) for b in range(nb)])
    # spatial inertia about world origin, 6x6 [rot; lin]
    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    Ic = np.zeros((nb, 6, 6))
    for b in range(1, nb):
        R = _quat_to_mat(_quat_mul(xquat[b], m.body_iquat[b]))
        I3 = R @ np.diag(m.body_inertia[b]) @ R.T
        c, ms = xipos[b], m.body_mass[b]
        S = skew(c)
        Ic[b, :3, :3] = I3 - ms * S @ S
        Ic[b, :3, 3:] = ms * S
        Ic[b, 3:, :3] = -ms * S
        Ic[b, 3:, 3:] = ms * np.eye(3)
    for b in range(nb - 1, 0, -1):
        Ic[m.body_parentid[b]] += Ic[b]
    # motion subspace per dof about world origin: [w; v]
    S = np.zeros((nv, 6))
    for j in range(m.njnt):
        b, d = m.jnt_bodyid[j], m.jnt_dofadr[j]
        R = _quat_to_mat(xquat[b])
        anchor = xpos[b] + R @ m.jnt_pos[j]
        axis = R @ m.jnt_axis[j]
        t = m.jnt_type[j]
        if t == JNT_FREE:
            for k in range(3):
                S[d + k, 3 + k] = 1.0
                ax = R[:, k]
                S[d + 3 + k, :3] = ax
                S[d + 3 + k, 3:] = np.cross(xpos[b], ax)
        elif t == JNT_BALL:
            for k in range(3):
                ax = R[:, k]
                S[d + k, :3] = ax
                S[d + k, 3:] = np.cross(anchor, ax)
        elif t == JNT_HINGE:
            S[d, :3] = axis
            S[d, 3:] = np.cross(anchor, axis)
        else:
            S[d, 3:] = axis
    M = np.zeros((nv, nv))
    for i in range(nv):
        f = Ic[m.dof_bodyid[i]] @ S[i]
        j = i
        while j >= 0:
            M[i, j] = M[j, i] = S[j] @ f
            j = m.dof_parentid[j]
    M += np.diag(m.dof_armature)
    return M, S, xpos, xquat, xipos


def _set_const(m: MjModel):
    """dof_invweight0, body_invweight0, stat.meaninertia (MuJoCo's mj_setConst/set0 semantics:
    diagonal of M^-1 resp. J M^-1 J^T at qpos0, averaged over the axes of free/ball joints)."""
    nv, nb = m.nv, m.nbody
    m.stat = MjStatistic(meaninertia=1.0)
    m.dof_invweight0 = np.zeros(nv)
    m.body_invweight0 = np.zeros((nb, 2))
    if nv == 0:
        return
    M, S, xpos, xquat, xipos = mass_matrix_qpos0(m)
    m.stat.meaninertia = float(np.mean(np.diag(M)))
    Minv = np.linalg.inv(M)
    d = np.diag(Minv).copy()
    for j in range(m.njnt):
        a = m.jnt_dofadr[j]
        if m.jnt_type[j] == JNT_FREE:
            d[a:a + 3] = d[a:a + 3].mean()
            d[a + 3:a + 6] = d[a + 3:a + 6].mean()
        elif m.jnt_type[j] == JNT_BALL:
            d[a:a + 3] = d[a:a + 3].mean()
    m.dof_invweight0 = np.maximum(d, MINVAL)
    for b in range(1, nb):
        if m.body_weldid[b] == 0:
            continue
        # 6 x nv Jacobian of the body's centre of mass: rows [translational(3); rotational(3)]
        J = np.zeros((6, nv))
        i = m.body_dofadr[m.body_weldid[b]] + m.body_dofnum[m.body_weldid[b]] - 1
        while i >= 0:
            w, v0 = S[i, :3], S[i, 3:]
            J[:3, i] = v0 + np.cross(w, xipos[b])
            J[3:, i] = w
            i = m.dof_parentid[i]
        A = J @ Minv @ J.T
        m.body_invweight0[b, 0] = max(np.trace(A[:3, :3]) / 3.0, MINVAL)
        m.body_invweight0[b, 1] = max(np.trace(A[3:, 3:]) / 3.0, MINVAL)


# ----------------------------------------------------------------------------- collision pairs
# Pair functions the hot path implements; ids shared with oracle/ and csrc/.
PAIR_PLANE_SPHERE, PAIR_PLANE_CAPSULE, PAIR_PLANE_CONVEX = 0, 1, 2
PAIR_SPHERE_SPHERE, PAIR_SPHERE_CAPSULE, PAIR_CAPSULE_CAPSULE = 3, 4, 5
PAIR_CONVEX_CONVEX = 6  # box / mesh hull against box / mesh hull: SAT + face clipping
PAIR_SPHERE_CONVEX = 7  # sphere against a box / mesh hull: closest point of the best-supported face
PAIR_CAPSULE_CONVEX = 11  # capsule against a box / mesh hull (r04): the axis clipped to the best-supported face (two contacts), or the closest edge (8 - 10 are the planner's row groups)
PAIR_UNSUPPORTED = -1  # condim != 3, hulls beyond the size limits
_PAIR_NCON = {0: 1, 1: 2, 2: 4, 3: 1, 4: 1, 5: 1, 6: 4, 7: 1, 11: 2, -1: 0}
HULL_MAX_VERT, HULL_MAX_FACE, HULL_MAX_FACEVERT, HULL_MAX_EDGE = 64, 128, 8, 192  # a triangulated 64-vertex hull has 124 faces and 186 edges
_BOX_CORNERS = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=float)


_BOX_FACES = [[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1], [2, 3, 7, 6], [0, 2, 6, 4], [1, 5, 7, 3]]  # over _BOX_CORNERS, outward CCW


def _hull_polygons(verts: np.ndarray, tris: np.ndarray):
    """Polygon faces of a convex hull given as outward triangles: coplanar triangles are merged (as the reference physics'
    mesh preprocessing does), each face's vertices ordered counter-clockwise seen from outside.  Returns (faces: list of
    vertex-id lists, normals [nface,3], edges [nedge,2] unique vertex pairs)."""
    scale = max(1e-12, np.abs(verts).max())
    groups = []  # (normal, offset, set of vertex ids)
    for t in tris:
        a, b, c = verts[t[0]], verts[t[1]], verts[t[2]]
        n = np.cross(b - a, c - a)
        ln = np.linalg.norm(n)
        if ln < 1e-14 * scale * scale:
            continue
        n = n / ln
        off = float(n @ a)
        for g in groups:
            if g[0] @ n > 1 - 1e-8 and abs(g[1] - off) < 1e-7 * scale:
                g[2].update(int(x) for x in t)
                break
        else:
            groups.append([n, off, {int(x) for x in t}])
    faces, normals = [], []
    for n, off, ids in groups:
        ids = sorted(ids)
        c = verts[ids].mean(axis=0)
        u = verts[ids[0]] - c
        u = u / np.linalg.norm(u)
        w = np.cross(n, u)
        ang = [np.arctan2((verts[i] - c) @ w, (verts[i] - c) @ u) for i in ids]
        order = [i for _, i in sorted(zip(ang, ids))]  # counter-clockwise about the outward normal
        k = order.index(min(order))
        faces.append(order[k:] + order[:k])  # canonical start: the lowest vertex id
        normals.append(n)
    edges = sorted({(min(f[k], f[(k + 1) % len(f)]), max(f[k], f[(k + 1) % len(f)])) for f in faces for k in range(len(f))})
    return faces, np.array(normals, dtype=float).reshape(-1, 3), np.array(edges, dtype=int).reshape(-1, 2)


def _geom_hull(m: "MjModel", g: int):
    """(verts, faces, normals, edges) of a box or mesh geom in its own frame, or None when it exceeds the hull limits."""
    if m.geom_type[g] == GEOM_BOX:
        v = _BOX_CORNERS * m.geom_size[g]
        faces = [list(f) for f in _BOX_FACES]
        normals = np.array([np.cross(v[f[1]] - v[f[0]], v[f[2]] - v[f[0]]) for f in faces])
        normals = normals / np.linalg.norm(normals, axis=1, keepdims=True)
        edges = np.array(sorted({(min(f[k], f[(k + 1) % 4]), max(f[k], f[(k + 1) % 4])) for f in faces for k in range(4)}), dtype=int)
    else:
        mid = m.geom_dataid[g]
        v = m.mesh_vert[m.mesh_vertadr[mid]: m.mesh_vertadr[mid] + m.mesh_vertnum[mid]]
        if len(v) > HULL_MAX_VERT:
            return None
        faces, normals, edges = _hull_polygons(v, m.mesh_faces[mid])
    if len(v) > HULL_MAX_VERT or len(faces) > HULL_MAX_FACE or len(edges) > HULL_MAX_EDGE or max(len(f) for f in faces) > HULL_MAX_FACEVERT:
        return None
    return v, faces, normals, edges


def _make_collision_pairs(m: MjModel):
    """Static broadphase: candidate geom pairs are fixed at compile time (the reference physics
    fixes them at trace time), with MuJoCo's contype/conaffinity, same-body and
    parent-child filters and its per-pair parameter mixing."""
    pairs = []
    box_vert_adr = {}
    extra = []
    hulls, hull_of_geom = [], {}  # convex hulls of the geoms in convex-convex pairs

    def hull_id(g):
        if g not in hull_of_geom:
            h = _geom_hull(m, g)
            hull_of_geom[g] = -1 if h is None else len(hulls)
            if h is not None:
                hulls.append(h)
        return hull_of_geom[g]
    nvert0 = m.mesh_vert.shape[0]
    contact_on = not (m.opt.disableflags & (DisableBit.CONTACT | DisableBit.CONSTRAINT))
    for g1 in range(m.ngeom):
        for g2 in range(g1 + 1, m.ngeom):
            if not contact_on:
                continue
            if not ((m.geom_contype[g1] & m.geom_conaffinity[g2]) or (m.geom_contype[g2] & m.geom_conaffinity[g1])):
                continue
            b1, b2 = m.geom_bodyid[g1], m.geom_bodyid[g2]
            w1, w2 = m.body_weldid[b1], m.body_weldid[b2]
            if w1 == w2:
                continue
            if not (m.opt.disableflags & DisableBit.FILTERPARENT) and w1 != 0 and w2 != 0:
                if m.body_weldid[m.body_parentid[w1]] == w2 or m.body_weldid[m.body_parentid[w2]] == w1:
                    continue
            if (b1, b2) in m.exclude_pairs or (b2, b1) in m.exclude_pairs:
                continue
            a, b = (g1, g2) if m.geom_type[g1] <= m.geom_type[g2] else (g2, g1)
            ta, tb = m.geom_type[a], m.geom_type[b]
            if ta == GEOM_PLANE and tb == GEOM_PLANE:
                continue
            convex = (GEOM_BOX, GEOM_MESH)
            if ta == GEOM_PLANE and tb == GEOM_SPHERE:
                fn = PAIR_PLANE_SPHERE
            elif ta == GEOM_PLANE and tb == GEOM_CAPSULE:
                fn = PAIR_PLANE_CAPSULE
            elif ta == GEOM_PLANE and tb in convex:
                fn = PAIR_PLANE_CONVEX
            elif ta == GEOM_SPHERE and tb == GEOM_SPHERE:
                fn = PAIR_SPHERE_SPHERE
            elif ta == GEOM_SPHERE and tb == GEOM_CAPSULE:
                fn = PAIR_SPHERE_CAPSULE
            elif ta == GEOM_CAPSULE and tb == GEOM_CAPSULE:
                fn = PAIR_CAPSULE_CAPSULE
            elif ta in convex and tb in convex and hull_id(a) >= 0 and hull_id(b) >= 0:
                fn = PAIR_CONVEX_CONVEX
            elif ta == GEOM_SPHERE and tb in convex and hull_id(b) >= 0:
                fn = PAIR_SPHERE_CONVEX
            elif ta == GEOM_CAPSULE and tb in convex and hull_id(b) >= 0:
                fn = PAIR_CAPSULE_CONVEX
            else:
                # the host model still loads (as MuJoCo's compiler would); putting it on the device raises
                # NotImplementedError, like mjx.device_put does for unsupported collisions (reference io_utils.py:228-241)
                fn = PAIR_UNSUPPORTED
            if max(m.geom_condim[a], m.geom_condim[b]) != 3:
                fn = PAIR_UNSUPPORTED
            # parameter mixing (MuJoCo mj_contactParam)
            pa, pb = m.geom_priority[a], m.geom_priority[b]
            if pa != pb:
                h = a if pa > pb else b
                fri, solref, solimp = m.geom_friction[h], m.geom_solref[h], m.geom_solimp[h]
            else:
                fri = np.maximum(m.geom_friction[a], m.geom_friction[b])
                sa, sb = m.geom_solmix[a], m.geom_solmix[b]
                if sa >= MINVAL and sb >= MINVAL:
                    mix = sa / (sa + sb)
                elif sa < MINVAL and sb < MINVAL:
                    mix = 0.5
                else:
                    mix = 0.0 if sa < MINVAL else 1.0
                ra, rb = m.geom_solref[a], m.geom_solref[b]
                if ra[0] > 0 and rb[0] > 0:
                    solref = mix * ra + (1 - mix) * rb
                else:
                    solref = np.minimum(ra, rb)
                solimp = mix * m.geom_solimp[a] + (1 - mix) * m.geom_solimp[b]
            margin = max(m.geom_margin[a], m.geom_margin[b])
            gap = max(m.geom_gap[a], m.geom_gap[b])
            vadr, vnum = -1, 0
            if fn == PAIR_PLANE_CONVEX:
                if tb == GEOM_MESH:
                    vadr, vnum = m.mesh_vertadr[m.geom_dataid[b]], m.mesh_vertnum[m.geom_dataid[b]]
                else:
                    if b not in box_vert_adr:
                        box_vert_adr[b] = nvert0 + 8 * len(extra)
                        extra.append(_BOX_CORNERS * m.geom_size[b])
                    vadr, vnum = box_vert_adr[b], 8
            h1, h2 = (hull_id(a), hull_id(b)) if fn == PAIR_CONVEX_CONVEX else ((-1, hull_id(b)) if fn in (PAIR_SPHERE_CONVEX, PAIR_CAPSULE_CONVEX) else (-1, -1))
            pairs.append(dict(g1=a, g2=b, fn=fn, ncon=_PAIR_NCON[fn], friction=fri[:3].copy(), solref=np.array(solref),
                              solimp=np.array(solimp), includemargin=margin - gap, vadr=vadr, vnum=vnum, h1=h1, h2=h2))
    # contacts are grouped by pair function (as the reference physics groups them), stable in geom order
    pairs.sort(key=lambda p: p["fn"])
    m.nunsupported_pairs = sum(1 for p in pairs if p["fn"] < 0)
    m.npair = len(pairs)
    m.pair_geom1 = np.array([p["g1"] for p in pairs], dtype=np.int32).reshape(-1)
    m.pair_geom2 = np.array([p["g2"] for p in pairs], dtype=np.int32).reshape(-1)
    m.pair_fn = np.array([p["fn"] for p in pairs], dtype=np.int32).reshape(-1)
    m.pair_ncon = np.array([p["ncon"] for p in pairs], dtype=np.int32).reshape(-1)
    m.pair_friction = np.array([p["friction"] for p in pairs], dtype=float).reshape(-1, 3)
    m.pair_solref = np.array([p["solref"] for p in pairs], dtype=float).reshape(-1, 2)
    m.pair_solimp = np.array([p["solimp"] for p in pairs], dtype=float).reshape(-1, 5)
    m.pair_margin = np.array([p["includemargin"] for p in pairs], dtype=float).reshape(-1)
    m.pair_vertadr = np.array([p["vadr"] for p in pairs], dtype=np.int32).reshape(-1)
    m.pair_vertnum = np.array([p["vnum"] for p in pairs], dtype=np.int32).reshape(-1)
    m.convex_vert = np.concatenate([m.mesh_vert] + extra, axis=0) if extra else m.mesh_vert.copy()
    m.ncon = int(m.pair_ncon.sum())
    # hull tables of the convex-convex pairs (amx_model_desc: pair_hull*, hull_*)
    used = sorted({h for p in pairs for h in (p["h1"], p["h2"]) if h >= 0})
    remap = {h: k for k, h in enumerate(used)}
    m.pair_hull1 = np.array([remap.get(p["h1"], -1) for p in pairs], dtype=np.int32).reshape(-1)
    m.pair_hull2 = np.array([remap.get(p["h2"], -1) for p in pairs], dtype=np.int32).reshape(-1)
    va, vn, fa, fnum, ea, en, fva, fvn = [], [], [], [], [], [], [], []
    hv, hn, hfv, he = [], [], [], []
    for h in used:
        v, faces, normals, edges = hulls[h]
        va.append(sum(vn)); vn.append(len(v)); fa.append(sum(fnum)); fnum.append(len(faces)); ea.append(sum(en)); en.append(len(edges))
        hv.append(v); hn.append(normals); he.append(edges)
        for f in faces:
            fva.append(len(hfv)); fvn.append(len(f)); hfv.extend(f)
    I32 = lambda x: np.array(x, dtype=np.int32).reshape(-1)
    m.hull_vertadr, m.hull_vertnum, m.hull_faceadr, m.hull_facenum, m.hull_edgeadr, m.hull_edgenum = I32(va), I32(vn), I32(fa), I32(fnum), I32(ea), I32(en)
    m.hull_vert = np.concatenate(hv, axis=0) if hv else np.zeros((0, 3))
    m.hull_face_normal = np.concatenate(hn, axis=0) if hn else np.zeros((0, 3))
    m.hull_face_vertadr, m.hull_face_vertnum, m.hull_facevert = I32(fva), I32(fvn), I32(hfv)
    m.hull_edge = np.concatenate(he, axis=0).astype(np.int32) if he else np.zeros((0, 2), dtype=np.int32)


def recompile_options(m: MjModel) -> MjModel:
    """Re-derives option-dependent static tables after `opt` was replaced (e.g. disableflags)."""
    m = copy.copy(m)
    _make_collision_pairs(m)
    return m


def load_mjcf_tree(path) -> ET.Element:
    """MJCF element tree with `<include>`s resolved; a URDF is converted first (ambersim_amd/urdf.py)."""
    p = Path(path)
    if p.suffix.lower() == ".urdf":  # mujoco.MjModel.from_xml_path also accepts URDF (reference conversion_utils.py:26)
        from ambersim_amd.urdf import urdf_to_mjcf
        return urdf_to_mjcf(p)
    root = ET.parse(p).getroot()
    if root.tag != "mujoco":
        raise ValueError("not an MJCF file")
    _resolve_includes(root, p.parent)
    return root


def compile_mjcf_tree(root: ET.Element, model_dir) -> MjModel:
    return _Compiler(root, Path(model_dir)).compile()


def compile_mjcf(path: str) -> MjModel:
    return compile_mjcf_tree(load_mjcf_tree(path), Path(path).parent)
