"""URDF -> MJCF element tree, following MuJoCo's URDF import rules [UPSTREAM-RECALL: MuJoCo "URDF extensions"].

The reference never parses URDF kinematics itself: `mj.MjModel.from_xml_path(urdf)` does it inside MuJoCo
(reference ambersim/utils/conversion_utils.py:24-29), and the reference then patches actuators and mimic
equalities into the saved XML (ambersim/utils/io_utils.py:18-117).  MuJoCo is absent here, so this module
restates the import: one body per link, child pose from the joint `<origin>`, joint at the child's origin,
`<limit>` -> range, `<dynamics>` -> damping/frictionloss, `<inertial>` -> pos/quat/mass/fullinertia,
`<collision>` -> colliding geom, `<visual>` -> non-colliding geom (kept only with discardvisual="false"),
meshes by basename under `meshdir` (strippath), and the `<mujoco><compiler .../></mujoco>` extension element
(fusestatic, discardvisual, meshdir, strippath, balanceinertia; URDF defaults fusestatic = discardvisual =
strippath = true).  Angles are radians; rpy is fixed-axis roll-pitch-yaw.
"""
import re
import xml.etree.ElementTree as ET
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np


def parse_urdf(path) -> ET.Element:
    """Parses a URDF, tolerating undeclared namespace prefixes such as `drake:declare_convex` (the reference uses
    lxml's recover mode for the same reason, ambersim/utils/io_utils.py:29-32)."""
    text = Path(path).read_text()
    declared = set(re.findall(r"xmlns:([A-Za-z_][\w.-]*)\s*=", text))

    def fix(m):
        return m.group(0) if m.group(2) in declared else f"<{m.group(1)}{m.group(2)}_"

    text = re.sub(r"<(/?)([A-Za-z_][\w.-]*):", fix, text)
    root = ET.fromstring(text)
    if root.tag != "robot":
        raise ValueError("not a URDF file (root element must be <robot>)")
    return root


def _f(s: Optional[str], n: int, default: float = 0.0) -> np.ndarray:
    if s is None:
        return np.full(n, default)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if v.size != n:
        raise ValueError(f"expected {n} numbers, got '{s}'")
    return v


def _rpy_quat(rpy: np.ndarray) -> np.ndarray:
    """Fixed-axis roll (x), pitch (y), yaw (z): R = Rz(yaw) Ry(pitch) Rx(roll); returns (w, x, y, z)."""
    cr, sr = np.cos(rpy[0] / 2), np.sin(rpy[0] / 2)
    cp, sp = np.cos(rpy[1] / 2), np.sin(rpy[1] / 2)
    cy, sy = np.cos(rpy[2] / 2), np.sin(rpy[2] / 2)
    return np.array([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy])


def _qmul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _qmat(q):
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _origin(e: Optional[ET.Element]) -> Tuple[np.ndarray, np.ndarray]:
    o = e.find("origin") if e is not None else None
    if o is None:
        return np.zeros(3), np.array([1.0, 0, 0, 0])
    return _f(o.get("xyz"), 3), _rpy_quat(_f(o.get("rpy"), 3))


def _fmt(v) -> str:
    return " ".join(repr(float(x)) for x in np.atleast_1d(v))


class _Link:
    def __init__(self, e: ET.Element):
        self.name = e.get("name")
        self.elem = e
        self.joint: Optional[ET.Element] = None  # the joint that attaches this link to its parent
        self.children: List["_Link"] = []


def _inertial(link: ET.Element) -> Optional[dict]:
    i = link.find("inertial")
    if i is None:
        return None
    pos, quat = _origin(i)
    mass = float(i.find("mass").get("value"))
    t = i.find("inertia")
    g = lambda k: float(t.get(k, 0.0))
    I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
    return dict(pos=pos, quat=quat, mass=mass, I=I)


def _merge_inertials(parts: List[Tuple[np.ndarray, np.ndarray, dict]]) -> Optional[dict]:
    """Combines inertials given as (pos, quat of the owner's frame in the target frame, inertial) into one inertial
    expressed in the target frame (fusestatic)."""
    parts = [(p, q, i) for p, q, i in parts if i is not None and i["mass"] > 0]
    if not parts:
        return None
    mass = sum(i["mass"] for _, _, i in parts)
    coms, Is = [], []
    for p, q, i in parts:
        R = _qmat(q)
        coms.append(p + R @ i["pos"])
        Rt = R @ _qmat(i["quat"])
        Is.append(Rt @ i["I"] @ Rt.T)
    com = sum(i["mass"] * c for (_, _, i), c in zip(parts, coms)) / mass
    I = np.zeros((3, 3))
    for (_, _, i), c, Ic in zip(parts, coms, Is):
        d = c - com
        I += Ic + i["mass"] * (d @ d * np.eye(3) - np.outer(d, d))
    return dict(pos=com, quat=np.array([1.0, 0, 0, 0]), mass=mass, I=I)


class _Converter:
    def __init__(self, root: ET.Element, base: Path):
        self.root, self.base = root, base
        comp: Dict[str, str] = {}
        for mj in root.findall("mujoco"):
            for c in mj.findall("compiler"):
                comp.update(c.attrib)
        self.comp = comp
        self.fusestatic = comp.get("fusestatic", "true") == "true"
        self.discardvisual = comp.get("discardvisual", "true") == "true"
        self.strippath = comp.get("strippath", "true") == "true"
        self.meshes: Dict[Tuple[str, str], str] = {}  # (file, scale) -> asset name
        self.mesh_elems: List[ET.Element] = []

    # -- geoms
    def _mesh_asset(self, filename: str, scale: Optional[str]) -> str:
        f = filename
        for prefix in ("package://", "file://"):
            if f.startswith(prefix):
                f = f[len(prefix):]
        if self.strippath:
            f = Path(f).name
        key = (f, scale or "")
        if key not in self.meshes:
            name = Path(f).stem
            if name in self.meshes.values():  # same file at another scale
                name = f"{name}_{len(self.meshes)}"
            self.meshes[key] = name
            at = dict(name=name, file=f)
            if scale:
                at["scale"] = scale
            self.mesh_elems.append(ET.Element("mesh", at))
        return self.meshes[key]

    def _geoms(self, link: ET.Element, pos0: np.ndarray, quat0: np.ndarray) -> List[ET.Element]:
        """Geoms of one link, posed in the frame (pos0, quat0) relative to the body that will own them."""
        out = []
        kinds = ([] if self.discardvisual else [("visual", False)]) + [("collision", True)]  # MuJoCo: visuals first
        R0 = _qmat(quat0)
        for tag, collide in kinds:
            for k, e in enumerate(link.findall(tag)):
                geo = e.find("geometry")
                if geo is None or len(geo) == 0:
                    continue
                shape = geo[0]
                at: Dict[str, str] = {}
                name = e.get("name")
                if name:
                    at["name"] = name
                if shape.tag == "box":
                    at["type"], at["size"] = "box", _fmt(0.5 * _f(shape.get("size"), 3))
                elif shape.tag == "sphere":
                    at["type"], at["size"] = "sphere", _fmt(float(shape.get("radius")))
                elif shape.tag in ("cylinder", "capsule"):
                    at["type"] = shape.tag
                    at["size"] = _fmt([float(shape.get("radius")), 0.5 * float(shape.get("length"))])
                elif shape.tag == "mesh":
                    at["type"], at["mesh"] = "mesh", self._mesh_asset(shape.get("filename"), shape.get("scale"))
                else:
                    raise NotImplementedError(f"URDF geometry <{shape.tag}>")
                p, q = _origin(e)
                at["pos"], at["quat"] = _fmt(pos0 + R0 @ p), _fmt(_qmul(quat0, q))
                if not collide:
                    at["contype"], at["conaffinity"], at["group"] = "0", "0", "1"
                    mat = e.find("material")
                    col = mat.find("color") if mat is not None else e.find("color")
                    if col is not None and col.get("rgba"):
                        at["rgba"] = col.get("rgba")
                out.append(ET.Element("geom", at))
        return out

    # -- bodies
    def _static(self, link: _Link) -> bool:
        j = link.joint
        return j is None or j.get("type", "fixed") == "fixed"

    def _emit(self, link: _Link, owner: ET.Element, acc_inertials: Optional[list], frame=None):
        """Adds `link` under the body element `owner`.  `frame` = pose in `owner` of the (possibly fused) URDF parent
        link.  A static link is fused into `owner` when fusestatic is on: its geoms and inertia move to the owner
        (`acc_inertials` collects what the owner will carry; None for the world, which has no inertia)."""
        p, q = _origin(link.joint)
        if frame is not None:
            p, q = frame[0] + _qmat(frame[1]) @ p, _qmul(frame[1], q)
        if self._static(link) and self.fusestatic:
            for g in self._geoms(link.elem, p, q):
                owner.append(g)
            if acc_inertials is not None:
                acc_inertials.append((p, q, _inertial(link.elem)))
            for c in link.children:
                self._emit(c, owner, acc_inertials, (p, q))
            return
        body = ET.SubElement(owner, "body", dict(name=link.name, pos=_fmt(p), quat=_fmt(q)))
        inertials = [(np.zeros(3), np.array([1.0, 0, 0, 0]), _inertial(link.elem))]
        if not self._static(link):
            for je in self._joints(link.joint):
                body.append(je)
        for g in self._geoms(link.elem, np.zeros(3), np.array([1.0, 0, 0, 0])):
            body.append(g)
        for c in link.children:
            self._emit(c, body, inertials)
        I = _merge_inertials(inertials)
        if I is not None:
            ie = ET.Element("inertial", dict(
                pos=_fmt(I["pos"]), quat=_fmt(I["quat"]), mass=_fmt(I["mass"]),
                fullinertia=_fmt([I["I"][0, 0], I["I"][1, 1], I["I"][2, 2], I["I"][0, 1], I["I"][0, 2], I["I"][1, 2]])))
            body.insert(0, ie)

    def _joints(self, j: ET.Element) -> List[ET.Element]:
        t, name = j.get("type"), j.get("name")
        if t == "floating":
            return [ET.Element("freejoint", dict(name=name))]
        ax = j.find("axis")
        axis = _f(ax.get("xyz"), 3) if ax is not None else np.array([1.0, 0, 0])
        at = dict(name=name, pos="0 0 0", axis=_fmt(axis))
        lim, dyn = j.find("limit"), j.find("dynamics")
        if t in ("revolute", "continuous"):
            at["type"] = "hinge"
        elif t == "prismatic":
            at["type"] = "slide"
        elif t == "planar":
            raise NotImplementedError("URDF planar joints")
        else:
            raise ValueError(f"unknown URDF joint type '{t}'")
        if t in ("revolute", "prismatic") and lim is not None and lim.get("lower") is not None and lim.get("upper") is not None:
            lo, hi = float(lim.get("lower")), float(lim.get("upper"))
            at["range"] = _fmt([lo, hi])
            at["limited"] = "true" if lo < hi else "false"
        else:
            at["limited"] = "false"
        if dyn is not None:
            if dyn.get("damping") is not None:
                at["damping"] = dyn.get("damping")
            if dyn.get("friction") is not None and float(dyn.get("friction")) != 0.0:
                at["frictionloss"] = dyn.get("friction")
        return [ET.Element("joint", at)]

    # -- whole model
    def convert(self) -> ET.Element:
        links = {e.get("name"): _Link(e) for e in self.root.findall("link")}
        for j in self.root.findall("joint"):
            p, c = j.find("parent").get("link"), j.find("child").get("link")
            if p not in links or c not in links:
                raise ValueError(f"joint '{j.get('name')}' references an unknown link")
            if links[c].joint is not None:
                raise ValueError(f"link '{c}' has two parent joints")
            links[c].joint = j
            links[p].children.append(links[c])
        roots = [l for l in links.values() if l.joint is None]
        if not roots:
            raise ValueError("URDF has no root link")
        mj = ET.Element("mujoco", dict(model=self.root.get("name", "robot")))
        comp = dict(angle="radian", autolimits="true")
        for k in ("meshdir", "balanceinertia", "boundmass", "boundinertia", "inertiafromgeom"):
            if k in self.comp:
                comp[k] = self.comp[k]
        ET.SubElement(mj, "compiler", comp)
        asset = ET.SubElement(mj, "asset")
        world = ET.SubElement(mj, "worldbody")
        for r in roots:
            self._emit(r, world, None)
        for m in self.mesh_elems:
            asset.append(m)
        if len(asset) == 0:
            mj.remove(asset)
        return mj


def urdf_to_mjcf(path) -> ET.Element:
    """MJCF tree of a URDF file, as MuJoCo's importer would build it (no actuators, no equalities: the reference adds
    those afterwards, see ambersim_amd/utils/io_utils.py)."""
    path = Path(path)
    return _Converter(parse_urdf(path), path.parent).convert()
