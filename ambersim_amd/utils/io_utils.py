"""Model loading API (counterpart of reference ambersim/utils/io_utils.py:18-248)."""
import xml.etree.ElementTree as ET
from pathlib import Path
from typing import Optional, Tuple, Union

from ambersim_amd import mjx
from ambersim_amd.mjcf import SOL_CG, SOL_NEWTON, MjModel, compile_mjcf_tree, load_mjcf_tree
from ambersim_amd.urdf import parse_urdf, urdf_to_mjcf
from ambersim_amd.utils._internal_utils import _check_filepath


def _add_actuators(urdf_tree: ET.Element, xml_tree: ET.Element) -> None:
    """One `<motor name="<joint>_actuator">` per URDF `<transmission>`; `ctrlrange = +-effort` of the joint's `<limit>`
    when it has one, unlimited otherwise; joints that already have a motor are skipped (reference
    ambersim/utils/io_utils.py:18-70, which patches the saved XML file; here the tree is patched in memory)."""
    actuators = xml_tree.find("actuator")
    if actuators is None:
        actuators = ET.SubElement(xml_tree, "actuator")
    joints = {j.get("name"): j for j in urdf_tree.findall("joint")}
    for transmission in urdf_tree.findall("transmission"):
        joint_name = transmission.find("joint").get("name")
        if any(m.get("joint") == joint_name for m in actuators.findall("motor")):
            continue
        limit = joints[joint_name].find("limit")
        effort = limit.get("effort") if limit is not None else None
        if effort is not None:
            ET.SubElement(actuators, "motor", dict(name=joint_name + "_actuator", ctrllimited="true", ctrlrange=f"-{effort} {effort}", joint=joint_name))
        else:
            ET.SubElement(actuators, "motor", dict(name=joint_name + "_actuator", ctrllimited="false", joint=joint_name))


def _add_mimics(urdf_tree: ET.Element, xml_tree: ET.Element) -> None:
    """One joint equality `<joint1>_<joint2>_equality` with `polycoef = offset multiplier 0 0 0` per URDF mimic joint
    (reference ambersim/utils/io_utils.py:73-117)."""
    equality = xml_tree.find("equality")
    if equality is None:
        equality = ET.SubElement(xml_tree, "equality")
    for joint in urdf_tree.findall("joint"):
        mimic = joint.find("mimic")
        if mimic is None:
            continue
        joint1, joint2 = joint.get("name"), mimic.get("joint")
        multiplier = mimic.get("multiplier") if mimic.get("multiplier") is not None else 1
        offset = mimic.get("offset") if mimic.get("offset") is not None else 0
        ET.SubElement(equality, "joint", dict(name=f"{joint1}_{joint2}_equality", joint1=joint1, joint2=joint2, polycoef=f"{offset} {multiplier} 0 0 0"))


def _modify_robot_float_base(model: Union[str, Path, ET.Element], model_dir: Optional[Union[str, Path]] = None) -> MjModel:
    """Gives the robot a floating base if it lacks one: the first body under the worldbody gets a `<freejoint
    name="freejoint">` when it has no joint of its own (reference ambersim/utils/io_utils.py:120-136).  Accepts an MJCF
    path (as the reference does) or an already loaded tree plus its directory."""
    if isinstance(model, ET.Element):
        tree, base = model, Path(model_dir or ".")
    else:
        assert str(model).split(".")[-1] == "xml"
        tree, base = load_mjcf_tree(model), Path(model).parent
    worldbody = tree.find("worldbody")
    assert worldbody is not None
    first = worldbody.find("body")
    assert first is not None
    if first.find("joint") is None and first.find("freejoint") is None:
        first.insert(0, ET.Element("freejoint", dict(name="freejoint")))
        assert first.find("inertial") is not None or first.find("geom") is not None  # a free body needs mass
    return compile_mjcf_tree(tree, base)


def load_mj_model_from_file(
    filepath: Union[str, Path],
    force_float: bool = False,
    solver: Optional[Union[str, int]] = None,
    iterations: Optional[int] = None,
    ls_iterations: Optional[int] = None,
) -> MjModel:
    """Loads a host model from an MJCF path (absolute, cwd-relative or ROOT-relative).

    Same arguments and error behaviour as the reference (ambersim/utils/io_utils.py:139-219): the
    solver defaults to Newton (the reference's choice for mujoco>=3.0.1, :173-174); a bad solver
    string raises ValueError (:181); an unknown extension raises NotImplementedError (:199).
    URDF files go through ambersim_amd.urdf (MuJoCo's import rules) and get one motor per transmission and one
    joint equality per mimic joint (:190-196); `force_float` adds a free joint to the first body (:202-204).
    """
    if solver is None:
        solver = SOL_NEWTON
    elif isinstance(solver, str):
        if solver.lower() == "newton":
            solver = SOL_NEWTON
        elif solver.lower() == "cg":
            solver = SOL_CG
        else:
            raise ValueError("Solver must be one of: ['cg', 'newton']!")
    else:
        assert solver in (SOL_CG, SOL_NEWTON)

    filepath = _check_filepath(filepath)
    ext = str(filepath).split(".")[-1]
    base = Path(filepath).parent
    if ext == "urdf":
        # the reference saves MuJoCo's conversion next to the URDF, patches it and reloads it (:190-196); same steps
        # on the in-memory tree
        urdf_tree = parse_urdf(filepath)
        tree = urdf_to_mjcf(filepath)
        _add_actuators(urdf_tree, tree)
        _add_mimics(urdf_tree, tree)
    elif ext == "xml":
        tree = load_mjcf_tree(filepath)
    else:
        raise NotImplementedError
    if force_float:
        mj_model = _modify_robot_float_base(tree, base)
    else:
        mj_model = compile_mjcf_tree(tree, base)

    mj_model.opt.solver = solver
    if iterations is not None:
        mj_model.opt.iterations = iterations
    if ls_iterations is not None:
        mj_model.opt.ls_iterations = ls_iterations
    return mj_model


def mj_to_mjx_model_and_data(mj_model: MjModel, device=None) -> Tuple[mjx.Model, mjx.Data]:
    """Host model -> (device Model, zero Data) (reference ambersim/utils/io_utils.py:222-241)."""
    try:
        mjx_model = mjx.put_model(mj_model, device)
        mjx_data = mjx.make_data(mjx_model)
        return mjx_model, mjx_data
    except NotImplementedError as e:
        # same behaviour as the reference (:228-241): print what usually lies behind an unsupported-feature error, re-raise it.
        # The causes are this engine's own (its table of collision pairs and constraint rows, include/amx.h), not MJX's.
        print(UNSUPPORTED_HINT)
        raise e


UNSUPPORTED_HINT = """
        If you are seeing this, the model asks for something the MI355X engine (ambersim_amd) does not implement:

        (1) a collision pair outside its table: plane / sphere / capsule primitives against each other, a plane, sphere or
            convex hull (box, mesh) against a convex hull of at most 32 vertices, 32 faces and 64 edges; at most 16 contact
            slots per environment.  Restrict contype / conaffinity, simplify meshes (maxhullvert) or disable contact
            (DisableBit.CONTACT), as the reference's own Barrett-hand fixture does.

        (2) an elliptic instead of a pyramidal friction cone, or condim other than 3.

        (3) a model beyond the lane-role capacities (nv <= 32, 64 bodies / joints / pairs, 64 equality + limit rows), a non-Euler
            integrator, the PGS solver, or an equality / actuator type the engine does not carry (see DESIGN.md, capacity limits).
"""


def load_mjx_model_and_data_from_file(filepath: Union[str, Path], force_float: bool = False, device=None) -> Tuple[mjx.Model, mjx.Data]:
    """Convenience wrapper (reference ambersim/utils/io_utils.py:244-248)."""
    return mj_to_mjx_model_and_data(load_mj_model_from_file(filepath, force_float=force_float), device)
