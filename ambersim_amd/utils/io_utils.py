"""Model loading API (counterpart of reference ambersim/utils/io_utils.py:139-248)."""
from pathlib import Path
from typing import Optional, Tuple, Union

from ambersim_amd import mjx
from ambersim_amd.mjcf import SOL_CG, SOL_NEWTON, MjModel
from ambersim_amd.utils._internal_utils import _check_filepath


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
    The URDF branch and force_float are not built yet and raise
    NotImplementedError.
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
    if ext == "urdf":
        raise NotImplementedError("URDF loading is not built yet (SURVEY.md §8f rank 2); convert to MJCF first")
    if ext != "xml":
        raise NotImplementedError
    if force_float:
        raise NotImplementedError("force_float is not built yet (SURVEY.md §8f rank 2)")
    mj_model = MjModel.from_xml_path(filepath)

    mj_model.opt.solver = solver
    if iterations is not None:
        mj_model.opt.iterations = iterations
    if ls_iterations is not None:
        mj_model.opt.ls_iterations = ls_iterations
    return mj_model


def mj_to_mjx_model_and_data(mj_model: MjModel, device=None) -> Tuple[mjx.Model, mjx.Data]:
    """Host model -> (device Model, zero Data) (reference ambersim/utils/io_utils.py:222-241)."""
    mjx_model = mjx.put_model(mj_model, device)
    mjx_data = mjx.make_data(mjx_model)
    return mjx_model, mjx_data


def load_mjx_model_and_data_from_file(filepath: Union[str, Path], force_float: bool = False, device=None) -> Tuple[mjx.Model, mjx.Data]:
    """Convenience wrapper (reference ambersim/utils/io_utils.py:244-248)."""
    return mj_to_mjx_model_and_data(load_mj_model_from_file(filepath, force_float=force_float), device)
