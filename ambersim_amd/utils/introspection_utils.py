"""Name lookups on a host model (counterpart of reference ambersim/utils/introspection_utils.py:8-25)."""
from typing import List

from ambersim_amd.mjcf import MjModel


def get_actuator_names(model: MjModel) -> List[str]:
    return list(model.names_actuator)


def get_equality_names(model: MjModel) -> List[str]:
    return list(model.names_eq)


def get_geom_names(model: MjModel) -> List[str]:
    return list(model.names_geom)


def get_joint_names(model: MjModel) -> List[str]:
    return list(model.names_joint)
