"""Model conversion utilities (counterpart of reference ambersim/utils/conversion_utils.py:12-38; the CoACD convex
decomposition helpers of that file are offline tooling outside the hot path and are not built)."""
import xml.etree.ElementTree as ET
from pathlib import Path
from typing import Optional, Union

from ambersim_amd.mjcf import compile_mjcf_tree, load_mjcf_tree
from ambersim_amd.utils._internal_utils import _check_filepath


def save_model_xml(filepath: Union[str, Path], output_path: Optional[Union[str, Path]] = None) -> None:
    """Loads a URDF or MJCF model and saves it as one MJCF file (reference: `mj_saveLastXML`, conversion_utils.py:12-38).

    Without `output_path` the file lands in the current directory under the model's stem.  Asset directories are
    rewritten as absolute paths when the output is not written next to the source, so the saved file still loads.
    """
    try:
        filepath = _check_filepath(filepath)
        tree = load_mjcf_tree(filepath)
        src_dir = Path(filepath).parent.resolve()
        compile_mjcf_tree(tree, src_dir)  # same failure modes as loading the model (bad mesh paths, ...)
        if output_path is None:
            output_path = str(filepath).split("/")[-1].split(".")[0] + ".xml"
        if Path(output_path).resolve().parent != src_dir:
            comp = tree.find("compiler")
            if comp is None:
                comp = ET.Element("compiler")
                tree.insert(0, comp)
            for key in ("meshdir", "assetdir"):
                if key in comp.attrib or (key == "meshdir" and tree.find("asset/mesh") is not None):
                    comp.set(key, str((src_dir / comp.get(key, "")).resolve()))
        ET.indent(tree)
        ET.ElementTree(tree).write(str(output_path))
        print(f"XML file saved to {str(output_path)}!")
    except ValueError as e:
        print(e)
        print("If you're getting errors about mesh filepaths, make sure to add a mujoco tag to the URDF to specify the meshdir!")
