from pathlib import Path
from typing import Union

from ambersim_amd import ROOT


def _check_filepath(filepath: Union[str, Path]) -> str:
    """Resolves a model path: as given (absolute or cwd-relative) first, then relative to the package
    ROOT; ValueError when neither exists (semantics of reference ambersim/utils/_internal_utils.py:7-19)."""
    assert isinstance(filepath, (str, Path))
    path = Path(filepath)
    if not path.exists():
        path = ROOT / filepath
        if not path.exists():
            raise ValueError("The model file doesn't exist at the specified path!")
    return str(path)
