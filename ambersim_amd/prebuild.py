"""Bakes the per-model kernels of one or more models ahead of time:

    python -m ambersim_amd.prebuild models/biped/biped_stand.xml my_robot.xml [--iterations 5 --ls-iterations 10] [--cache-dir DIR]

`mjx.put_model` compiles csrc/ for a model's launch plan with hipcc the first time it sees the plan (about ten seconds; the generic kernels
run at well under half the per-model throughput - bench.py, "C3 generic kernels").  A serving host without ROCm's compiler cannot do that:
run this on a machine that has hipcc, ship the printed files (a library and a one-line compiler note each) in the package's csrc/jit/ - or
in the directory AMX_CACHE_DIR names on both sides - and `put_model` on the host finds them by the same key (launch-plan record + kernel
sources + flags; the solver's iteration caps are part of the plan, hence the options).  Loads models exactly as
`ambersim_amd.utils.io_utils.load_mj_model_from_file` does (reference ambersim/utils/io_utils.py:139-219: MJCF or URDF)."""
from __future__ import annotations

import argparse
import os
import sys


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m ambersim_amd.prebuild", description=__doc__.split("\n\n")[0])
    ap.add_argument("models", nargs="+", help="MJCF / URDF files (paths as load_mj_model_from_file takes them)")
    ap.add_argument("--iterations", type=int, default=None, help="solver iteration cap to bake (default: the file's)")
    ap.add_argument("--ls-iterations", type=int, default=None, help="line-search iteration cap to bake (default: the file's)")
    ap.add_argument("--cache-dir", default=None, help="where the libraries go (default: AMX_CACHE_DIR or the package's csrc/jit/)")
    args = ap.parse_args(argv)
    if args.cache_dir:
        os.environ["AMX_CACHE_DIR"] = args.cache_dir
    from . import _spec
    from .mjcf import recompile_options
    from .utils.io_utils import load_mj_model_from_file
    rc = 0
    for path in args.models:
        try:
            mj = load_mj_model_from_file(path)
            if args.iterations is not None or args.ls_iterations is not None:
                import copy
                mj = copy.copy(mj)
                mj.opt = mj.opt.replace(**{k: v for k, v in (("iterations", args.iterations), ("ls_iterations", args.ls_iterations)) if v is not None})
                mj = recompile_options(mj)
            out = _spec.build_specialised_library(mj)
            print(f"{path}: {out} (+ {_spec._compiler_note(out).name})")
        except Exception as e:  # one bad model does not stop the others; the exit code says so
            print(f"{path}: FAILED - {type(e).__name__}: {str(e)[:400]}", file=sys.stderr)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
