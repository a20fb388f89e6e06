"""Compile-time specialisation records of launch plans (csrc/amx_specs_gen.h) and per-model builds of the library.

A record holds every integer of a model's launch plan (the AMX_MODEL_INTS list of csrc/amx_dev.h) and its dof tree; the
kernels instantiated on it have static loop bounds, immediate LDS offsets and tree-sparse eliminations.  The shipped models'
records are compiled into libamx.so (tools/gen_specs.py).  `build_specialised_library(mj)` does the same for ANY model at
load time: it writes the model's record to a cache directory, compiles csrc/ against it with hipcc (its own kernels only: about ten seconds, once per
plan) and returns the path of that library - `mjx.put_model(mj, specialise=True)` loads it instead of the generic kernels."""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path

_CSRC = Path(__file__).resolve().parent / "csrc"


_PLAN_LIB = None


def _plan_library():
    """csrc/libamx_plan.so (host C++, no HIP), loaded once per process.  build() makes it; if it is missing (a source checkout
    that has not been built) it is compiled here into the cache directory under the build lock - never `make` in the package
    directory on every call: that fails on a read-only install and races between ranks."""
    global _PLAN_LIB
    if _PLAN_LIB is not None:
        return _PLAN_LIB
    path = _CSRC / "libamx_plan.so"
    srcs = [_CSRC / f for f in ("amx_plan.cpp", "amx_plan.h", "amx_dev.h")] + [_CSRC.parent.parent / "include" / "amx.h"]
    stale = (not path.exists()) or any(f.stat().st_mtime > path.stat().st_mtime for f in srcs)
    if stale:
        h = hashlib.sha256()
        for f in srcs:
            h.update(f.read_bytes())
        path = cache_dir() / f"libamx_plan_{h.hexdigest()[:16]}.so"
        if not path.exists():
            with _build_lock():
                if not path.exists():
                    tmp = _unique_tmp(path)
                    cxx = os.environ.get("CXX", "g++")
                    r = subprocess.run([cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-o", str(tmp), str(_CSRC / "amx_plan.cpp")],
                                       capture_output=True, text=True, cwd=str(_CSRC))
                    if r.returncode != 0:
                        raise RuntimeError(f"planner build failed:\n{r.stderr[-2000:]}")
                    os.replace(tmp, path)
    L = C.CDLL(str(path))
    L.amx_model_plan_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.amx_plan_last_error.restype = C.c_char_p
    _PLAN_LIB = L
    return L


def plan(mj, force_lanes: int = 0) -> dict:
    """The launch plan of a host model as {name: text}, from the planner built alone (csrc/libamx_plan.so: host C++, no HIP).
    force_lanes = 64: the model's WIDE plan (one environment per wavefront), which amx_model_create keeps beside the regular one
    of a 32-lane model for batches too small to fill the SIMDs."""
    from ._desc import build_desc
    L = _plan_library()
    d = build_desc(mj)
    buf = C.create_string_buffer(1 << 16)
    L.amx_plan_force_lanes(int(force_lanes))
    try:
        rc = L.amx_model_plan_describe(C.byref(d), buf, len(buf))
    finally:
        L.amx_plan_force_lanes(0)
    if rc != 0:
        raise RuntimeError(L.amx_plan_last_error().decode())
    return dict(line.split("=") for line in buf.value.decode().splitlines())


def render_record(name: str, p: dict) -> str:
    p = dict(p)
    parents = [int(x) for x in p.pop("dof_parent").split(",")]
    tree_h = int(p.pop("tree_H"))
    p2_col = [int(x) for x in p.pop("p2_col").split(",")]
    p.pop("mrow", None)  # (tooling: the compact-M row table as text; the kernels read it from the constants)
    pad = parents + [-1] * (32 - len(parents))
    return "\n".join([
        f"struct Spec_{name} {{",
        "  static constexpr bool is_static = true;",
        f'  static constexpr const char *name = "{name}";',
        "  " + " ".join(f"static constexpr int {k} = {v};" for k, v in p.items()),
        f"  static constexpr bool tree_H = {'true' if tree_h else 'false'};  // J'DJ keeps the dof tree's sparsity",
        f"  static constexpr int dof_parent[32] = {{{', '.join(str(x) for x in pad)}}};",
        f"  static constexpr int p2_col[32] = {{{', '.join(str(x) for x in p2_col)}}};  // mfma2: column of register k of pair k / 16's block (dof, nv = gradient, -1 none)",
        "};"])


def render_header(models, banner: str) -> str:
    lines = [banner, "// One record per model: every integer of its launch plan (AMX_MODEL_INTS) and its dof tree, as compile-time constants.",
             "#pragma once", ""]
    names = []
    for entry in models:  # (name, model) or (name, model, forced lane-group width)
        name, mj = entry[0], entry[1]
        lines.append(render_record(name, plan(mj, entry[2] if len(entry) > 2 else 0)))
        names.append(name)
    lines += ["", "#define AMX_SPECS(X) " + " ".join(f"X(Spec_{n})" for n in names)]
    return "\n".join(lines) + "\n"


def cache_dir() -> Path:
    """AMX_CACHE_DIR, or csrc/jit/ inside the package (built libraries are git-ignored like libamx.so itself)."""
    d = Path(os.environ.get("AMX_CACHE_DIR", _CSRC / "jit"))
    d.mkdir(parents=True, exist_ok=True)
    return d


def _unique_tmp(out: Path) -> Path:
    """A temporary name next to `out` that no other process uses: pid + random suffix (ranks of one job share the cache)."""
    import uuid
    return out.with_name(f"{out.name}.{os.getpid()}.{uuid.uuid4().hex[:8]}.tmp")


class _build_lock:
    """Exclusive advisory lock on <cache>/.build.lock: one compile per cache directory at a time.  A rank that finds the
    library already there when it gets the lock does nothing (torchrun / bench.py --gpus N with AMX_SPECIALISE=1: every rank
    asks for the same build at once)."""

    def __enter__(self):
        import fcntl
        self.f = open(cache_dir() / ".build.lock", "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


ARCH = os.environ.get("AMX_ARCH", "gfx950")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
         "-Wno-unused-function", "-Wno-unused-variable"]
_HIPCC_ID = {}


def _hipcc_id(hipcc: str) -> str:
    """`hipcc --version`, cached: a compiler upgrade must not reuse the old compiler's libraries."""
    if hipcc not in _HIPCC_ID:
        try:
            _HIPCC_ID[hipcc] = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
        except Exception as e:
            raise RuntimeError(f"{hipcc} is not runnable ({e}): a per-model build needs ROCm's hipcc on this machine")
    return _HIPCC_ID[hipcc]


def _extra_flags():
    """AMX_SPEC_FORCE_FALLBACK=1: build with the crash-fallback flags from the start (tests/test_gpu_parity.py runs that build
    against the oracle, so the instruction mix a compiler crash would silently switch to is a tested one)."""
    flags = list(CRASH_FALLBACK_FLAGS) if os.environ.get("AMX_SPEC_FORCE_FALLBACK", "0") not in ("", "0") else []
    # (r04's AMX_ACCURATE_ELIM=1 - true pivot rows through the crossbar - is retired: since r05 the default eliminations are bitwise
    #  symmetric and ARE the true-row eliminations, csrc/amx_kernels.hip sym_pivot; -DAMX_TRUE_ROW_ELIM=1 remains as an A/B switch of tools/devbuild.sh)
    return flags


_SRC_DIGEST = {}


def _sources_digest() -> bytes:
    """sha256 over the kernel sources, once per process and (mtime, size) state: Model.replace(opt=...) asks for a library path on every
    call, and re-reading 300 KB of sources each time was the cost of a cached build."""
    srcs = [_CSRC / f for f in ("amx_api.cpp", "amx_kernels.hip", "amx_plan.h", "amx_dev.h")] + [_CSRC.parent.parent / "include" / "amx.h"]
    key = tuple((str(f), f.stat().st_mtime_ns, f.stat().st_size) for f in srcs)
    if key not in _SRC_DIGEST:
        h = hashlib.sha256()
        for f in srcs:
            h.update(f.read_bytes())
        _SRC_DIGEST.clear()
        _SRC_DIGEST[key] = h.digest()
    return _SRC_DIGEST[key]


def specialised_library_path(mj):
    """(path, header text) of the per-model build of `mj`: keyed by the record, the sources, the compile flags, the
    architecture and the compiler's version string."""
    header = render_header([("model", mj)], "// generated by ambersim_amd._spec.build_specialised_library for one model; do not edit.")
    h = hashlib.sha256(header.encode())
    h.update(_sources_digest())
    h.update(" ".join(FLAGS + _extra_flags()).encode())
    # (r05: the compiler's version is no longer part of the name - a host without hipcc must be able to find a library that
    #  `python -m ambersim_amd.prebuild` made elsewhere; it is recorded beside the library and checked wherever hipcc is runnable)
    return cache_dir() / f"libamx_{h.hexdigest()[:16]}.so", header


def _compiler_note(out: Path) -> Path:
    return out.with_suffix(".hipcc.txt")


def _cached_build_usable(out: Path) -> bool:
    """The cached library exists and - where this machine has a runnable hipcc - was made by that very compiler (an upgrade must not
    reuse the old compiler's code); on a machine without hipcc a prebuilt library is taken as it is."""
    if not out.exists():
        return False
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        return True
    note = _compiler_note(out)
    try:
        return note.exists() and note.read_text() == _hipcc_id(hipcc)
    except RuntimeError:
        return True


# What a per-model build is retried with when the compiler itself crashes (see build_specialised_library): the packed-fp32 forms
# off.  tests/test_abi.py::test_compiler_crash_fallback_is_pinned holds the plan that needs it and checks that these two macros
# gate nothing but the packed spellings of the same sums.
CRASH_FALLBACK_FLAGS = ("-DAMX_PK_SV=0", "-DAMX_PK_DOT=0")
SCHED_FLAGS = ("-mllvm", "-amdgpu-sched-strategy=iterative-ilp")  # second retry: dropped (see build_specialised_library)
_LAST_BUILD = {"path": None, "fallback": False, "stage": None}  # what the last compile in this process did (tests, logs)


def build_specialised_library(mj, verbose: bool = False) -> Path:
    """Compiles csrc/ with a single specialisation record - this model's - and returns the library's path (cached; see
    specialised_library_path).  Safe when several processes ask for the same build at once: the compile runs under a file
    lock, writes to a name of its own and is renamed into place; a process that finds the library there uses it.
    Needs hipcc (ROCm) on the machine; raises RuntimeError otherwise."""
    out, header = specialised_library_path(mj)
    if _cached_build_usable(out):
        return out
    with _build_lock():
        if _cached_build_usable(out):  # another rank built it while this one waited
            return out
        tag = out.stem.split("_")[-1]
        hdr = cache_dir() / f"amx_spec_{tag}.h"
        tmp_h = _unique_tmp(hdr)
        tmp_h.write_text(header)
        os.replace(tmp_h, hdr)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = _unique_tmp(out)
        cmd = [hipcc] + _extra_flags() + FLAGS + [f'-DAMX_SPECS_HEADER="{hdr}"', "-DAMX_NO_GENERIC=1", "-no-hip-rt", "-shared", "-o", str(tmp), "-x", "hip", str(_CSRC / "amx_api.cpp")]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(_CSRC))
        _LAST_BUILD.update(path=str(out), fallback=False, stage=None)
        crashed = lambda rr: rr.returncode != 0 and "frontend command failed due to signal" in rr.stderr
        if crashed(r) and os.environ.get("AMX_SPEC_NO_RETRY", "0") in ("", "0"):
            # ROCm 7.2's greedy register allocator crashes (a segmentation fault inside VirtRegAuxInfo::isRematerializable) on some plans'
            # rollout kernels.  Two retries, each the same arithmetic on the same HIP path:
            #  1. the packed-fp32 spellings of the spatial sums off (r03: a 16-lane model with two free bodies compiled that way);
            #  2. the default machine scheduler instead of -amdgpu-sched-strategy=iterative-ilp (r05: a 32-lane CG model of the fuzz family,
            #     tests/fuzz_models.py seed 1005, crashes with and without the packed forms and compiles so; the kernels lose the
            #     latency-oriented schedule, ~10 % on the shipped models, and keep everything else).
            import logging
            log = logging.getLogger("ambersim_amd")
            first = r.stderr
            base = list(cmd)
            ladder = (("packed_off", base[:1] + list(CRASH_FALLBACK_FLAGS) + base[1:], "with " + " ".join(CRASH_FALLBACK_FLAGS)),
                      ("default_scheduler", [a for a in base if a not in SCHED_FLAGS], "without " + " ".join(SCHED_FLAGS)))
            for stage, cmd, what in ladder:
                _LAST_BUILD.update(fallback=True, stage=stage)
                log.warning("hipcc crashed on this model's kernels (%s); rebuilding %s", first.strip().splitlines()[-1][:120] if first.strip() else "", what)
                r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(_CSRC))
                if not crashed(r):
                    break
            if r.returncode != 0:
                r.stderr = first[-1000:] + "\n--- after the retries ---\n" + r.stderr
        if r.returncode != 0:
            try:
                tmp.unlink()
            except OSError:
                pass
            raise RuntimeError(f"specialised build failed ({' '.join(cmd)}):\n{r.stderr[-2000:]}")
        note = _unique_tmp(_compiler_note(out))
        note.write_text(_hipcc_id(hipcc))
        os.replace(note, _compiler_note(out))
        os.replace(tmp, out)
    if verbose:
        print(f"built {out}")
    return out
