"""C-ABI library: loads without a GPU, exports every symbol include/amx.h declares, struct layout matches ctypes."""
import ctypes as C
import os
import re
import subprocess

import pytest

from ambersim_amd import _lib
from ambersim_amd._desc import AmxModelDesc, build_desc

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "amx.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(amx_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert set(names) == set(_lib.EXPORTS) and len(names) == 13
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.amx_version() == 3


def test_desc_layout_matches_c(tmp_path):
    """sizeof/offsetof of amx_model_desc as gcc sees the header vs the ctypes mirror."""
    fields = ["version", "nq", "nconvexvert", "nhull", "nhulledge", "pair_hull1", "hull_vert", "hull_face_normal", "hull_edge", "timestep", "gravity", "meaninertia", "iterations", "disableflags",
              "body_parentid", "body_invweight0", "jnt_type", "qpos_spring", "dof_invweight0", "geom_size", "pair_geom1",
              "convex_vert", "actuator_trnid", "actuator_forcerange", "eq_obj1id", "eq_solimp"]
    src = tmp_path / "layout.c"
    lines = "".join(f'  printf("{f} %zu\\n", offsetof(amx_model_desc, {f}));\n' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "amx.h"\nint main(void){\n  printf("sizeof %zu\\n", sizeof(amx_model_desc));\n'
                   + lines + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof"]) == C.sizeof(AmxModelDesc)
    for f in fields:
        assert int(out[f]) == getattr(AmxModelDesc, f).offset, f


def test_env_tail_layout_matches_c(tmp_path):
    """amx_env_tail_desc / amx_env_tail_io (the fused env.step, amx_env_step) as gcc sees the header vs the ctypes mirrors."""
    src = tmp_path / "layout2.c"
    dfields = ["nobs", "obs_kind", "obs_index", "nrew", "rew_kind", "rew_index", "rew_target", "rew_weight", "episode_length"]
    lines = "".join(f'  printf("d.{f} %zu\\n", offsetof(amx_env_tail_desc, {f}));\n' for f in dfields)
    lines += "".join(f'  printf("io.{f} %zu\\n", offsetof(amx_env_tail_io, {f}));\n' for f in _lib.TAIL_IO_FIELDS)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "amx.h"\nint main(void){\n  printf("sizeof_d %zu\\n", sizeof(amx_env_tail_desc));\n'
                   '  printf("sizeof_io %zu\\n", sizeof(amx_env_tail_io));\n' + lines + "  return 0;\n}\n")
    exe = tmp_path / "layout2"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof_d"]) == C.sizeof(_lib.AmxEnvTailDesc) and int(out["sizeof_io"]) == C.sizeof(_lib.AmxEnvTailIO)
    for f in dfields:
        assert int(out["d." + f]) == getattr(_lib.AmxEnvTailDesc, f).offset, f
    for f in _lib.TAIL_IO_FIELDS:
        assert int(out["io." + f]) == getattr(_lib.AmxEnvTailIO, f).offset, f
    lib = _lib.lib()
    assert lib.amx_env_step(None, 0, 1, None, None, None, None, None, None, None, None) == _lib.AMX_EINVAL  # codes, not exceptions


def test_error_reporting_without_gpu(mj_models):
    """Bad arguments come back as codes + messages, never exceptions across the boundary; codes map to the
    exception types the reference's callers expect."""
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.amx_model_create(None, 0, C.byref(h)) == _lib.AMX_EINVAL
    assert b"null" in lib.amx_last_error()
    d = build_desc(mj_models["pendulum"])
    d.version = 99
    assert lib.amx_model_create(C.byref(d), 0, C.byref(h)) == _lib.AMX_EINVAL
    d = build_desc(mj_models["pendulum"])
    d.integrator = 1
    assert lib.amx_model_create(C.byref(d), 0, C.byref(h)) == _lib.AMX_EUNSUPPORTED
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.AMX_EUNSUPPORTED)
    with pytest.raises(ValueError):
        _lib.check(_lib.AMX_EINVAL)
    assert lib.amx_rollout(None, 1, 1, None, 0, None, None, None, None) == _lib.AMX_EINVAL
    assert lib.amx_quadcost(1, 1, 1, 1, None, None, None, None, None, None, None, None) == _lib.AMX_EINVAL
    assert lib.amx_fd_grad(None, 2, 1, None, None, None, 0, None, None, None, None, None, None, 0, None) == _lib.AMX_EINVAL
    assert lib.amx_fd_workspace_bytes(None, 2, 1, None) == _lib.AMX_EINVAL


def test_descriptor_indices_are_validated(mj_models):
    """amx_model_create is a public ABI: a descriptor with an out-of-range index is refused with AMX_EINVAL naming the
    field, before any of it is used as an index (no GPU needed: validation precedes the upload)."""
    import numpy as np
    lib = _lib.lib()
    h = C.c_void_p()

    def corrupt(field, idx, val):
        d = build_desc(mj_models["quadruped"])
        arr = np.ctypeslib.as_array(getattr(d, field), shape=(idx + 1,)).copy()
        arr[idx] = val
        d._keep = getattr(d, "_keep", []) + [arr]
        setattr(d, field, arr.ctypes.data_as(C.POINTER(C.c_int)))
        rc = lib.amx_model_create(C.byref(d), 0, C.byref(h))
        return rc, lib.amx_last_error().decode()

    for field, idx, val in [("body_parentid", 3, 7), ("jnt_qposadr", 2, 1000), ("jnt_dofadr", 1, -3), ("dof_bodyid", 4, 99),
                            ("dof_parentid", 5, 5), ("dof_jntid", 0, 40), ("geom_bodyid", 1, -2), ("pair_geom1", 0, 500),
                            ("jnt_type", 1, 9), ("jnt_bodyid", 1, 0)]:
        rc, msg = corrupt(field, idx, val)
        assert rc == _lib.AMX_EINVAL and field in msg, (field, rc, msg)


def test_capacity_limits_are_reported(mj_models):
    """Models beyond the lane-role capacities are rejected up front (AMX_ECAPACITY -> NotImplementedError)."""
    import numpy as np
    from ambersim_amd.mjcf import MjModel
    bodies = "".join(f'<body pos="0 {i} 0"><joint type="hinge" axis="0 0 1"/><geom type="sphere" size="0.1"/></body>' for i in range(40))
    big = MjModel.from_xml_string(f"<mujoco><worldbody>{bodies}</worldbody></mujoco>")
    assert big.nv == 40
    lib = _lib.lib()
    h = C.c_void_p()
    d = build_desc(big)
    assert lib.amx_model_create(C.byref(d), 0, C.byref(h)) == _lib.AMX_ECAPACITY
    assert b"nv=40" in lib.amx_last_error()


def test_generated_c_descriptor_is_current(tmp_path):
    """examples/c_host/pendulum_desc.h is what tools/dump_desc_c.py writes for the shipped pendulum (the C host uploads it)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "pendulum_desc.h"
    subprocess.run([sys.executable, os.path.join(root, "tools", "dump_desc_c.py"), "models/pendulum/pendulum.xml", str(out), "pendulum"], check=True,
                   capture_output=True, cwd=root)
    strip = lambda s: "\n".join(l for l in s.splitlines() if not l.startswith("/* Generated"))
    assert strip(out.read_text()) == strip(open(os.path.join(root, "examples", "c_host", "pendulum_desc.h")).read())


def test_generated_spec_header_is_current():
    """ambersim_amd/csrc/amx_specs_gen.h (compile-time records of the shipped models' launch plans) is what tools/gen_specs.py
    writes from the planner in this tree; the kernels specialised on it would otherwise not match their models."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_specs.py"), "--check"], cwd=root).returncode == 0


def test_spec_matching_compares_the_two_pair_columns(mj_models):
    """the two-pair Hessian of a per-model kernel adds each accumulator column to the register `Spec::p2_col` names at compile
    time, and those columns come from the pairs' dof lists - which no integer of the launch plan fixes.  amx_model_create must therefore
    compare the list itself: the shipped bipeds match their record; the same plan with ONE column of the list changed (every integer, the
    dof tree and the sparsity class still equal) matches nothing; a model without the two-pair form ignores the list."""
    import ctypes as C
    from ambersim_amd import _lib
    from ambersim_amd._desc import build_desc
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    L = _lib.lib()  # (loads the HIP runtime the library binds to first)
    L.amx_spec_index_of.restype = C.c_int
    for path in ("models/biped/biped.xml", "models/biped/biped_stand.xml"):
        d = build_desc(load_mj_model_from_file(path))
        idx = L.amx_spec_index_of(C.byref(d), -1, 0)
        assert idx >= 0, path
        assert L.amx_spec_index_of(C.byref(d), 7, 24) == -1, path  # pair 0's second leg column moved to the other leg's dof
        assert L.amx_spec_index_of(C.byref(d), 7, 19) == idx, path  # (poking the value it already has changes nothing)
    dq = build_desc(mj_models["quadruped"])
    iq = L.amx_spec_index_of(C.byref(dq), -1, 0)
    assert iq >= 0 and L.amx_spec_index_of(C.byref(dq), 7, 24) == iq  # four single-contact pairs: no two-pair form, the list is not consulted


def test_dense_plan_of_a_contact_rich_model(monkeypatch):
    """r05: a Newton model whose contact rows alone ask for 64 lanes - the falls-and-rests biped, 16 slots - is planned in
    the dense form when that keeps more environments in a CU: two contact rows per lane of a 32-lane group, compact M, three-row Jacobian
    blocks, the solver's row records in the place of the contact points.  Checked here on the host: the plan's numbers; the library ships
    kernels for it; the compact matrix holds every entry the dof tree makes non-zero exactly once per stored row and nothing overlaps;
    AMX_NO_DENSE=1 gives the regular plan back; the headline model's plan is untouched."""
    import ctypes as C
    from ambersim_amd import _lib, _spec
    from ambersim_amd._desc import build_desc
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    mj = load_mj_model_from_file("models/biped/biped_rest.xml")
    p = _spec.plan(mj)
    g = lambda k: int(p[k])
    assert (g("G"), g("nbs"), g("mcomp"), g("jfr"), g("fusedH"), g("mfma2")) == (32, 2, 1, 1, 1, 0)
    assert g("wpb") * (64 // g("G")) == 16 and (g("nconst") + 16 * g("env_floats")) * 4 <= 160 * 1024
    nv, ldm, nb0 = g("nv"), g("ldm"), g("mnb0")
    assert nb0 == 6 and ldm >= nv + 2  # the floating base; two spare columns
    parent = list(mj.dof_parentid)
    anc = lambda a, i: a == i or (i >= 0 and anc(a, parent[i]))
    rows = [tuple(int(x) for x in r.split(":")) for r in p["mrow"].split(",")]
    assert len(rows) == nv
    used = set()
    for i, (base, clo, chi) in enumerate(rows):
        cells = set(range(base, base + 4 * (chi - clo + 1)))
        assert not (cells & used) and base % 4 == 0, i
        used |= cells
        for j in range(nv):
            related = anc(j, i) or anc(i, j)
            stored = clo <= j // 4 <= chi
            if i < nb0:
                assert stored  # whole rows
            elif j >= nb0 and related:
                assert stored, (i, j)  # every non-zero entry beyond the leading chain is in the row's own chunks
    assert max(used) < g("e_J") - g("e_M") < nv * ldm  # inside its block, smaller than whole rows
    assert g("env_floats") % 64 == 32  # the two environments of a wavefront 32 banks apart (measured later: no effect either way - DESIGN.md 7)
    # the three-row Jacobian block and what follows it fit in front of the slab's end; the solver's records sit where the contact points were
    assert g("e_J") + 3 * g("ncon") * g("ldc") <= g("e_vch") and g("e_con") <= g("e_H") < g("e_row") < g("e_M")
    L = _lib.lib()
    L.amx_spec_index_of.restype = C.c_int
    d = build_desc(mj)
    assert L.amx_spec_index_of(C.byref(d), -1, 0) >= 0  # shipped kernels for the dense form
    monkeypatch.setenv("AMX_NO_DENSE", "1")
    q = _spec.plan(mj)
    assert (int(q["G"]), int(q["nbs"]), int(q["mcomp"]), int(q["jfr"])) == (64, 1, 0, 0) and int(q["wpb"]) * (64 // int(q["G"])) == 8
    monkeypatch.delenv("AMX_NO_DENSE")
    # a 27-dof draw of the contact-rich fuzz family: rows of seven chunks have one spare column, compact M wants two - the planner gives the
    # rows an eighth chunk (and the model sixteen environments per CU instead of fourteen)
    from fuzz_models import random_dense_model
    f = _spec.plan(random_dense_model(0))
    assert (int(f["nv"]), int(f["ldm"]), int(f["mcomp"]), int(f["jfr"]), int(f["wpb"]) * (64 // int(f["G"]))) == (27, 32, 1, 1, 16)
    monkeypatch.setenv("AMX_DENSE_NO_MCOMP", "1")
    f = _spec.plan(random_dense_model(0))
    assert (int(f["ldm"]), int(f["mcomp"]), int(f["jfr"]), int(f["wpb"]) * (64 // int(f["G"]))) == (28, 0, 1, 14)
    monkeypatch.delenv("AMX_DENSE_NO_MCOMP")
    h = _spec.plan(load_mj_model_from_file("models/biped/biped_stand.xml"))
    assert (int(h["G"]), int(h["nbs"]), int(h["mcomp"]), int(h["jfr"]), int(h["mrot"]), int(h["env_floats"])) == (32, 1, 0, 0, 1, 2348)


def jit_test_model():
    """A model the library ships no specialisation for: the quadruped with two solver iterations instead of three."""
    import copy
    from ambersim_amd.mjcf import recompile_options
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    mj = copy.copy(load_mj_model_from_file("models/quadruped/quadruped.xml"))
    mj.opt = mj.opt.replace(iterations=2)
    return recompile_options(mj)


def jit_test_model_equality():
    """A second one: the quadruped with its two front knees coupled by a joint equality - equality rows, so the Hessian is
    stored and solved densely (no fused path) in a 32-lane group, the one static-solve combination no shipped model has."""
    from ambersim_amd.mjcf import MjModel
    xml = open(os.path.join(ROOT, "ambersim_amd", "models", "quadruped", "quadruped.xml")).read()
    xml = xml.replace("<actuator>", '<equality><joint joint1="fl_knee" joint2="fr_knee" polycoef="0 1 0 0 0"/></equality>\n  <actuator>')
    return MjModel.from_xml_string(xml)


def test_per_model_specialised_build():
    """ambersim_amd._spec.build_specialised_library: the record of an arbitrary model is rendered and csrc/ compiles against it
    (hipcc cross-compiles without a GPU); the library exports the whole ABI.  The build is cached in csrc/jit/, where the GPU
    test of the same model finds it."""
    from ambersim_amd import _spec
    mj = jit_test_model()
    p = _spec.plan(mj)
    assert p["iterations"] == "2" and p["nv"] == "18" and p["G"] == "32" and p["fusedH"] == "1"
    rec = _spec.render_record("model", p)
    assert "static constexpr int iterations = 2;" in rec and "dof_parent[32]" in rec
    path = _spec.build_specialised_library(mj)
    assert path.exists() and path == _spec.build_specialised_library(mj)  # cached
    L = _lib.load(path)
    assert L.amx_version() == 3
    mj2 = jit_test_model_equality()
    p2 = _spec.plan(mj2)
    assert p2["neq"] == "1" and p2["G"] == "32" and p2["fusedH"] == "0"
    assert _spec.build_specialised_library(mj2).exists()


def test_a_plan_that_crashes_the_compiler_still_gets_its_kernels():
    """A REAL plan on which ROCm 7.2's register allocator segfaults (r05, found by tools/dbg/fuzz_spec.py: tests/fuzz_models.py seed 1005 - a
    15-dof CG model on 32-lane groups; the crash is in r04's tree too, with and without the packed-fp32 forms): the per-model build walks its
    retry ladder to the default machine scheduler and succeeds.  (A compiler that no longer crashes makes this a plain build.)  The GPU side
    runs the result against the oracle (test_gpu_parity.py::test_per_model_build_of_the_plan_that_crashes_the_compiler)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fuzz_models import random_model
    from ambersim_amd import _spec
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    mj = random_model(1005)
    p = _spec.plan(mj)
    assert (p["nv"], p["G"], p["solver"]) == ("15", "32", "1")
    path = _spec.build_specialised_library(mj)
    assert path.exists() and _lib.load(path).amx_version() == 3


def test_prebuild_cli_makes_what_a_host_without_hipcc_finds(tmp_path, monkeypatch):
    """`python -m ambersim_amd.prebuild model.xml --cache-dir D` (r05) bakes a model's per-model kernels where hipcc is installed; a host
    WITHOUT the compiler finds that library by the same key (record + sources + flags - the compiler's version is a note beside the
    library, checked only where hipcc can be asked) instead of falling back to the generic kernels.  Here: the CLI builds the quadruped
    at Newton 2 / 5 into a fresh directory and prints the path; with HIPCC pointing nowhere the cached build is still what
    build_specialised_library returns; with a compiler whose version differs from the note the cache is NOT reused."""
    import subprocess, sys
    from ambersim_amd import _spec
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "ambersim_amd.prebuild", "models/quadruped/quadruped.xml", "--iterations", "2", "--cache-dir", str(tmp_path)],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    built = [f for f in os.listdir(tmp_path) if f.startswith("libamx_") and f.endswith(".so")]
    assert len(built) == 1 and built[0] in r.stdout and os.path.exists(tmp_path / built[0].replace(".so", ".hipcc.txt"))
    monkeypatch.setenv("AMX_CACHE_DIR", str(tmp_path))
    mj = jit_test_model()  # the same plan: quadruped, two solver iterations
    assert _spec.specialised_library_path(mj)[0].name == built[0]
    monkeypatch.setenv("HIPCC", str(tmp_path / "no_such_hipcc"))
    assert _spec.build_specialised_library(mj).name == built[0]  # no compiler here: the prebuilt library is taken as it is
    fake = tmp_path / "other_hipcc"
    fake.write_text("#!/bin/sh\necho 'HIP version: 0.0.0-other'\nexit 1\n")
    fake.chmod(0o755)
    monkeypatch.setenv("HIPCC", str(fake))
    _spec._HIPCC_ID.clear()
    with pytest.raises(RuntimeError, match="specialised build failed"):  # another compiler: rebuild attempted (and this stand-in cannot build)
        _spec.build_specialised_library(mj)
    _spec._HIPCC_ID.clear()
    r = subprocess.run([sys.executable, "-m", "ambersim_amd.prebuild", "models/does_not_exist.xml", "--cache-dir", str(tmp_path)],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 1 and "FAILED" in r.stderr


def test_compiler_crash_fallback_is_pinned(tmp_path, monkeypatch):
    """A per-model build whose compile dies inside the compiler ("frontend command failed due to signal", seen once on ROCm 7.2's
    register allocator in r03; the source site that triggered it has since been rewritten, so no current plan reproduces it) is
    retried with _spec.CRASH_FALLBACK_FLAGS.  Pinned here with a stand-in compiler that fails exactly that way unless the
    fallback flags are present and otherwise IS hipcc: (a) the retry happens, once, and the two command lines differ by those
    flags only; (b) the two macros gate nothing but the packed-fp32 spellings of sums (every use in the kernel source is an
    `#if` around a `pk_` / f2v form with a scalar `#else`); (c) the resulting library is a complete ABI.  The GPU test
    test_crash_fallback_build_matches_the_oracle runs that instruction mix against the oracle."""
    import re
    import stat
    import sys
    from ambersim_amd import _spec
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    log = tmp_path / "calls.log"
    fake = tmp_path / "hipcc_that_crashes"
    fake.write_text(f"""#!{sys.executable}
import os, subprocess, sys
args = sys.argv[1:]
if "--version" in args:
    sys.exit(subprocess.run([{hipcc!r}] + args).returncode)
open({str(log)!r}, "a").write(" ".join(args) + chr(10))
if not all(f in args for f in {list(_spec.CRASH_FALLBACK_FLAGS)!r}):
    sys.stderr.write("clang++: error: clang frontend command failed due to signal (use -v to see invocation)" + chr(10))
    sys.exit(1)
sys.exit(subprocess.run([{hipcc!r}] + args).returncode)
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("HIPCC", str(fake))
    monkeypatch.setenv("AMX_CACHE_DIR", str(tmp_path / "cache"))
    monkeypatch.setattr(_spec, "_HIPCC_ID", {})
    path = _spec.build_specialised_library(jit_test_model())
    assert path.exists() and _spec._LAST_BUILD["fallback"] is True
    calls = [c.split() for c in log.read_text().splitlines()]
    assert len(calls) == 2 and _spec._LAST_BUILD["stage"] == "packed_off"
    assert [a for a in calls[1] if a not in calls[0]] == list(_spec.CRASH_FALLBACK_FLAGS) and [a for a in calls[0] if a not in calls[1]] == []
    assert _lib.load(path).amx_version() == 3
    # AMX_SPEC_NO_RETRY=1: the crash is reported instead (how a new crashing plan would be found and reduced)
    monkeypatch.setenv("AMX_SPEC_NO_RETRY", "1")
    monkeypatch.setenv("AMX_CACHE_DIR", str(tmp_path / "cache2"))
    with pytest.raises(RuntimeError, match="frontend command failed"):
        _spec.build_specialised_library(jit_test_model())
    # (b) what the two macros gate in the source
    src = open(os.path.join(ROOT, "ambersim_amd", "csrc", "amx_kernels.hip")).read()
    for macro in ("AMX_PK_SV", "AMX_PK_DOT"):
        uses = [m.start() for m in re.finditer(rf"#if {macro}\b", src)]
        other = [m.start() for m in re.finditer(rf"\b{macro}\b", src)]
        assert len(uses) >= 2 and len(other) == len(uses) + 2, (macro, len(uses), len(other))  # + the #ifndef and #define of its definition: no other use
        for u in uses:
            block = src[u:src.index("#endif", u)]
            packed_arm = block.split("#else")[0]
            assert "f2v" in packed_arm, (macro, packed_arm[:200])  # the guarded arm is a packed spelling ...
            if "#else" in block:
                assert "f2v" not in block.split("#else")[1], (macro, "... and the other arm is scalar")


def test_inline_assembly_dpp_reads_keep_their_wait_states():
    """The crossbar-free eliminations issue v_fmac_f32_dpp as inline assembly, which the compiler's hazard recogniser does not
    look into: tools/check_dpp_hazard.py re-checks on the emitted gfx950 code that no such instruction reads a register a
    VALU instruction wrote fewer than 2 wait states earlier."""
    hipcc = os.environ.get("AMX_HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "ambersim_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asm"], check=True, capture_output=True)
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "check_dpp_hazard.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert int(r.stdout.strip().splitlines()[-1].split()[0]) > 1000  # the check saw the eliminations


def test_hazard_checker_knows_both_rules(tmp_path):
    """The checker itself, on three hand-written snippets: a DPP read one wait state behind its producer is flagged, the same read behind an
    `s_nop 1` is not, and - the rule r05 added after an assembly block consumed a compiler-emitted v_rsq_f32's result with no wait state,
    silently corrupting every elimination - a transcendental's result read by the very next vector instruction is flagged too."""
    tool = os.path.join(ROOT, "tools", "check_dpp_hazard.py")
    def run(text):
        f = tmp_path / "k.s"
        f.write_text("_Zkernel:\n" + "".join("\t" + l + "\n" for l in text))
        r = subprocess.run(["python3", tool, str(f)], capture_output=True, text=True)
        return r.returncode, r.stdout
    dpp = "v_fmac_f32_dpp v1, v2, v3 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    rc, out = run(["v_mul_f32 v2, v4, v5", "v_add_f32 v9, v9, v9", dpp])
    assert rc == 1 and "HAZARD" in out
    rc, out = run(["v_mul_f32 v2, v4, v5", "s_nop 1", dpp])
    assert rc == 0, out
    rc, out = run(["v_rsq_f32_dpp v32, v30 row_newbcast:13 row_mask:0xf bank_mask:0xf bound_ctrl:1", "v_mul_f32 v68, v31, v32"])
    assert rc == 1 and "TRANS HAZARD" in out
    rc, out = run(["v_rsq_f32_e32 v32, v30", "s_lshl_b64 s[0:1], s[2:3], 5", "v_mul_f32 v68, v31, v32"])
    assert rc == 0, out
