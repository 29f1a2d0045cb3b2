"""The drop-in claim of INTEGRATION.md section 1, taken literally: the REFERENCE's own example drivers -- `examples/iiwa14/*.cpp`, `examples/anymal/*.cpp`,
twelve programs -- are compiled UNMODIFIED, from where they lie under the reference tree, against this repository's `include/` and `libidocp_hip.so`
(`examples/Makefile`, target `ref`; outputs in `examples/_ref/`, git-ignored; nothing of the reference is copied), and run on the GPU.

* without a GPU (here, where the reference tree is present): all twelve compile and link;
* on the GPU box (the binaries travel with the tree like every built file; the reference does not exist there): all twelve run to completion, print
  finite KKT errors, and print THE SAME errors as this repository's own drivers of the same workloads (`examples/*.cpp`, whose output
  `tests/test_examples_gpu.py` and `tests/test_fixed_base_ocp_gpu.py` hold to the oracle) -- so the mirrors are faithful and the facade's classes
  behave under the reference's call sequences, not only under ours."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("IDOCP_REFERENCE_DIR", "/root/reference")
REF_BIN = os.path.join(ROOT, "examples", "_ref")
# reference driver -> this repository's driver of the same workload (None: no twin with the same call sequence)
TWINS = {
    "iiwa14_unocp_benchmark": "iiwa14_unocp_benchmark",
    "iiwa14_unparnmpc_benchmark": "iiwa14_unparnmpc_benchmark",
    "iiwa14_config_space_ocp": "iiwa14_config_space_ocp",
    "iiwa14_task_space_ocp": "iiwa14_task_space_ocp",
    "iiwa14_ocp_benchmark": None,             # ours initialises the constraints, the reference's driver never does
    "iiwa14_parnmpc_benchmark": None,
    "anymal_ocp_benchmark": "anymal_ocp_benchmark",
    "anymal_parnmpc_benchmark": "anymal_parnmpc_benchmark",
    "anymal_anymal_trotting": "anymal_trotting",
    "anymal_anymal_running": "anymal_running",
    "anymal_anymal_jumping": "anymal_jumping",
    "anymal_anymal_trotting_parnmpc": "anymal_trotting_parnmpc",
}


def kkt_lines(stdout):
    return [l for l in stdout.splitlines() if "KKT error" in l]


def kkt_values(stdout):
    return [float(x) for x in re.findall(r"KKT error[^=]*= ([-+0-9.eE]+|nan|-nan|inf)", stdout)]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples", "iiwa14")), reason="the reference tree is not on this machine")
def test_the_reference_example_drivers_compile_and_link_unmodified():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "ref", "REF=" + REF], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    built = sorted(os.listdir(REF_BIN))
    assert built == sorted(TWINS), built
    # nothing of the reference sits in the tree: the outputs are binaries, and git does not see them
    assert all(open(os.path.join(REF_BIN, b), "rb").read(4) == b"\x7fELF" for b in built)
    assert subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "examples/_ref"], capture_output=True, text=True).stdout == ""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples", "iiwa14")) or shutil.which("cmake") is None, reason="needs the reference tree and cmake")
def test_the_reference_cmake_projects_configure_and_build_against_the_package_config(tmp_path):
    """README "Usage" of the reference: find_package(idocp REQUIRED) + idocp::idocp + ${IDOCP_INCLUDE_DIR}.  cmake/idocpConfig.cmake provides that package
    for this tree; the reference's examples/iiwa14 and examples/anymal PROJECTS -- their own CMakeLists.txt, C++11 requested -- configure and build with
    it, nothing edited (out-of-source: nothing is written under the reference tree)."""
    for robot, n in (("iiwa14", 6), ("anymal", 6)):
        build = tmp_path / robot
        r = subprocess.run(["cmake", "-S", os.path.join(REF, "examples", robot), "-B", str(build), "-Didocp_DIR=" + os.path.join(ROOT, "cmake"),
                            "-DCMAKE_BUILD_TYPE=Release", "-DCMAKE_POLICY_VERSION_MINIMUM=3.5"],      # (the projects ask for CMake 3.1, which current CMake only accepts with this)
                           capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        r = subprocess.run(["cmake", "--build", str(build), "-j", "4"], capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        exes = [f for f in os.listdir(build) if os.path.isfile(build / f) and os.access(build / f, os.X_OK) and open(build / f, "rb").read(4) == b"\x7fELF"]
        assert len(exes) == n, exes
        # linked against THIS library
        ldd = subprocess.run(["ldd", str(build / exes[0])], capture_output=True, text=True).stdout
        assert os.path.join(ROOT, "idocp_amd", "lib", "libidocp_hip.so") in ldd, ldd


@pytest.mark.gpu
def test_the_reference_example_drivers_run_on_the_gpu_and_print_what_our_drivers_print(tmp_path):
    if not os.path.isdir(REF_BIN) or sorted(os.listdir(REF_BIN)) != sorted(TWINS):
        pytest.skip("examples/_ref is not built (it is built where the reference tree is: __graft_entry__.build())")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the directory the reference's drivers expect: they open ../<robot description>/urdf/<robot>.urdf from their build directory
    run = tmp_path / "build"
    run.mkdir()
    for d, f in (("iiwa_description", "iiwa14.urdf"), ("anymal_b_simple_description", "anymal.urdf")):
        os.makedirs(tmp_path / d / "urdf")
        shutil.copy(os.path.join(ROOT, "tests", "golden", "urdf", f), tmp_path / d / "urdf" / f)
    for name, twin in TWINS.items():
        r = subprocess.run([os.path.join(REF_BIN, name)], cwd=run, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, (r.stdout + r.stderr)[-2000:])
        vals = kkt_values(r.stdout)
        assert len(vals) >= 11 and np.isfinite(vals).all(), (name, vals[:5])
        if name != "anymal_anymal_trotting_parnmpc":      # (full steps of ParNMPC from that cold start do not contract: tests/test_oracle_parnmpc.py)
            assert vals[-1] < 0.5 * vals[0], (name, vals[0], vals[-1])
        if twin is None:
            continue
        urdf = os.path.join(ROOT, "tests", "golden", "urdf", "iiwa14.urdf" if twin.startswith("iiwa14") else "anymal.urdf")
        t = subprocess.run([os.path.join(ROOT, "examples", twin), urdf], cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert t.returncode == 0, (twin, (t.stdout + t.stderr)[-2000:])
        a, b = kkt_lines(r.stdout), kkt_lines(t.stdout)
        n = min(len(a), len(b), 30)
        assert n >= 11 and a[:n] == b[:n], (name, [x for x in zip(a[:n], b[:n]) if x[0] != x[1]][:3])
