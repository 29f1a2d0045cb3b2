"""idocp::DiscreteEvent / idocp::ContactSequence of the facade (include/idocp/hybrid/*.hpp): tests/cpp/hybrid_host.cpp compiled with g++ and run on the host --
the rules of the reference's contact_sequence.hxx / discrete_event.hxx on a scripted gait, and the reference's way of failing (message on stderr, EXIT_FAILURE)."""
import os
import subprocess

import pytest

from helpers import ANYMAL_URDF, ROOT
from idocp_amd import capi

EXE = os.path.join(ROOT, "tests", "cpp", "_build", "hybrid_host")


@pytest.fixture(scope="module")
def exe():
    capi.lib()                                             # (the product library is built)
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "idocp_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Wshadow", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "hybrid_host.cpp"),
           "-L" + libdir, "-lidocp_hip", "-Wl,-rpath," + libdir, "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return EXE


def test_scripted_gait(exe):
    r = subprocess.run([exe, ANYMAL_URDF], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("violation,message", [
    ("no_event", "discrete_event.existDiscreteEvent() must be true!"),
    ("time_order", "must be larger than the last event time="),
    ("inconsistent", "discrete_event.preContactStatus() is not consistent with the last contact status!"),
    ("update_past_next", "lift_time=0.600000 must be smaller than event_time_[event_index+1]=0.500000!"),
    ("bad_phase", "contact_phase=6 must be smaller than numContactPhases()6!"),
    ("too_many", "Number of discrete events=2 exceeds predefined max_num_events=1!"),
    ("max_num_events", "invalid argument: max_num_events must be positive!"),
])
def test_violations_fail_the_reference_way(exe, violation, message):
    r = subprocess.run([exe, ANYMAL_URDF, violation], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and message in r.stderr, (r.returncode, r.stdout, r.stderr)
