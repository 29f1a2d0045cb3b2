"""The CPU tests of the restatement against its AddressSanitizer / UBSan build (`make -C oracle liboracle_asan.so`: -O1 -g, assertions -- the bound checks of
mat.hpp -- ON).  Test infrastructure checking test infrastructure; a few minutes.
    LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1 python oracle/run_sanitized.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"]).returncode == 0
import helpers  # noqa: E402
import pytest  # noqa: E402

helpers.ORACLE_PATH_OVERRIDE = os.path.join(ROOT, "oracle", "liboracle_asan.so")
files = ["test_oracle_fixed_base", "test_oracle_ocp", "test_oracle_unocp", "test_oracle_hybrid", "test_oracle_parnmpc", "test_oracle_unparnmpc", "test_golden_kkt",
         "test_oracle_task_space", "test_model_lie_host", "test_golden_rbd", "test_golden_riccati"]
sys.exit(pytest.main(["-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] + [os.path.join(ROOT, "tests", f + ".py") for f in files]))
