"""The URDF reader of the library (idocp_amd/csrc/urdf_model.cpp; the reference gets its model from pinocchio / urdfdom) on damaged input: truncated files,
missing lines, corrupted numbers and attribute names.  It must answer with an error code and a message, or with a model -- never crash, hang or read out of
bounds (tests/run_host_sanitizers.sh runs this file against the ASan / UBSan build).  No GPU involved."""
import ctypes as C
import os
import re

import numpy as np

from helpers import ANYMAL_URDF, IIWA_URDF
from idocp_amd import capi


def try_model(path, frames):
    m = capi.Model()
    fr = (C.c_int * max(1, len(frames)))(*frames)
    return capi.lib().idocp_model_from_urdf(path.encode(), fr if frames else None, len(frames), C.byref(m)), m


def test_damaged_urdf_files_are_refused_or_read_never_crashed_on(tmp_path):
    lib = capi.lib()
    lib.idocp_model_from_urdf.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(capi.Model)]
    rng = np.random.default_rng(11)
    seen = {"ok": 0, "refused": 0}
    for urdf, frames in ((IIWA_URDF, []), (ANYMAL_URDF, [14, 24, 34, 44])):
        text = open(urdf).read()
        lines = text.splitlines(keepends=True)
        variants = []
        for _ in range(25):
            variants.append(text[:int(rng.integers(0, len(text)))])                                          # truncated anywhere
        for _ in range(25):
            drop = set(rng.integers(0, len(lines), size=int(rng.integers(1, 12))).tolist())                 # lines missing
            variants.append("".join(l for i, l in enumerate(lines) if i not in drop))
        nums = [m.span() for m in re.finditer(r"-?\d+\.\d+(e-?\d+)?", text)]
        for _ in range(25):
            a, b = nums[int(rng.integers(0, len(nums)))]
            junk = ["nan", "", "1e999", "abc", "--3", "0x10", "1 2 3 4 5 6 7 8 9"][int(rng.integers(0, 7))]   # a number that is not one
            variants.append(text[:a] + junk + text[b:])
        for _ in range(15):
            pos = int(rng.integers(0, len(text) - 8))
            variants.append(text[:pos] + "".join(chr(int(c)) for c in rng.integers(1, 127, size=8)) + text[pos + 8:])   # eight random bytes
        variants += ["", "<robot", "<robot name='x'/>", "<robot><link name='a'/></robot>", "<robot><joint name='j' type='revolute'/></robot>", "\x00" * 64]
        for k, v in enumerate(variants):
            p = tmp_path / ("v%d.urdf" % k)
            p.write_bytes(v.encode("latin-1", "replace"))
            rc, m = try_model(str(p), frames)
            if rc == 0:
                seen["ok"] += 1
                # a model came back: its dimensions are sane and every number in it is one
                assert 0 < m.njoints <= capi.MAX_JOINTS and 0 < m.nv <= capi.MAX_NV and m.nq in (m.nv, m.nv + 1) and m.nu <= m.nv
                nums = np.concatenate([np.ctypeslib.as_array(getattr(m, f)).ravel()[:n] for f, n in
                                       (("mass", m.njoints), ("com", 3 * m.njoints), ("inertia", 9 * m.njoints), ("axis", 3 * m.njoints),
                                        ("plc_R", 9 * m.njoints), ("plc_p", 3 * m.njoints))])
                assert np.isfinite(nums).all() and np.isfinite(m.total_mass), (k, "a non-finite number went through")
            else:
                seen["refused"] += 1
                assert len(lib.idocp_last_error()) > 0
    assert seen["refused"] > 50 and seen["ok"] >= 0, seen
    # a missing file and a contact frame that does not exist
    rc, _ = try_model(str(tmp_path / "absent.urdf"), [])
    assert rc != 0
    rc, _ = try_model(IIWA_URDF, [9999])
    assert rc != 0 and len(lib.idocp_last_error()) > 0
