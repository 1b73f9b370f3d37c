"""Checks on the BUILT device code (no GPU needed: the library is disassembled with llvm-objdump)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "obia_amd", "csrc", "libobia_hip.so")


@pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="library not built / no llvm-objdump")
def test_no_64_bit_shift_by_the_last_allocated_vgpr():
    """tools/check_shift64.py: the erratum that made `slic_prep_lane_kernel` read sum_x as 1.0 in a few lanes per thousand (round 3)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_shift64.py"), LIB], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " functions, " in r.stdout and "0 functions" not in r.stdout, r.stdout   # (the scan did see the kernels)


def test_the_check_recognises_the_pattern():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_shift64 as c
    assert c.SHIFT.search("\tv_lshlrev_b64 v[38:39], v47, v[38:39]").group(2) == "47"
    assert c.SHIFT.search("\tv_lshlrev_b64 v[0:1], 6, v[12:13]") is None          # an immediate amount is not the pattern
    assert c.SHIFT.search("\tv_ashrrev_i64 v[2:3], v7, v[4:5]").group(1) == "v_ashrrev_i64"


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "libobia_oracle.so")) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"),
                    reason="oracle not built / no llvm-objdump")
def test_the_check_fails_when_it_scanned_nothing():
    """a file without device code (here: the host-only oracle library) must not pass as "0 bad shifts" """
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_shift64.py"), os.path.join(ROOT, "oracle", "libobia_oracle.so")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "EMPTY" in r.stdout, r.stdout + r.stderr
