"""The hand-scheduled kernels read LDS operands ahead of a counted wait; a compiler-inserted copy of such a
register before the wait would read a value that has not landed.  Compile the kernels to ISA and check."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", ["pwdw.hip", "pwdw_ws.hip", "pwdw_ws_gate.hip", "gemm_nn_res.hip"])
def test_no_read_of_inflight_lds_result(src, tmp_path):
    import check_lds_inflight
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "superresolution_amd", "csrc", src), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    assert "v_mfma" in text and "s_waitcnt lgkmcnt" in text
    assert check_lds_inflight.check(text) == 0


def test_checker_flags_a_premature_copy():
    import check_lds_inflight
    bad = "\n_Zkernel:\n\tds_read_b32 v1, v2\n\tv_mov_b32_e32 v3, v1\n\ts_waitcnt lgkmcnt(0)\n\t.amdhsa_kernel x\n"
    good = "\n_Zkernel:\n\tds_read_b32 v1, v2\n\tds_read_b32 v4, v2\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32_e32 v3, v1\n\ts_waitcnt lgkmcnt(0)\n\t.amdhsa_kernel x\n"
    assert check_lds_inflight.check(bad) == 1
    assert check_lds_inflight.check(good) == 0
