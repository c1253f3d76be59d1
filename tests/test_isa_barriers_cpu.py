"""Build-time guard for the barrier structure of the pipelined two-product kernels (csrc/gcr_infonce.hip): the gfx950
assembly of every instantiation is scanned (scripts/exp/check_barrier_waits.py) for
  * an LDS instruction between the last `s_waitcnt lgkmcnt(0)` and an `s_barrier` (a ring slot re-used too early), and
  * in the 512-thread form, a barrier that sits inside an EXEC-masked (possibly divergent) region: the two wave groups' extra
    barriers are only sound under branches the compiler has proven uniform (s_cbranch_scc* / vcc*).
A compiler change that breaks either shows up here, in the CPU suite, instead of as a hang on the GPU box."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_pipe_kernel_barriers_are_waited_for_and_uniformly_guarded(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    asm = tmp_path / "gcr_infonce.s"
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "recommendation_amd", "csrc"), os.path.join(ROOT, "recommendation_amd", "csrc", "gcr_infonce.hip"),
           "-o", str(asm)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:]
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exp", "check_barrier_waits.py"), str(asm)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert chk.returncode == 0, chk.stdout
    assert "'barriers_in_8wave_branches'" in chk.stdout, chk.stdout      # the scan did find (and pass) the guarded barriers
