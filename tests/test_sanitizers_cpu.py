"""CPU-side sanitizer runs (GPU AddressSanitizer is not available on the pool): the host planner of the SpMM and the
C oracle, compiled with -fsanitize=address,undefined and driven over adversarial inputs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_spmm_planner_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "plan_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "sanitize_plan_main.cpp"),
           os.path.join(ROOT, "recommendation_amd", "csrc", "gcr_plan.cpp"), "-o", exe]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                         env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert run.returncode == 0 and "cases OK" in run.stdout, run.stdout
