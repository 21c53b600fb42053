"""The C part of the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU ASAN / XNACK runs are not available on the
pool, so the sanitizers cover what runs on the host -- the checker the NMS parity tests and bench.py's cpu_baseline leg rely on)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_oracle_nms_c_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "nms_ref_san"
    # (-ffp-contract=off as in oracle/Makefile; float division by zero is the reference's behaviour for zero-volume pairs, not a finding)
    cmd = ["gcc", "-O1", "-g", "-ffp-contract=off", "-fno-fast-math", "-fsanitize=address,undefined", "-fno-sanitize=float-divide-by-zero",
           "-fno-sanitize-recover=all", os.path.join(ROOT, "oracle", "nms_ref.c"), os.path.join(ROOT, "tests", "csrc", "nms_ref_sanitizer_driver.c"),
           "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "asan" in (build.stderr or "").lower():
        pytest.skip("this toolchain has no sanitizer runtime: " + build.stderr.splitlines()[-1])
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.startswith("ok ") and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr
