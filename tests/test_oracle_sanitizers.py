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


HIPCC = "/opt/rocm/bin/hipcc"
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


def _host_only_build(tmp_path, san_flags, driver_defs=()):
    """Every .hip source compiled host-only with `san_flags`, linked with the sweep driver against empty device images -> the executable."""
    csrc = os.path.join(ROOT, "nndetection_amd", "csrc")
    srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    flags = ["--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-result", "-ffp-contract=off"] + list(san_flags)
    procs = [(f, subprocess.Popen([HIPCC] + flags + ["-c", os.path.join(csrc, f), "-o", str(tmp_path / (f[:-4] + ".o"))],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)) for f in srcs]
    for f, p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, (f, err[-2000:])
    objs = [str(tmp_path / (f[:-4] + ".o")) for f in srcs]
    # the host objects refer to their device images (registered at start-up): empty ones will do, nothing is launched
    und = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout
    syms = sorted({w for line in und.splitlines() for w in line.split() if w.startswith("__hip_fatbin_")})
    stubs = tmp_path / "fatbin_stubs.c"
    stubs.write_text("".join('__attribute__((section(".hip_fatbin"), aligned(4096))) const char %s[4096] = {0};\n' % s for s in syms))
    assert subprocess.run(["gcc", "-c", str(stubs), "-o", str(tmp_path / "fatbin_stubs.o")]).returncode == 0
    drv = tmp_path / "driver.o"
    b = subprocess.run([CLANGXX, "-O1", "-g", "-std=c++17"] + list(san_flags) + list(driver_defs) + ["-c",
                        os.path.join(ROOT, "tests", "csrc", "host_sanitizer_driver.cpp"), "-o", str(drv)], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    exe = tmp_path / "host_san"
    link = subprocess.run([CLANGXX] + [f for f in san_flags if f.startswith("-fsanitize=")] + [str(drv), str(tmp_path / "fatbin_stubs.o")] + objs +
                          ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert link.returncode == 0, link.stderr[-3000:]
    return exe


# (the GPUs of the box, if any, are hidden from these executables: the empty device images must never be loaded)
_HIDE = dict(HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")


@pytest.mark.skipif(not (os.path.isfile(HIPCC) and os.path.isfile(CLANGXX)), reason="no ROCm toolchain")
def test_library_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    """Every source of the library compiled HOST-ONLY (`hipcc --cuda-host-only`: no device code, no GPU) with AddressSanitizer + UBSan, linked
    against empty device images, and the entry points that answer from the descriptor alone -- plan construction (tap / class / tile tables,
    magic multipliers), split-K / weight-gradient / sampler / NMS / ATSS / post-processing workspace sizes, kernel-coverage queries, argument
    validation of the compute entry points -- swept over 6 000 valid and invalid convolution problems (tests/csrc/host_sanitizer_driver.cpp).
    Round 6 found one kind of finding this way (pointer arithmetic on a null workspace base in five size queries)."""
    exe = _host_only_build(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", **_HIDE)      # (leaks: the HIP runtime's own start-up allocations)
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0 and run.stdout.startswith("ok 6000 problems"), (run.stdout[-500:], run.stderr[-3000:])
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]


@pytest.mark.skipif(not (os.path.isfile(HIPCC) and os.path.isfile(CLANGXX)), reason="no ROCm toolchain")
def test_library_host_side_is_clean_under_thread_sanitizer(tmp_path):
    """The same sweep from FOUR threads at once under ThreadSanitizer: the host side keeps per-process state (switches read once, per-device
    attribute flags) and is called from the autograd engine's thread as well as from the caller's."""
    exe = _host_only_build(tmp_path, ["-fsanitize=thread"], ["-DSWEEP_THREADS=4"])
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, **_HIDE), timeout=900)
    assert run.returncode == 0 and run.stdout.count("ok 6000 problems") == 4, (run.stdout[-500:], run.stderr[-3000:])
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
