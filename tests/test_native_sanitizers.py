"""The host-side C++ of the library (lattice generation, penalisation, boundary index: plain C++17 with std::thread) under
AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer (race detection / sanitizers
run on the CPU build - GPU sanitizers are not available on this pool).  The driver tests/native/hostgen_sanitize.cpp also
checks the invariants of what it generates.  CPU only, a few seconds per build."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "native", "hostgen_sanitize.cpp"),
       os.path.join(ROOT, "pylatticedso_amd", "csrc", "pl_hostgen.cpp")]


@pytest.mark.parametrize("name,flags", [("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
                                        ("tsan", ["-fsanitize=thread"])])
def test_host_generator_under_sanitizers(tmp_path, name, flags):
    exe = str(tmp_path / f"hostgen_{name}")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-ffp-contract=off", *flags, *SRC, "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and ("cannot find" in build.stderr or "unrecognized" in build.stderr):
        pytest.skip(f"this toolchain has no {name} runtime: {build.stderr[-200:]}")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, PL_HOST_THREADS="8", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
               TSAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    assert run.stdout.strip().endswith("OK")
    assert "ERROR: AddressSanitizer" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr
    assert "runtime error" not in run.stderr
