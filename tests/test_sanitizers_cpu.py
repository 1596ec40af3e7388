"""CPU sanitizer builds of the host sampler (SURVEY.md section 5 build stance; GPU sanitizers are not available on this
pool): tests/native/sampler_sanitize.cpp + pmgt_amd/csrc/sampler.cpp compiled with ThreadSanitizer and with
AddressSanitizer + UBSan, then run -- concurrent threaded batches on one handle, the sequential entry on a second handle
and handle creation / destruction on a third, all at once.  A sanitizer finding or a failed determinism check is a
non-zero exit."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_sampler_under_sanitizer(san, tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "sampler_san")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={san}", "-pthread", "-o", exe,
                            os.path.join(ROOT, "tests", "native", "sampler_sanitize.cpp"),
                            os.path.join(ROOT, "pmgt_amd", "csrc", "sampler.cpp")], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-4000:])
    assert "sanitize harness ok" in run.stdout
