"""The host threads of armon_hip_mgpu_cycle (armon.jl_amd/csrc/tile_pool.hpp: one thread per local tile, barriers between the
steps of the protocol) are plain C++: exercised here on the CPU under ThreadSanitizer — ordering across the barriers with
nothing but the barrier as synchronisation, failure propagation (the first failing tile's status and message, no later step
on any tile), reuse over thousands of runs, teardown. The GPU tests run the same pool under the real exchange."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_tile_pool_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "tile_pool_test")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer",
                            os.path.join(ROOT, "tests", "native", "tile_pool_test.cpp"), "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no -fsanitize=" + sanitizer)
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    run = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0 and "tile_pool OK" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
