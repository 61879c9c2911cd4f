"""ba::InterpolationBufferT (include/ba/InterpolationBuffer.h): host-only
C++ test compiled with g++ — the worked example SURVEY.md §8c records for the reference's header
(10 samples, GetRange(0.15, 0.55) -> 6 elements, 1.5 @ 0.15 .. 5.5 @ 0.55), clamping, GetNext."""
import os
import subprocess


def test_interpolation_buffer_host(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "interp_buffer_test"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpp", "interp_buffer_test.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
