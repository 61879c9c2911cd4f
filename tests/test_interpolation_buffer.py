"""ba::InterpolationBufferT (include/ba/InterpolationBuffer.h), host only.

* differential test against the REFERENCE's own header: `tests/golden/interp_buffer.json` holds the
  outputs of /root/reference/include/ba/InterpolationBuffer.h (it compiles stand-alone) for the
  query list of `tests/cpp/interp_buffer_dump.cpp` — GetElement with its index, GetNext walks and
  GetRange on uniform, jittered, gapped and two-sample buffers, including query times that coincide
  with stored samples (generator: `tests/golden/make_interp_golden.py`, build container only).
  The same driver compiled against this repo's header must print the same numbers, bit for bit.
* the worked example SURVEY.md §8c records (10 samples, GetRange(0.15, 0.55) -> 6 elements).
"""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, src, name):
    exe = tmp_path / name
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", src), "-o", str(exe)], check=True)
    return exe


def test_interpolation_buffer_host(tmp_path):
    exe = _build(tmp_path, "interp_buffer_test.cpp", "interp_buffer_test")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_interpolation_buffer_matches_the_reference_header(tmp_path):
    exe = _build(tmp_path, "interp_buffer_dump.cpp", "interp_buffer_dump")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    ours = json.loads(r.stdout)
    with open(os.path.join(ROOT, "tests", "golden", "interp_buffer.json")) as f:
        ref = json.load(f)
    ref.pop("_generator")
    assert sorted(ours) == sorted(ref)
    nq = nr = 0
    for name in ref:
        a, b = ours[name], ref[name]
        for key in ("n", "start_time", "end_time", "average_dt"):
            assert a[key] == b[key], (name, key)
        for qa, qb in zip(a["get_element"], b["get_element"]):
            assert qa == qb, (name, "get_element", qb["t"], qa, qb)   # value, time AND index
            nq += 1
        for ra, rb in zip(a["get_range"], b["get_range"]):
            assert ra == rb, (name, "get_range", rb["start"], rb["end"])
            nr += 1
        for wa, wb in zip(a["get_next"], b["get_next"]):
            assert wa == wb, (name, "get_next", wb["from"], wb["max_time"])
        assert len(a["get_element"]) == len(b["get_element"]) and len(a["get_range"]) == len(b["get_range"])
    assert nq > 500 and nr > 300
