#!/usr/bin/env python3
"""Writes tests/golden/interp_buffer.json: outputs of the REFERENCE's own
include/ba/InterpolationBuffer.h (it compiles stand-alone) for the query list
of tests/cpp/interp_buffer_dump.cpp.  Runs in the build container only (it needs
/root/reference); the JSON it writes is data — inputs and the reference's outputs — and is what
tests/test_interpolation_buffer.py checks this repo's header against.

    python tests/golden/make_interp_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_INCLUDE = "/root/reference/include"


def main():
    if not os.path.exists(os.path.join(REF_INCLUDE, "ba", "InterpolationBuffer.h")):
        sys.exit("reference header not found under %s (run this in the build container)" % REF_INCLUDE)
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "dump_ref")
        # no -ffast-math / -march flags: plain IEEE double arithmetic, as the test build uses
        subprocess.run(["g++", "-O1", "-std=c++11", "-I" + REF_INCLUDE,
                        os.path.join(ROOT, "tests", "cpp", "interp_buffer_dump.cpp"), "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    data = json.loads(out)  # validates
    data["_generator"] = ("tests/golden/make_interp_golden.py: tests/cpp/interp_buffer_dump.cpp compiled against "
                          "/root/reference/include/ba/InterpolationBuffer.h")
    with open(os.path.join(HERE, "interp_buffer.json"), "w") as f:
        json.dump(data, f, indent=0, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "interp_buffer.json"), {k: len(v["get_range"]) for k, v in data.items() if k != "_generator"})


if __name__ == "__main__":
    main()
