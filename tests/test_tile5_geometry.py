"""CPU: index arithmetic of the streaming evaluation kernel k_eval5 (csrc/va_tile5.h) -- strips, staged images with
cyclic ghost columns, stencil neighbours, gather senders, packed store lanes -- checked for every even state width
from 66 to 1024 by a g++ build of tests/cpu_emul/tile5_check.cpp (the same header the kernel includes)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_even_width(tmp_path):
    exe = str(tmp_path / "tile5_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "varanneal_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpu_emul", "tile5_check.cpp")])
    widths = [str(d) for d in range(66, 1026, 2)]
    out = subprocess.run([exe] + widths, capture_output=True, text=True)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0, lines[-3:]
    assert len(lines) == len(widths) and all(ln.startswith("OK ") for ln in lines), [ln for ln in lines if not ln.startswith("OK ")][:5]
    c4 = [ln for ln in lines if ln.startswith("OK 200 ")][0]
    assert "NS=4" in c4 and "CW=56" in c4 and "PR=32" in c4 and "WPG=4 NSG=1" in c4      # BASELINE config 4: 48 + 48 + 48 + 56 columns


def test_narrow_and_odd_widths_are_refused(tmp_path):
    exe = str(tmp_path / "tile5_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "varanneal_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpu_emul", "tile5_check.cpp")])
    out = subprocess.run([exe, "20", "64", "67", "201"], capture_output=True, text=True)
    assert out.stdout.split() == ["NOTOK", "20", "NOTOK", "64", "NOTOK", "67", "NOTOK", "201"]
