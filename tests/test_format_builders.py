"""The header-word token builders the HIP encoder uses (minlz_amd/csrc/mlz_format.h) are plain
host+device C++: compile them for the host and compare with the oracle emitters (which are pinned
on the reference's TestEmitLiteral / TestEmitCopy tables) over a grid of offsets/lengths/literal counts."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_builders_match_oracle_emitters(tmp_path):
    exe = str(tmp_path / "fmt_check")
    obj = str(tmp_path / "oracle.o")
    subprocess.check_call(["gcc", "-O1", "-c", "-D_POSIX_C_SOURCE=200809L", os.path.join(ROOT, "oracle", "minlz_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "minlz_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cpp", "format_builders_check.cpp"), obj, "-lpthread", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 bad" in out.stdout
