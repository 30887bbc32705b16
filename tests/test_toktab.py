"""The table form of the token decode (minlz_amd/csrc/mlz_toktab.h: what the index and exec passes of the decoder use) against the
field-by-field form decode_tok, on the host: tools/tok_fields_check.cpp compiled with g++ (no GPU), every 41st value of a token's first
two bytes x 65540 tails (random + the extreme ones).  The full sweep (stride 1, 4.3e9 tokens) takes a minute: run the tool by hand."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table_decode_matches_field_decode(tmp_path):
    exe = tmp_path / "tfc"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "tok_fields_check.cpp")], check=True)
    r = subprocess.run([str(exe), "41"], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout
