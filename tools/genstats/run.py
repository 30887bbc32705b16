"""Words still open after each phase of the general-block decoder (CPU simulation, tools/genstats/genstats.cpp) on one 8 MiB block of
the enwik-like stream encoded by the oracle at level 1 and 2.  usage: python tools/genstats/run.py"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import oracle as O
from minlz_amd import synth
from minlz_amd.stream import uvarint
exe = os.path.join(here, "genstats")
subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(here, "genstats.cpp")])
d = synth.enwik_like(8 << 20, 1)
for lvl in (1, 2):
    e = O.encode(d, lvl)
    _, hl = uvarint(e, 1)
    open("/tmp/genstats_b%d.bin" % lvl, "wb").write(e[1 + hl:])
    print("oracle level", lvl)
    print(subprocess.run([exe, "/tmp/genstats_b%d.bin" % lvl, str(d.size)], capture_output=True, text=True).stdout)
