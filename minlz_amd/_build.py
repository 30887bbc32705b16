"""Builds minlz_amd/libminlz_hip.so for gfx950 with hipcc (in-tree; the .so travels to the GPU box)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mlz_hip.hip")
SO = os.path.join(HERE, "libminlz_hip.so")
CSRC = os.path.join(HERE, "csrc")


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    # every file under csrc/ is part of the one translation unit (mlz_hip.hip includes the rest)
    paths = [os.path.join(CSRC, d) for d in sorted(os.listdir(CSRC))] + [os.path.join(os.path.dirname(HERE), "include", "minlz_hip.h")]
    return any(os.path.getmtime(p) > t for p in paths)


def build(force=False, verbose=False):
    if not force and not stale():
        return SO
    # unaligned-ds-access: the ROCm runtime runs compute queues with unaligned LDS access enabled (verified on
    # MI355X, tools/ldsalign_test); telling the compiler lets unaligned 4/8-byte LDS copies be single DS ops.
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "+unaligned-ds-access",
           "-Wno-unused-command-line-argument", "-o", SO, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
