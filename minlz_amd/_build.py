"""Builds minlz_amd/libminlz_hip.so for gfx950 with hipcc (in-tree; the .so travels to the GPU box)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mlz_hip.hip")
SO = os.path.join(HERE, "libminlz_hip.so")
DEPS = ["mlz_hip.hip", "mlz_format.h", "mlz_kernels.h", "mlz_encode.hip.inc", "mlz_encode2.hip.inc", "mlz_decode.hip.inc", "mlz_decode_exec.hip.inc", "mlz_decode_general.hip.inc", "mlz_decode_serial.hip.inc", "mlz_crc.hip.inc", "mlz_stream.hip.inc"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    paths = [os.path.join(HERE, "csrc", d) for d in DEPS] + [os.path.join(os.path.dirname(HERE), "include", "minlz_hip.h")]
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
