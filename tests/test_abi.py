"""The C-ABI shared library loads and exports every symbol include/minlz_hip.h declares; the
host-only entry points behave like the reference's size helpers.  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from minlz_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _build.build()
    return _lib.lib()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "minlz_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mlz_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), "libminlz_hip.so does not export %s" % s
    assert sorted(_lib.SYMBOLS) == syms


def test_max_encoded_len(lib):
    # TestMaxEncodedLen, minlz_test.go:42-69
    assert lib.mlz_max_encoded_len(0) == 1
    assert lib.mlz_max_encoded_len(32) == 34
    assert lib.mlz_max_encoded_len(8 << 20) == (8 << 20) + 2
    assert lib.mlz_max_encoded_len((8 << 20) + 1) == -1
    assert lib.mlz_max_encoded_len(0xffffffff) == -1
    for n in (1, 15, 16, 4095, 4096, 65536, 1 << 20):
        assert lib.mlz_max_encoded_len(n) == n + 2


def _dl(lib, b):
    a = np.frombuffer(b, dtype=np.uint8)
    return lib.mlz_decoded_len(a.ctypes.data if a.size else None, a.size)


def test_decoded_len_matches_reference_rules(lib, twain_mzb):
    assert _dl(lib, twain_mzb) == 14168
    assert _dl(lib, b"\x00") == 0
    assert _dl(lib, b"") == -1                       # ErrCorrupt
    assert _dl(lib, b"\x00\x05") == -1               # header only
    assert _dl(lib, b"\x00\x00abc") == 3             # v == 0: rest are literals
    assert _dl(lib, b"\x00\x02\x08abc") == -1        # decoded smaller than body
    assert _dl(lib, b"\x00\x81\x80\x80\x04\x00") == -2  # > 8 MiB: ErrTooLarge
    assert _dl(lib, b"\x00\xff\xff\xff\xff\xff\xff\xff\xff\xff\x7f") == -1  # varint overflow
    assert _dl(lib, b"\x03abc") == 3                 # Snappy block: size still reported (decode.go:132-136)


def test_version_and_timer_names(lib):
    assert lib.mlz_version() >= 1
    names = [lib.mlz_timer_name(i).decode() for i in range(9)]
    assert "enc_tiles" in names and "dec_exec" in names


def test_init_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.mlz_init(0, C.byref(h)) < 0 and not h.value
    import minlz_amd as mz
    with pytest.raises(mz.ErrHIP):
        mz.Context(0)


def test_stream_decoded_len_is_host_only():
    # mlz_stream_decoded_len walks the chunks on the host (no device call): usable without a GPU
    import oracle as O
    from minlz_amd import synth
    L = _lib.lib()
    d = synth.text_like(3_000_000, 4).tobytes()
    st = O.stream_encode(d, 1, 1 << 20, add_index=True)
    import numpy as np
    a = np.frombuffer(st, dtype=np.uint8)
    assert L.mlz_stream_decoded_len(a.ctypes.data, a.size) == len(d)
    assert L.mlz_stream_decoded_len(a.ctypes.data, a.size // 2) == -1          # ErrCorrupt: truncated
    assert L.mlz_stream_bound(len(d), 1 << 20, 1) >= len(st)
    assert L.mlz_stream_bound(len(d), 1000, 0) < 0


def test_bench_refuses_a_world_it_cannot_start():
    """`python bench.py --gpus N` without a launcher starts the N ranks itself and must fail loudly — not run a smaller world
    under the same label — when the box has fewer than N GPUs (there is none here)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # whatever the host has, the subprocess sees no GPU at all: the refusal is what is under test, never a real 2-rank run
    env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = env["ROCR_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "GPU" in r.stderr
    assert "{" not in r.stdout          # no JSON line from a run that did not happen


def _build_c_client(tmp_path):
    """tests/c/abi_client.c: the header compiled as C99 by gcc (the compiler cgo runs on go/minlz_hip.go's preamble), linked against the library."""
    import shutil
    import subprocess
    _build.build()
    exe = str(tmp_path / "abi_client")
    so_dir = os.path.dirname(_lib.SO)
    cmd = [shutil.which("gcc") or "gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_client.c"), "-o", exe, "-L", so_dir, "-lminlz_hip", "-Wl,-rpath," + so_dir]
    subprocess.check_call(cmd)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return exe, env


def test_c_client_compiles_as_c99_and_runs_host_checks(tmp_path):
    """No GPU here: the client's host-only checks pass and mlz_init's failure is the exit code 77 (on a GPU box it goes on and returns 0)."""
    import subprocess
    exe, env = _build_c_client(tmp_path)
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode in (0, 77), (p.returncode, p.stderr)


@pytest.mark.gpu
def test_c_client_on_the_device(tmp_path):
    """The whole C client on the GPU: Encode/Decode, block contracts, batches, streams on one device and on two contexts behind one handle."""
    import subprocess
    exe, env = _build_c_client(tmp_path)
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "abi_client ok" in p.stdout, (p.returncode, p.stdout, p.stderr)
