"""GPU encoder (LevelFastest, match + serialize kernels) against the CPU model of the same algorithm
(tools/encmodel2): the block bodies must be byte-identical; every block is also decoded by the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "encmodel2"))
import ctypes as C
import run2
import minlz_amd as mz
from minlz_amd import synth
import oracle as O

def model_body(a, level=1, **kw):
    p = run2.P(**dict(run2.def_for(a.size, level), **kw))
    out = np.zeros(a.size + a.size // 8 + 64, dtype=np.uint8)
    n = run2.L.model2_block(a.ctypes.data, a.size, C.byref(p), out.ctypes.data, None)
    return out[:n].tobytes()

def main():
    ctx = mz.Context(0)
    rng = np.random.default_rng(5)
    cases = []
    for n in [0, 1, 15, 16, 17, 63, 64, 100, 1000, 4095, 8191, 8192, 8193, 8200, 20000, 32767, 32768, 32769, 40000, 65536, 100000, 1 << 20, (1 << 20) + 77]:
        cases.append(("text%d" % n, synth.text_like(max(n, 1), seed=7)[:n]))
    cases.append(("text8M", synth.text_like(8 << 20, seed=1)))
    cases.append(("json8M", synth.json_like(8 << 20, seed=2)))
    cases.append(("rand1M", rng.integers(0, 256, 1 << 20, dtype=np.uint8)))
    cases.append(("zeros1M", np.zeros(1 << 20, dtype=np.uint8)))
    cases.append(("zeros100k+", np.concatenate([np.zeros(100000, dtype=np.uint8), synth.text_like(50000, seed=3)])))
    mix = np.concatenate([synth.text_like(300000, seed=4), rng.integers(0, 256, 200000, dtype=np.uint8), synth.text_like(300000, seed=4)])
    cases.append(("mix", mix))
    pat = np.tile(np.frombuffer(b"abcdefghij", dtype=np.uint8), 30000)
    cases.append(("period10", pat))
    bad = 0
    for name, data, level in [(n_, d_, 1) for n_, d_ in cases] + [(n_ + "/L2", d_, 2) for n_, d_ in cases if len(d_) >= 8000]:
        a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
        enc = mz.Encode(a, mz.LevelFastest if level == 1 else mz.LevelBalanced, ctx)
        ok_rt = O.decode(enc) == a.tobytes()
        msg = ""
        if a.size >= 16 and not (len(enc) >= 2 and enc[0] == 0 and enc[1] == 0 and len(enc) == a.size + 2):
            # header: 00 uvarint(N)
            hl = 1
            while enc[hl] & 0x80: hl += 1
            hl += 1
            body = bytes(enc[hl:])
            mb = model_body(a, level)
            if body != mb:
                k = next((i for i in range(min(len(body), len(mb))) if body[i] != mb[i]), min(len(body), len(mb)))
                msg = " MODEL DIFF at %d (gpu %d B, model %d B)" % (k, len(body), len(mb))
                if a.size <= (1 << 20):
                    os.makedirs("gpurun_out/enc2", exist_ok=True)
                    open("gpurun_out/enc2/%s.gpu" % name, "wb").write(body)
        else:
            msg = " (stored)"
        print("%-12s n=%8d enc=%8d ratio %.4f roundtrip %s%s" % (name, a.size, len(enc), len(enc) / max(a.size, 1), "ok" if ok_rt else "FAIL", msg), flush=True)
        bad += (not ok_rt) or ("DIFF" in msg)
    print("FAILURES:", bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
