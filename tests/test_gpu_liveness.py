"""The decoders' forward progress beside OTHER work on the device.

The exec pass and the general-block pass have workgroups that wait for flags raised by workgroups of a lower index (mlz_decode_exec.hip.inc,
mlz_decode_general.hip.inc): safe as long as the lower-indexed ones are dispatched first — which HIP does not promise in words, and which is
most at risk when another queue keeps the CUs busy (a co-running tenant; the reference has no such concern: decode.go:178-622 is one loop
per block).  The waits are bounded (a time-out surfaces as -MLZ_ERR_HIP, the shim's cue for its CPU path), so the failure mode would be
"slow and counted", never "wrong" — this test looks for it under a saturating matmul loop on a second stream and on a second thread."""
import threading
import time

import numpy as np
import pytest

import minlz_amd as mz
import oracle as O
from minlz_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _busy(stop, started, counts):
    import torch
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    with torch.cuda.stream(s):
        while not stop.is_set():
            for _ in range(8):
                a = (a @ b).clamp_(-1, 1)          # every CU busy with matrix work
                big.add_(1)                        # ... and the HBM with a streaming pass
            s.synchronize()
            counts[0] += 8
            started.set()


def test_decode_beside_a_busy_queue():
    ctx = mz.Context(0)
    stop, started, counts = threading.Event(), threading.Event(), [0]
    th = threading.Thread(target=_busy, args=(stop, started, counts), daemon=True)
    try:
        data = [np.ascontiguousarray(synth.text_like((4 << 20) + 777 * i, seed=80 + i) if i & 1 else synth.json_like((3 << 20) + 99 * i, seed=90 + i)) for i in range(10)]
        want = [d.tobytes() for d in data]
        own = mz.encode_batch(data, mz.LevelFastest, ctx)              # tile-levelled streams: the exec pass's ordered waits
        bal = mz.encode_batch(data, mz.LevelBalanced, ctx)             # general blocks, teams of four workgroups
        ref = [O.encode(d, 1 + (i & 1)) for i, d in enumerate(data)]   # the reference algorithm's blocks: one settling workgroup per block
        quiet = [mz.decode_batch(b, ctx) for b in (own, bal, ref)]
        assert quiet == [want, want, want]
        th.start()
        assert started.wait(120), "the co-running load did not start"
        before = counts[0]
        t0 = time.time()
        rounds = 0
        while rounds < 6 or (time.time() - t0 < 10 and rounds < 60):
            for name, blocks in (("own", own), ("balanced", bal), ("reference", ref), ("mixed", own[:3] + ref[3:6] + bal[6:])):
                got = mz.decode_batch(blocks, ctx)                      # a time-out would raise ErrHIP here
                assert got == want, name
            # the encoder has no ordered waits, but it shares the device all the same
            assert mz.encode_batch(data[:4], mz.LevelFastest, ctx) == own[:4]
            rounds += 1
        assert counts[0] > before, "the co-running load made no progress: the test did not test anything"
        assert int(_lib.lib().mlz_get_counter(ctx.handle, 5)) == 0   # no call fell back to the tile chain
    finally:
        stop.set()
        if th.is_alive():
            th.join(60)
        ctx.close()
