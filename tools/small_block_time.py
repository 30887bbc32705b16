"""Config-5 shape: many 64 KiB blocks.  Device-resident encode (levels 1, 2) and decode of this library's and of
the oracle's (reference-algorithm) blocks.  Test infrastructure: uses the oracle to make the foreign blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
import oracle as O
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
BLOCK = int(os.environ.get("BLOCK", 64 << 10)); NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = mz.Context(0); dev = torch.device("cuda", 0)
host = synth.text_like(NB * BLOCK, 1)
S = NB * BLOCK
st = torch.cuda.current_stream(dev).cuda_stream
src = torch.from_numpy(host).to(dev)
stride = BLOCK + 256
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
d_enc = torch.zeros(NB * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(NB, dtype=torch.int64, device=dev)
d_dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); d_dlen = torch.zeros(NB, dtype=torch.int64, device=dev)
edesc = (BlockDesc * NB)(*[BlockDesc(i * BLOCK, BLOCK, i * stride, stride) for i in range(NB)])
for level in (1, 2):
    dt = timed(lambda: ctx.encode_batch_device(st, level, src.data_ptr(), d_enc.data_ptr(), edesc, d_len.data_ptr()))
    lens = d_len.cpu().tolist()
    ddesc = (BlockDesc * NB)(*[BlockDesc(i * stride, lens[i], i * BLOCK, BLOCK) for i in range(NB)])
    dd = timed(lambda: ctx.decode_batch_device(st, d_enc.data_ptr(), d_dec.data_ptr(), ddesc, d_dlen.data_ptr()))
    ok = bool(torch.equal(d_dec[:S], src))
    print("%d x %d KiB, level %d: ratio %.4f encode %.2f ms = %.1f GB/s, decode %.2f ms = %.1f GB/s ok=%s" % (
        NB, BLOCK >> 10, level, sum(lens) / S, dt * 1e3, S / dt / 1e9, dd * 1e3, S / dd / 1e9, ok))
nf = min(NB, 1024)   # oracle-encoded sample (L3 is slow on the CPU)
for level in (1, 3):
    encs = [np.frombuffer(O.encode(host[i * BLOCK:(i + 1) * BLOCK], level), dtype=np.uint8) for i in range(nf)]
    buf = np.zeros(nf * stride, dtype=np.uint8)
    for i, e in enumerate(encs): buf[i * stride:i * stride + e.size] = e
    f_enc = torch.from_numpy(buf).to(dev)
    ddesc = (BlockDesc * nf)(*[BlockDesc(i * stride, encs[i].size, i * BLOCK, BLOCK) for i in range(nf)])
    dd = timed(lambda: ctx.decode_batch_device(st, f_enc.data_ptr(), d_dec.data_ptr(), ddesc, d_dlen.data_ptr()))
    ok = bool(torch.equal(d_dec[:nf * BLOCK], src[:nf * BLOCK]))
    print("oracle L%d blocks: %d x %d KiB ratio %.4f decode %.2f ms = %.1f GB/s ok=%s" % (level, nf, BLOCK >> 10, sum(e.size for e in encs) / (nf * BLOCK), dd * 1e3, nf * BLOCK / dd / 1e9, ok))
