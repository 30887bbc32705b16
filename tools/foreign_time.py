"""Decode time of reference-algorithm streams (oracle L1 output, 8 MiB blocks) by kernel family.
usage (GPU box): [MINLZ_HIP_LIB=tools/var/gs1.so] [BLOCK=bytes] python tools/foreign_time.py [workload] [MB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
import oracle as O
wl = sys.argv[1] if len(sys.argv) > 1 else "enwik"
S = int(sys.argv[2]) * 1_000_000 if len(sys.argv) > 2 else 100_000_000
BLOCK = int(os.environ.get("BLOCK", 8 << 20))
host = {"enwik": synth.enwik_like, "text": synth.text_like, "json": synth.json_like}[wl](S, 1)
nb = (S + BLOCK - 1) // BLOCK
dev = torch.device("cuda", 0)
ctx = mz.Context(0)
stride = BLOCK + 256
henc = np.zeros(nb * stride, dtype=np.uint8); lens = []
for i in range(nb):
    e = O.encode(host[i * BLOCK:(i + 1) * BLOCK], 1)
    henc[i * stride:i * stride + len(e)] = np.frombuffer(e, dtype=np.uint8); lens.append(len(e))
enc = torch.from_numpy(henc).to(dev)
dec = torch.zeros(S + 256, dtype=torch.uint8, device=dev)
dl = torch.zeros(nb, dtype=torch.int64, device=dev)
desc = (BlockDesc * nb)(*[BlockDesc(i * stride, lens[i], i * BLOCK, min(BLOCK, S - i * BLOCK)) for i in range(nb)])
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), desc, dl.data_ptr())
torch.cuda.synchronize()
ok = bytes(dec[:S].cpu().numpy()) == host.tobytes()
ctx.set_option(mz.OPT_TIMING, 2)
t0 = time.perf_counter()
for _ in range(10):
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), desc, dl.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("%s %d MB in %d KiB blocks: %.3f ms = %.1f GB/s  correct=%s  general=%d  timers %s" % (wl, S // 1_000_000, BLOCK >> 10, dt * 1e3, S / 1e9 / dt, ok, ctx.general_blocks(),
      {k: round(v, 3) for k, v in ctx.timers().items() if k.startswith("dec")}))
