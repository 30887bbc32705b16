"""Device CRC32C throughput (mlz_crc_batch_device) on the bench workload's blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
S = 100_000_000; BLOCK = 8 << 20
ctx = mz.Context(0); dev = torch.device("cuda", 0)
host = synth.text_like(S, 1); src = torch.from_numpy(host).to(dev)
nblk = (S + BLOCK - 1) // BLOCK
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = [BlockDesc(i * BLOCK, blk_len[i], 0, 0) for i in range(nblk)]
out = torch.zeros(nblk, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3): ctx.crc_batch_device(st, src.data_ptr(), desc, out.data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ctx.crc_batch_device(st, src.data_ptr(), desc, out.data_ptr())
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
import oracle as O
ok = [int(x) & 0xffffffff for x in out.cpu().tolist()] == [O.crc(host[i * BLOCK:i * BLOCK + blk_len[i]]) for i in range(nblk)]
print("crc of %d blocks (100 MB): %.3f ms = %.0f GB/s, matches oracle: %s" % (nblk, dt * 1e3, S / 1e9 / dt, ok))
