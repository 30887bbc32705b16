"""Encode of batch i+1 beside decode of batch i: two contexts (each has its own workspace) on two streams, two encoded buffers.
The exec pass of the decoder leaves the chip partly idle during its thin level rounds; this measures how much of
that a co-scheduled encode recovers.  usage (GPU box): python tools/pipeline_time.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = 100_000_000; BLOCK = 8 << 20
dev = torch.device("cuda", 0)
host = synth.enwik_like(S, 1)
src = torch.from_numpy(host).to(dev)
nb = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
blk = [min(BLOCK, S - i * BLOCK) for i in range(nb)]
ce, cd = mz.Context(0), mz.Context(0)
enc = [torch.empty(nb * stride, dtype=torch.uint8, device=dev) for _ in range(2)]
el = [torch.zeros(nb, dtype=torch.int64, device=dev) for _ in range(2)]
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dl = torch.zeros(nb, dtype=torch.int64, device=dev)
edesc = (BlockDesc * nb)(*[BlockDesc(i * BLOCK, blk[i], i * stride, stride) for i in range(nb)])
s_e, s_d = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
# sizes once (the same data every step): the decode descriptors need them on the host
ce.encode_batch_device(s_e.cuda_stream, 1, src.data_ptr(), enc[0].data_ptr(), edesc, el[0].data_ptr()); torch.cuda.synchronize()
lens = el[0].cpu().tolist()
ddesc = (BlockDesc * nb)(*[BlockDesc(i * stride, lens[i], i * BLOCK, blk[i]) for i in range(nb)])

def run(steps, overlap):
    ev_e = [torch.cuda.Event() for _ in range(steps)]; ev_d = [torch.cuda.Event() for _ in range(steps)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        b = i & 1
        if i >= 2: s_e.wait_event(ev_d[i - 2])          # the buffer is free once batch i-2 has been decoded
        if not overlap and i >= 1: s_e.wait_event(ev_d[i - 1])
        ce.encode_batch_device(s_e.cuda_stream, 1, src.data_ptr(), enc[b].data_ptr(), edesc, el[b].data_ptr()); ev_e[i].record(s_e)
        s_d.wait_event(ev_e[i])
        cd.decode_batch_device(s_d.cuda_stream, enc[b].data_ptr(), dec.data_ptr(), ddesc, dl.data_ptr()); ev_d[i].record(s_d)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
for _ in range(2): run(4, True)
a = run(K, False); b_ = run(K, True)
ok = bool(torch.equal(dec[:S], src))
print("serial %.3f ms = %.1f GB/s; encode(i+1) beside decode(i) %.3f ms = %.1f GB/s  correct=%s" % (a * 1e3, S / 1e9 / a, b_ * 1e3, S / 1e9 / b_, ok))
