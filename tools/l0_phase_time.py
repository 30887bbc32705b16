"""How long does the exec pass take for tiles that are ALL level 0 (one-tile blocks), and does it depend on how many share a CU?
usage (GPU box): MINLZ_HIP_LIB=tools/var/X.so python tools/l0_phase_time.py [n_blocks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 191
BLOCK = 32 << 10
S = NB * BLOCK
ctx = mz.Context(0)
host = synth.enwik_like(S, 1); dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); stride = BLOCK + 256
enc = torch.empty(NB * stride, dtype=torch.uint8, device=dev); el = torch.zeros(NB, dtype=torch.int64, device=dev)
desc = (BlockDesc * NB)(*[BlockDesc(i * BLOCK, BLOCK, i * stride, stride) for i in range(NB)])
st = torch.cuda.current_stream(dev).cuda_stream
ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), desc, el.data_ptr()); torch.cuda.synchronize()
lens = el.cpu().tolist()
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dl = torch.zeros(NB, dtype=torch.int64, device=dev)
dd = (BlockDesc * NB)(*[BlockDesc(i * stride, lens[i], i * BLOCK, BLOCK) for i in range(NB)])
ctx.set_option(mz.OPT_TIMING, 1)
acc = []
for it in range(8):
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), dd, dl.data_ptr()); torch.cuda.synchronize()
    if it >= 3: acc.append(ctx.timers()["dec_exec"])
print(os.environ.get("TAG", ""), "blocks", NB, "correct", bool(torch.equal(dec[:S], src)), "dec_exec ms %.4f" % float(np.mean(acc)))
