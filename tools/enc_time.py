"""Encode-only kernel times of the bench stream (no decode, no checks): for ablation builds whose output is not a valid parse.
usage (GPU box): MINLZ_HIP_LIB=tools/var/X.so python tools/enc_time.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
wl = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("WL", "enwik")
LEVEL = int(os.environ.get("LEVEL", "1"))
S = 100_000_000; BLOCK = 8 << 20
host = {"enwik": synth.enwik_like, "text": synth.text_like, "json": synth.json_like}[wl](S, 1)
dev = torch.device("cuda", 0)
ctx = mz.Context(0); ctx.set_option(mz.OPT_ENCODE_FAR, int(os.environ.get("FAR", "1")))
nb = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
src = torch.from_numpy(host).to(dev); enc = torch.empty(nb * stride, dtype=torch.uint8, device=dev); el = torch.zeros(nb, dtype=torch.int64, device=dev)
desc = (BlockDesc * nb)(*[BlockDesc(i * BLOCK, min(BLOCK, S - i * BLOCK), i * stride, stride) for i in range(nb)])
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3): ctx.encode_batch_device(st, LEVEL, src.data_ptr(), enc.data_ptr(), desc, el.data_ptr())
torch.cuda.synchronize()
ctx.set_option(mz.OPT_TIMING, 2)
for _ in range(int(os.environ.get('ENC_TIME_REPS', '10'))): ctx.encode_batch_device(st, LEVEL, src.data_ptr(), enc.data_ptr(), desc, el.data_ptr())
torch.cuda.synchronize()
print(os.path.basename(os.environ.get("MINLZ_HIP_LIB", "product")), "ratio %.4f" % (el.sum().item() / S), {k: round(v, 4) for k, v in ctx.timers().items()})
