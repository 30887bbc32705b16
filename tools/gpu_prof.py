# Needs a library built with -DMLZ_PROFILE=1: bash tools/build_profile_lib.sh, then run with
# MINLZ_HIP_LIB=$PWD/build_var/libminlz_hip_prof.so (the product build compiles the counters out).
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
far = int(sys.argv[1]) if len(sys.argv) > 1 else 1
staged = int(sys.argv[2]) if len(sys.argv) > 2 else 0
S = 100_000_000; BLOCK = 8 << 20
ctx = mz.Context(0); ctx.set_option(mz.OPT_ENCODE_FAR, far); ctx.set_option(1, int(os.environ.get("MLZ_DEC_ALGO", "0")))
host = getattr(synth, os.environ.get("MLZ_WORKLOAD", "enwik_like"))(S, 1); dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
st = torch.cuda.current_stream(dev).cuda_stream
ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
# ---- decode ----
lens = enc_len.cpu().tolist()
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dec_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
ddesc = (BlockDesc * nblk)(*[BlockDesc(i * stride, lens[i], i * BLOCK, blk_len[i]) for i in range(nblk)])
ctx.set_option(4, 0)
ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), ddesc, dec_len.data_ptr()); torch.cuda.synchronize()
ctx.set_option(4, 1)
ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), ddesc, dec_len.data_ptr()); torch.cuda.synchronize()
buf = (C.c_uint64 * 16)(); ctx.set_option(5, C.addressof(buf))
v = list(buf)[8:]
algo = int(os.environ.get("MLZ_DEC_ALGO", "0"))
if algo == 0:
    names = ["part1-rest(Fstore,long)", "turn-wait", "part2+pass", "other(wait,barrier,flush)", "inputs+parse", "classify+Fissue", "literals+tail", "x"]
    ntiles = sum((l + (32 << 10) - 1) // (32 << 10) for l in blk_len)
    print("passes/tile %.1f, cycles per pass copy %.0f" % (v[3] / ntiles, v[7] / max(v[3], 1)))
    print("decode (wave 0 of each tile) cycles/tile: " + ", ".join("%s=%.0f" % (n, x / ntiles) for n, x in zip(names, v[:8])), "tiles", ntiles)
else:
    names = ["loop", "decode+scan", "literals", "classify+wait+Fissue", "N", "Fstore", "S", "n_chunks"]
    print("decode", {n: x for n, x in zip(names, v)})
    print("cycles/chunk: " + ", ".join("%s=%.0f" % (n, x / max(v[7], 1)) for n, x in zip(names[:7], v[:7])), "total/chunk=%.0f" % (sum(v[:7]) / max(v[7], 1)))
