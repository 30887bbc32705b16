# Needs a library built with -DMLZ_PROFILE=1: bash tools/build_profile_lib.sh, then run with
# MINLZ_HIP_LIB=$PWD/build_var/libminlz_hip_prof.so (the product build compiles the counters out).
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
S = 100_000_000; BLOCK = 8 << 20
ctx = mz.Context(0); ctx.set_option(6, 4)
host = synth.text_like(S, 1); dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
st = torch.cuda.current_stream(dev).cuda_stream
ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
ctx.set_option(4, 1)
ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
buf = (C.c_uint64 * 16)(); ctx.set_option(5, C.addressof(buf))
v = list(buf)
nt = 3052
print("producer cycles/tile: plan+wait=%.0f probe=%.0f handover=%.0f | consumer: wait=%.0f select=%.0f emit=%.0f  matches/tile=%.0f" % (v[0]/nt, v[1]/nt, v[2]/nt, v[4]/nt, v[5]/nt, v[7]/nt, v[6]/nt))
