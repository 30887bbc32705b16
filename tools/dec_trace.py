"""Per-tile timeline of the exec pass (debug option 7): start / ready / loop end / published, by level."""
# Needs a library built with -DMLZ_PROFILE=1: bash tools/build_profile_lib.sh, then run with
# MINLZ_HIP_LIB=$PWD/build_var/libminlz_hip_prof.so (the product build compiles the counters out).

import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
S = 100_000_000; BLOCK = 8 << 20
ctx = mz.Context(0)
host = getattr(synth, os.environ.get("MLZ_WORKLOAD", "enwik_like"))(S, 1); dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
st = torch.cuda.current_stream(dev).cuda_stream
ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
lens = enc_len.cpu().tolist()
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dec_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
ddesc = (BlockDesc * nblk)(*[BlockDesc(i * stride, lens[i], i * BLOCK, blk_len[i]) for i in range(nblk)])
for _ in range(3):
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), ddesc, dec_len.data_ptr()); torch.cuda.synchronize()
ctx.set_option(4, 1)
ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), ddesc, dec_len.data_ptr()); torch.cuda.synchronize()
buf = np.zeros(4096 * 4, dtype=np.uint64)
ctx.set_option(7, buf.ctypes.data)
t = buf.reshape(-1, 4)
ntiles = sum((l + (32 << 10) - 1) // (32 << 10) for l in blk_len)
t = t[:ntiles]
valid = t[:, 3] != 0
lvl = np.where(valid, (t[:, 3] >> np.uint64(60)).astype(int), -1)   # (-1: not traced — level-0 tiles decoded by dec_level0_kernel)
n_alone = (t[:, 2] >> np.uint64(52)).astype(int); n_pass = ((t[:, 2] >> np.uint64(40)) & np.uint64(0xfff)).astype(int) * 4
tt = t.copy(); tt[:, 3] &= np.uint64((1 << 60) - 1)
hwid = (tt[:, 0] >> np.uint64(40)).astype(np.int64); tt &= np.uint64((1 << 40) - 1)   # (the clock itself has more than 40 bits after some days of uptime: every column is cut the same way)
nchunks = (t[:, 1] >> np.uint64(48)).astype(int)
t0 = tt[valid, 0].min()
us = (tt - t0).astype(np.float64) / 100.0  # 100 MHz -> microseconds
print("tiles", ntiles, "traced", int(valid.sum()), "span %.0f us" % us[valid, 3].max())
for L in range(4):
    m = lvl == L
    if not m.any(): continue
    u = us[m]
    print("level %d n=%d  start %.0f..%.0f  ready %.0f..%.0f (median %.0f)  loopend median %.0f  end %.0f..%.0f | wait med %.0f  loop med %.0f p90 %.0f max %.0f  flush med %.0f" % (
        L, m.sum(), u[:, 0].min(), u[:, 0].max(), u[:, 1].min(), u[:, 1].max(), np.median(u[:, 1]), np.median(u[:, 2]), u[:, 3].min(), u[:, 3].max(),
        np.median(u[:, 1] - u[:, 0]), np.median(u[:, 2] - u[:, 1]), np.percentile(u[:, 2] - u[:, 1], 90), (u[:, 2] - u[:, 1]).max(), np.median(u[:, 3] - u[:, 2])))

for L in range(4):
    m = lvl == L
    if not m.any(): continue
    loop = us[m, 2] - us[m, 1]
    nc = nchunks[m]
    r = np.corrcoef(loop, nc)[0, 1]
    fit = np.polyfit(nc, loop, 1)
    print("level %d: rounds per tile %d..%d (median %d); corr(loop time, rounds) = %.2f; loop = %.2f us/round * rounds + %.0f us; residual std %.1f us" % (
        L, nc.min(), nc.max(), np.median(nc), r, fit[0], fit[1], np.std(loop - np.polyval(fit, nc))))

# level-1 tiles: what the slow ones have in common — position in the block (tile index mod 16), rounds, the CU they ran on
tile_idx = np.concatenate([np.arange((l + (32 << 10) - 1) // (32 << 10)) for l in blk_len])
xcc = hwid & 0xf; hw = hwid >> 4
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = xcc * 1000 + se * 100 + sh * 50 + cu
for L in (1, 2):
    m = lvl == L
    loop = us[m, 2] - us[m, 1]
    print("level %d loop time by tile index mod 16:" % L, {int(k): "%.0f" % np.median(loop[tile_idx[m] % 16 == k]) for k in sorted(set(tile_idx[m] % 16))})
    q = np.argsort(loop)
    lo, hi = q[:len(q) // 10], q[-(len(q) // 10):]
    print("  fastest tenth: loop %.0f, rounds %.1f, copies alone %.0f, passes %.0f;  slowest tenth: loop %.0f, rounds %.1f, copies alone %.0f, passes %.0f;  corr(loop, alone) %.2f corr(loop, passes) %.2f" % (
        loop[lo].mean(), nchunks[m][lo].mean(), n_alone[m][lo].mean(), n_pass[m][lo].mean(), loop[hi].mean(), nchunks[m][hi].mean(), n_alone[m][hi].mean(), n_pass[m][hi].mean(),
        np.corrcoef(loop, n_alone[m])[0, 1], np.corrcoef(loop, n_pass[m])[0, 1]))
    ids, inv, cnt = np.unique(cuid[m], return_inverse=True, return_counts=True)
    per_cu = np.array([loop[inv == i].mean() for i in range(ids.size)])
    print("  %d distinct (xcc, se, sh, cu) ids; mean loop per id: min %.0f median %.0f max %.0f; tiles per id %d..%d" % (ids.size, per_cu.min(), np.median(per_cu), per_cu.max(), cnt.min(), cnt.max()))
    byx = {int(x): "%.0f" % np.median(loop[(xcc[m] == x)]) for x in sorted(set(xcc[m]))}
    print("  median loop by XCC:", byx)
