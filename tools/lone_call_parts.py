"""What a lone mlz_decode / mlz_encode call of one 8 MiB block is made of (GPU box): host memcpy rate, pageable and pinned
copies of the block's bytes, the kernels alone, the call itself."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth, _lib
from minlz_amd._lib import BlockDesc
N = 8 << 20
ctx = mz.Context(0)
data = synth.text_like(N, 1)
enc = np.frombuffer(mz.Encode(data, 1, ctx), dtype=np.uint8).copy()
L = _lib.lib()
dev = torch.device("cuda", 0)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
a = np.empty(N, dtype=np.uint8); b = np.empty(N, dtype=np.uint8); a[:] = 1; b[:] = 2
print("host memcpy 8 MiB (numpy, one thread)      : %.3f ms" % t(lambda: np.copyto(b, a)))
pin = torch.empty(N, dtype=torch.uint8, pin_memory=True); pin.zero_()
pag = torch.empty(N, dtype=torch.uint8); pag.zero_()
d = torch.empty(N, dtype=torch.uint8, device=dev)
print("D2H 8 MiB to pinned                        : %.3f ms" % t(lambda: pin.copy_(d, non_blocking=True)))
print("D2H 8 MiB to pageable (runtime staging)    : %.3f ms" % t(lambda: pag.copy_(d)))
print("D2H to pinned, then host memcpy to pageable: %.3f ms" % t(lambda: (pin.copy_(d), np.copyto(b, pin.numpy()))))
ce = torch.from_numpy(enc)
dd = torch.empty(enc.size, dtype=torch.uint8, device=dev)
print("H2D %.1f MB from pageable                    : %.3f ms" % (enc.size / 1e6, t(lambda: dd.copy_(ce))))
desc = (BlockDesc * 1)(BlockDesc(0, enc.size, 0, N)); dl = torch.zeros(1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
print("decode kernels, one 8 MiB block in HBM     : %.3f ms" % t(lambda: ctx.decode_batch_device(st, dd.data_ptr(), d.data_ptr(), desc, dl.data_ptr())))
out = np.empty(N + 64, dtype=np.uint8)
print("mlz_decode, pageable to pageable           : %.3f ms" % t(lambda: L.mlz_decode(ctx.handle, enc.ctypes.data, enc.size, out.ctypes.data, N)))
pe = torch.empty(enc.size, dtype=torch.uint8, pin_memory=True); pe.numpy()[:] = enc
po = torch.empty(N + 64, dtype=torch.uint8, pin_memory=True)
print("mlz_decode, pinned to pinned               : %.3f ms" % t(lambda: L.mlz_decode(ctx.handle, pe.data_ptr(), enc.size, po.data_ptr(), N)))
src = np.frombuffer(data, dtype=np.uint8).copy() if not isinstance(data, np.ndarray) else data
eo = np.empty(N + 4096, dtype=np.uint8)
print("mlz_encode, pageable to pageable           : %.3f ms" % t(lambda: L.mlz_encode(ctx.handle, 1, src.ctypes.data, N, eo.ctypes.data, eo.size)))
