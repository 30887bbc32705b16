import sys, torch, numpy as np
sys.path.insert(0,'.')
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
B=8<<20
host=synth.json_like(100_000_000, 77)
dev=torch.device("cuda",0); ctx=mz.Context(0)
import os
if os.environ.get('SETTLE_CAP'): ctx.set_option(20, int(os.environ['SETTLE_CAP']))
BLOCKS=[int(x) for x in os.environ.get('BLOCKS','12,24,64,128').split(',')]
for nblk in BLOCKS:
    S=nblk*B
    src=torch.from_numpy(host).to(dev).repeat((S+host.size-1)//host.size)[:S].contiguous()
    stride=B+256
    enc=torch.empty(nblk*stride,dtype=torch.uint8,device=dev); el=torch.zeros(nblk,dtype=torch.int64,device=dev)
    dec=torch.empty(S+256,dtype=torch.uint8,device=dev); dl=torch.zeros(nblk,dtype=torch.int64,device=dev)
    ed=(BlockDesc*nblk)(*[BlockDesc(i*B,B,i*stride,stride) for i in range(nblk)])
    st=torch.cuda.current_stream(dev).cuda_stream
    ctx.encode_batch_device(st,2,src.data_ptr(),enc.data_ptr(),ed,el.data_ptr()); torch.cuda.synchronize()
    lens=el.cpu().tolist()
    dd=(BlockDesc*nblk)(*[BlockDesc(i*stride,lens[i],i*B,B) for i in range(nblk)])
    ctx.decode_batch_device(st,enc.data_ptr(),dec.data_ptr(),dd,dl.data_ptr()); torch.cuda.synchronize()
    assert torch.equal(dec[:S],src)
    ctx.set_option(mz.OPT_TIMING,2)
    for _ in range(5): ctx.decode_batch_device(st,enc.data_ptr(),dec.data_ptr(),dd,dl.data_ptr())
    torch.cuda.synchronize()
    t=ctx.timers(); tot=sum(v for k,v in t.items() if k.startswith("dec_")); ctx.set_option(mz.OPT_TIMING,0)
    print(nblk,'blocks decode %.3f ms = %.1f GB/s'%(tot,S/1e6/tot), {k:round(v,3) for k,v in t.items() if k.startswith("dec_")}, flush=True)
