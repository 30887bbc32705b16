#!/bin/bash
# GPU box: LevelBalanced encode + decode of the config-3 JSON stream (12 x 8 MiB) for far-gap settings. usage: tools/l2dec.sh gap...
cd $GRAFT_REPO_ROOT
for g in "$@"; do
python - <<PY
import sys, json, torch, numpy as np
sys.path.insert(0,'.')
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
S=100_000_000; B=8<<20
for wl in ("json","enwik"):
    host={"json":synth.json_like,"enwik":synth.enwik_like}[wl](S, 77 if wl=="json" else 1)
    dev=torch.device("cuda",0); ctx=mz.Context(0); ctx.set_option(19, $g)
    nb=(S+B-1)//B; stride=B+256
    src=torch.from_numpy(host).to(dev); enc=torch.empty(nb*stride,dtype=torch.uint8,device=dev); el=torch.zeros(nb,dtype=torch.int64,device=dev)
    dec=torch.empty(S+256,dtype=torch.uint8,device=dev); dl=torch.zeros(nb,dtype=torch.int64,device=dev)
    bl=[min(B,S-i*B) for i in range(nb)]
    ed=(BlockDesc*nb)(*[BlockDesc(i*B,bl[i],i*stride,stride) for i in range(nb)])
    st=torch.cuda.current_stream(dev).cuda_stream
    ctx.encode_batch_device(st,2,src.data_ptr(),enc.data_ptr(),ed,el.data_ptr()); torch.cuda.synchronize()
    lens=el.cpu().tolist()
    dd=(BlockDesc*nb)(*[BlockDesc(i*stride,lens[i],i*B,bl[i]) for i in range(nb)])
    ctx.decode_batch_device(st,enc.data_ptr(),dec.data_ptr(),dd,dl.data_ptr()); torch.cuda.synchronize()
    assert dl.cpu().tolist()==bl and torch.equal(dec[:S],src), "roundtrip"
    gen=ctx.general_blocks()
    ctx.set_option(mz.OPT_TIMING,2)
    for _ in range(10): ctx.decode_batch_device(st,enc.data_ptr(),dec.data_ptr(),dd,dl.data_ptr())
    torch.cuda.synchronize()
    t=ctx.timers(); tot=sum(v for k,v in t.items() if k.startswith("dec_"))
    print("gap $g", wl, "ratio %.4f general %d decode %.3f ms = %.1f GB/s" % (sum(lens)/S, gen, tot, S/1e6/tot), {k:round(v,3) for k,v in t.items() if k.startswith("dec_")}, flush=True)
PY
done
