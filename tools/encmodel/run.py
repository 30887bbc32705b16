import ctypes as C, sys, os, subprocess, numpy as np
here=os.path.dirname(os.path.abspath(__file__))
so=os.path.join(here,'libencmodel.so')
subprocess.check_call(['gcc','-O2','-shared','-fPIC','-o',so,os.path.join(here,'encmodel.c')])
L=C.CDLL(so)
class P(C.Structure):
    _fields_=[(k,C.c_int) for k in 'tile_log hash_bits hash_bytes use_rep back_ext far far_bits far_stride far_min epoch_log skip wave nlevels lpat pw_tiles pw_bits ways lazy far_all preseed nohole'.split()]
class S(C.Structure):
    _fields_=[(k,C.c_size_t) for k in 'out n_near n_rep n_far far_bytes near_bytes lit_bytes'.split()]+[('depth_hist',C.c_size_t*16)]
L.model_block.restype=C.c_size_t
L.model_block.argtypes=[C.c_void_p,C.c_size_t,C.POINTER(P),C.POINTER(S)]
def run(data,**kw):
    d=dict(tile_log=16,hash_bits=13,hash_bytes=4,use_rep=1,back_ext=1,far=0,far_bits=17,far_stride=4,far_min=8,epoch_log=20,skip=0,wave=64,nlevels=0,lpat=0,pw_tiles=0,pw_bits=13,ways=1,lazy=0,far_all=0,preseed=0,nohole=0)
    d.update(kw); p=P(**d); st=S()
    a=np.ascontiguousarray(data)
    tot=0
    for o in range(0,a.size,8<<20):
        blk=a[o:o+(8<<20)]
        tot+=L.model_block(blk.ctypes.data,blk.size,C.byref(p),C.byref(st))
    return tot/a.size, st
