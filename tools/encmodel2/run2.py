"""Ratio of the round-2 encoder design on the CPU model (tools/encmodel2/encmodel2.cpp); every stream is decoded
by the oracle.  usage: python tools/encmodel2/run2.py [MiB]"""
import ctypes as C, os, subprocess, sys, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, root)
so = os.path.join(here, 'libencmodel2.so')
subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(here, 'encmodel2.cpp'), '-I' + os.path.join(root, 'minlz_amd', 'csrc')])
L = C.CDLL(so)
FIELDS = 'sub nw near_bits lazy use_rep far lane_cap back dense skip_shift far_gate seed lazy_cost min_far probe_stride far_stride2 lazy_local far_hash24 far_prev'.split()
class P(C.Structure):
    _fields_ = [(k, C.c_int) for k in FIELDS] + [('pattern', C.c_uint), ('graded', C.c_int), ('far_bits', C.c_int), ('far_stride', C.c_int), ('seed_stride', C.c_int), ('pm', C.c_int), ('min_tiles', C.c_int), ('near_unit', C.c_int)]
L.model2_block.restype = C.c_size_t
L.model2_block.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(P), C.c_void_p, C.c_void_p]
DEF = dict(sub=4, nw=4, near_bits=12, lazy=3, use_rep=0, far=1, lane_cap=32, back=8, dense=1, skip_shift=10, far_gate=0, seed=1, lazy_cost=1, min_far=8, probe_stride=1, far_stride2=0, lazy_local=1, far_hash24=2, far_prev=0, pattern=0xA9A9A994, graded=1, far_bits=17, far_stride=4, seed_stride=2, pm=int(os.environ.get('MODEL_PMV', '0')), min_tiles=0, near_unit=992 if os.environ.get('MODEL_PMV') == '1' else 0)  # = the kernels' defaults for blocks of 1 MiB and more (smaller blocks: near_bits=13, see def_for)

# LevelBalanced: 13-bit near tables seeded with every earlier position, far tables twice as dense and twice as large, the
# previous epoch's table probed as well
DEF_L2 = dict(DEF, near_bits=13, near_unit=0, far_prev=1, far_bits=18, far_stride=2, seed_stride=1, pattern=0xffffffff, min_tiles=4)   # pattern 0xffffffff = no tile levels (LevelBalanced's default since round 4); 0 = the dense four-level pattern (option 14 = 0)
DEF_L2_LEVELS = dict(DEF_L2, pattern=0)

def small_far_bits(nbytes, shift=2):
    """mlz_encode.hip.inc small_far_bits: far-table entries (log2) of a LevelFastest block below 1 MiB."""
    lg = 14 + shift
    while lg < 17 + shift and (1 << lg) < nbytes:
        lg += 1
    return lg - shift

def def_for(nbytes, level=1, l2_free=True):
    """The kernels' configuration for a block of nbytes (mlz_encode2.hip.inc: kM2BigBlock)."""
    if level == 2:
        d2 = DEF_L2 if l2_free else DEF_L2_LEVELS
        return dict(d2) if nbytes >= (1 << 20) else dict(d2, far_bits=small_far_bits(nbytes) + 1)
    if nbytes >= (1 << 20):
        return dict(DEF)
    return dict(DEF, near_bits=13, near_unit=0, far_bits=small_far_bits(nbytes))

def run(data, check=True, block=8 << 20, **kw):
    d = dict(DEF); d.update(kw); p = P(**d)
    a = np.ascontiguousarray(data)
    tot = 0; st = np.zeros(16, dtype=np.uint64)
    for o in range(0, a.size, block):
        blk = a[o:o + block]
        out = np.zeros(blk.size + blk.size // 8 + 64, dtype=np.uint8)
        n = L.model2_block(blk.ctypes.data, blk.size, C.byref(p), out.ctypes.data, st.ctypes.data)
        if check:
            from oracle import oracle
            code, dec = oracle.decode_body(out[:n].tobytes(), blk.size)
            assert code == 0 and dec == blk.tobytes(), 'round trip failed'
        tot += n
    return tot / a.size, st

if __name__ == '__main__':
    from minlz_amd import synth
    from oracle import oracle
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    sets = {'text': synth.text_like(mib << 20, seed=1), 'json': synth.json_like(mib << 20, seed=2)}
    for name, data in sets.items():
        data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        oc = sum(len(oracle.encode_block(data[o:o + (8 << 20)].tobytes(), 1)) for o in range(0, data.size, 8 << 20)) / data.size
        print(name, 'oracle L1 %.4f' % oc)
        variants = [
            ('sub1 nw1', dict(sub=1)),
            ('sub4 nw1', dict()),
            ('sub4 nw2', dict(nw=2)),
            ('sub8 nw2', dict(sub=8, nw=2)),
            ('sub4 nw2 norep', dict(nw=2, use_rep=0)),
            ('sub4 nw2 noseed', dict(nw=2, seed=0)),
            ('sub4 nw2 fargate', dict(nw=2, far_gate=1)),
            ('sub4 nw2 back0', dict(nw=2, back=0)),
            ('sub4 nw2 lazy0', dict(nw=2, lazy=0)),
            ('sub4 nw2 lazycost', dict(nw=2, lazy_cost=1)),
            ('sub4 nw2 nofar', dict(nw=2, far=0)),
            ('sub4 nw2 cap16', dict(nw=2, lane_cap=16)),
        ]
        for vn, kw in variants:
            r, st = run(data, **kw)
            print('  %-22s %.4f (x%.3f)  tokens %d lits %d far %d rep %d windows %d' % (vn, r, r / oc, st[0], st[1], st[2], st[3], st[4] // 64))
