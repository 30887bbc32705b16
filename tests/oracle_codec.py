"""Oracle-backed per-rank codecs for the CPU (gloo) runs of minlz_amd/shard.py.  Test infrastructure: lives under tests/
because only tests may import the oracle; the product package never does."""
import numpy as np
import torch

import oracle as O


class OracleTensorCodec:
    """CPU tensors through the oracle: the same interface as minlz_amd.shard.HipTensorCodec."""

    def encode(self, src, block_lens, level):
        stride = max(block_lens + [0]) + 16
        enc = torch.zeros(max(len(block_lens) * stride, 1), dtype=torch.uint8)
        lens, crcs, o = [], [], 0
        a = src.numpy()
        for i, l in enumerate(block_lens):
            e = np.frombuffer(O.encode(a[o:o + l].tobytes(), level), dtype=np.uint8)
            enc[i * stride:i * stride + e.size] = torch.from_numpy(e.copy())
            lens.append(e.size); crcs.append(O.crc(a[o:o + l].tobytes())); o += l
        return enc, stride, torch.tensor(lens, dtype=torch.int64), torch.tensor(crcs, dtype=torch.int64)

    def decode(self, enc, blocks, out):
        a = enc.numpy()
        lens = []
        for so, sl, do, dl in blocks:
            try:
                d = O.decode(a[so:so + sl].tobytes())
            except Exception:
                lens.append(-1)
                continue
            if len(d) != dl:
                lens.append(-1)
                continue
            out[do:do + dl] = torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy())
            lens.append(len(d))
        return torch.tensor(lens, dtype=torch.int64)

    def crcs(self, base, spans):
        a = base.numpy()
        return torch.tensor([O.crc(a[o:o + l].tobytes()) for o, l in spans], dtype=torch.int64)
