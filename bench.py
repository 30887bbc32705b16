#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: MB/s encode+decode, 8 MB blocks, LevelFastest, on N MI355X.

One "step" = one pass of the hot path over one batch: encode every 8 MiB block of a per-GPU
text-like stream (the enwik8 stand-in of SURVEY.md 8(d) config 2; enwik8 itself is not
available offline) with the HIP encoder, then decode every block with the HIP decoder.  Inputs
and outputs stay resident in HBM; the C ABI's device-resident batch calls are timed.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  value = uncompressed bytes through the encode+decode pair per
second, whole job (all ranks).  roofline = dominant kernel against HBM peak (algorithmic bytes
N + C per launch / HIP-event launch time); cpu_baseline = the CPU oracle (a C restatement of the
reference's pure-Go L1 encoder + decoder, NOT the reference's AMD64 asm) on this box's cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

BLOCK = 8 << 20
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=100_000_000, help="uncompressed stream bytes per GPU (enwik8 = 1e8)")
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--far", type=int, default=1)
    ap.add_argument("--staged", type=int, default=-1, help="encoder variant: 0 in-place (default), 1 LDS-staged, 3 software-pipelined; -1 = library default")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--workload", default="text", choices=["text", "json", "random"])
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists for the product path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import minlz_amd as mz
    from minlz_amd import synth
    from minlz_amd._lib import BlockDesc

    ctx = mz.Context(local)
    ctx.set_option(mz.OPT_ENCODE_FAR, args.far)
    if args.staged >= 0:
        ctx.set_option(6, args.staged)

    # ---- synthetic stream for this rank (weak scaling: every rank has its own S bytes) ----
    S = args.bytes
    gen = {"text": lambda: synth.text_like(S, seed=1 + rank), "json": lambda: synth.json_like(S, seed=77 + rank),
           "random": lambda: synth.random_bytes(S, seed=5 + rank)}[args.workload]
    host = gen()
    nblk = (S + BLOCK - 1) // BLOCK
    src = torch.from_numpy(host).to(dev)
    stride = BLOCK + 256
    enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev)
    dec = torch.empty(S + 256, dtype=torch.uint8, device=dev)
    enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
    dec_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
    blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
    e_desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run_encode():
        ctx.encode_batch_device(stream, args.level, src.data_ptr(), enc.data_ptr(), e_desc, enc_len.data_ptr())

    d_desc_box = [None]

    def make_decode_desc():
        lens = enc_len.cpu().tolist()
        assert all(l > 0 for l in lens), lens
        d_desc_box[0] = (BlockDesc * nblk)(*[BlockDesc(i * stride, lens[i], i * BLOCK, blk_len[i]) for i in range(nblk)])
        return lens

    def run_decode():
        ctx.decode_batch_device(stream, enc.data_ptr(), dec.data_ptr(), d_desc_box[0], dec_len.data_ptr())

    # ---- correctness outside the timed region ----
    run_encode()
    torch.cuda.synchronize(dev)
    clens = make_decode_desc()
    run_decode()
    torch.cuda.synchronize(dev)
    assert dec_len.cpu().tolist() == blk_len, "decode reported errors"
    assert torch.equal(dec[:S], src), "GPU decode(encode(x)) != x"
    C_total = sum(clens)

    gathered = [torch.empty_like(enc_len) for _ in range(world)] if dist is not None else None

    def step():
        run_encode()
        if dist is not None:
            # the stream writer's only exchange: every rank learns every block's compressed size
            # (output offsets / index, writer.go:223-243) — an RCCL all_gather of nblk int64
            dist.all_gather(gathered, enc_len)
        run_decode()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    ctx.set_option(mz.OPT_TIMING, 2)   # running mean of the per-kernel HIP-event times; read once after the loop (no per-step sync)
    kern = {}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    for k, v in ctx.timers().items():  # HIP events recorded on the launch stream by the library, averaged over the K steps
        kern[k] = [v]
    ctx.set_option(mz.OPT_TIMING, 0)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ct = torch.tensor([C_total], dtype=torch.int64, device=dev)
        dist.all_reduce(ct)
        C_all = int(ct.item())
    else:
        C_all = C_total

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    total_bytes = S * world
    value = total_bytes / 1e6 / (elapsed / args.steps)
    kavg = {k: float(np.mean(v)) for k, v in kern.items()}
    enc_ms = sum(v for k, v in kavg.items() if k.startswith("enc_"))
    dec_ms = sum(v for k, v in kavg.items() if k.startswith("dec_"))
    # dominant kernel and its roofline (algorithmic bytes per launch = N + C of this rank's batch)
    dom = max(kavg, key=kavg.get) if kavg else None
    roofline = None
    if dom:
        alg = S + C_total
        ach = alg / 1e9 / (kavg[dom] / 1e3)
        # HBM-side bytes per launch from rocprofv3 PMC passes (tools/pmc_run.sh -> tools/pmc_traffic.py), recorded for
        # this exact workload; null when no matching measurement is committed
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            kname = {"enc_tiles": "encode_tiles_kernel<true, false, %d>" % (2 if args.level == 2 else 1), "dec_exec": "dec_exec2_kernel", "enc_far_build": "far_build_kernel",
                     "dec_parse": "dec_exit_kernel"}.get(dom)
            if tj.get("workload_bytes") == S and args.workload == "text" and kname in tj.get("kernels", {}):
                traffic = tj["kernels"][kname]["traffic"]
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(kavg[dom], 4)}

    cpu = None
    if not args.no_cpu and world == 1:   # the CPU leg runs on rank 0 of the single-GPU run only
        import oracle as O
        threads = os.cpu_count() or 1
        sample = host[:min(S, 8 * BLOCK)]
        # calibrate reps for ~10 s of CPU work
        t_e, _ = O.bench_encode(sample, BLOCK, 1, threads, 1)
        reps = max(1, min(50, int(5.0 / max(t_e, 1e-3))))
        t_e, cbytes = O.bench_encode(sample, BLOCK, 1, threads, reps)
        t_d = O.bench_decode(sample, BLOCK, 1, threads, reps)
        cpu = {"value": round(sample.size * reps / 1e6 / (t_e + t_d), 1), "unit": "MB/s", "cores": threads, "kind": "port",
               "sample": "%d x 8 MiB blocks of the same stream, %d reps, one block per thread; C restatement of the reference's pure-Go L1 "
                         "encoder+decoder (not its AMD64 asm)" % (sample.size // BLOCK, reps),
               "encode_MBps": round(sample.size * reps / 1e6 / t_e, 1), "decode_MBps": round(sample.size * reps / 1e6 / t_d, 1),
               "ratio": round(cbytes / sample.size, 4)}

    out = {
        "metric": "MB/s encode+decode, 8MB blocks L1",
        "value": round(value, 1),
        "unit": "MB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "%s-like synthetic stream (enwik8 stand-in), %d B per GPU in 8 MiB blocks (%d blocks), level %d, far=%d; "
                               "step = encode all blocks + decode all blocks, HBM-resident" % (args.workload, S, nblk, args.level, args.far),
                   "block_size": BLOCK, "bytes_per_gpu": S, "ratio": round(C_all / total_bytes, 4),
                   "encode_MBps": round(S / 1e6 / (enc_ms / 1e3), 1) if enc_ms else None,
                   "decode_MBps": round(S / 1e6 / (dec_ms / 1e3), 1) if dec_ms else None,
                   "kernel_ms": {k: round(v, 4) for k, v in kavg.items()},
                   "hbm_read_frac_north_star": round((S / 1e9 / ((enc_ms + dec_ms) / 1e3)) / HBM_PEAK_GBS, 5) if enc_ms + dec_ms else None,
                   "device": ctx.device_name()},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
