#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: MB/s encode+decode, 8 MB blocks, LevelFastest, on N MI355X.

One "step" = one pass of the hot path over one batch: encode every 8 MiB block of a per-GPU stream with the HIP
encoder, then decode every block with the HIP decoder.  Inputs and outputs stay resident in HBM; the C ABI's
device-resident batch calls are timed.  The stream is enwik8 when a path to it is given (--file / $MINLZ_BENCH_FILE;
the file is not in the reference tree and there is no network), else a seeded stand-in that the reference's L1
restatement compresses like text of that kind (ratio ~0.47; synth.enwik_like).  The easier round-1 stand-in
(synth.text_like, ratio ~0.33) is measured beside it and reported under config.r01_standin.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...                      # without a launcher: starts the N ranks itself (fails when the box has fewer GPUs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --mode stream --workload json --level 2 --bytes B [--gpus N]   # BASELINE config 3: ONE stream written and read by N GPUs
    python bench.py --gpus N --single-process          # ONE process, one context over N devices (mlz_init_devices): host-to-host rates, the form a Go host binds
    MINLZ_BENCH_ONE_GPU=1 python bench.py --gpus 2 ...  # TEST mode: N ranks, all on cuda:0, over gloo (host-staged) — runs the N > 1
                                                        # code path on a 1-GPU box; the line carries config.TEST_MODE and is no measurement

Prints ONE JSON line on rank 0.  value = uncompressed bytes through the encode+decode pair per second, whole job
(all ranks).  roofline = dominant kernel against HBM peak (algorithmic bytes N + C per launch / HIP-event launch
time).  cpu_baseline = the CPU oracle (a C restatement of the reference's pure-Go L1 encoder + decoder, NOT the
reference's AMD64 asm) on this box's cores: threads started and buffers touched before the clock, a sweep over
thread counts with the best one reported, and the single-thread rate.  On rank 0 of a 1-GPU run the line also
carries: config.decode_foreign_MBps (the same stream encoded by the reference's algorithm — every such stream
takes the decoder's general path; decode_foreign_2MiB_blocks_MBps: the same in blocks of the reference Writer's
default size), config.end_to_end_MBps (pinned host memory -> mlz_encode_batch /
mlz_decode_batch -> pinned host memory, PCIe included; never the headline value), and short legs for the other BASELINE
configs at one-GPU scale: config3_json_L2 (JSON stream, LevelBalanced, ratio against the oracle's L2), config4_incompressible_1GiB
(every block stored), config5_L3_64KiB_decode (4096 x 64 KiB blocks made by the oracle's LevelSmallest on the CPU), small_stream_blocks
(4 KiB and 16 KiB blocks), each with its kernel times; config3_json_L2_4GiB (config 3 at its stated size on one GPU: block legs, one
framed stream written and read, workspace held) and crc (the device CRC pass).  With N > 1 rank 0 also runs `--single-process` over the N devices as a
child process while the other ranks wait (config.single_process_all_devices: the host-to-host rates of one process over the node).
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

BLOCK = 8 << 20
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def usable_cpus():
    """Threads this process may run on: the affinity mask, capped by a cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(host, budget_s=20.0):
    """The oracle's L1 encoder + decoder over 8 MiB blocks of the same stream, on the host cores."""
    import oracle as O
    sample = host[:min(host.size, 8 * BLOCK)]
    ncpu = usable_cpus()
    # single thread: the per-core rate (the reference quotes ~400 MB/s per core for L1 stream encode, README.md:203-212)
    t1e, cb = O.bench_encode(sample[:2 * BLOCK], BLOCK, 1, 1, 1)
    t1d = O.bench_decode(sample[:2 * BLOCK], BLOCK, 1, 1, 1)
    per_core_e, per_core_d = 2 * BLOCK / 1e6 / t1e, 2 * BLOCK / 1e6 / t1d
    spent = t1e + t1d
    best = None
    sweep = {}
    th = 1
    cand = []
    while th < ncpu:
        cand.append(th)
        th *= 2
    cand.append(ncpu)
    for th in cand:
        if th == 1:
            rate_e, rate_d, reps = per_core_e, per_core_d, 1
        else:
            # enough (rep, block) items to keep every thread busy for several blocks' worth of time
            reps = max(1, (4 * th + 7) // 8)
            est = sample.size * reps / 1e6 * (1 / (per_core_e * min(th, 64)) + 1 / (per_core_d * min(th, 64)))
            if spent + est > budget_s and best is not None:
                break
            te, _ = O.bench_encode(sample, BLOCK, 1, th, reps)
            td = O.bench_decode(sample, BLOCK, 1, th, reps)
            spent += te + td
            rate_e, rate_d = sample.size * reps / 1e6 / te, sample.size * reps / 1e6 / td
        pair = 1.0 / (1.0 / rate_e + 1.0 / rate_d)
        sweep[str(th)] = round(pair, 1)
        if best is None or pair > best[0]:
            best = (pair, th, rate_e, rate_d, reps)
    pair, th, rate_e, rate_d, reps = best
    go = shutil.which("go")
    go_note = None
    if go:  # the reference's own asm path is timed only when Go AND the upstream module are present
        try:
            go_note = subprocess.run([go, "version"], capture_output=True, text=True, timeout=20).stdout.strip()
        except Exception:
            go_note = "go present, version probe failed"
    return {"value": round(pair, 1), "unit": "MB/s", "cores": th, "kind": "port",
            "sample": "%d x 8 MiB blocks of the same stream, %d reps at the best thread count; C restatement of the reference's pure-Go L1 "
                      "encoder + decoder (not its AMD64 asm); threads parked at a barrier with their buffers touched before the clock starts, "
                      "blocks dealt from a shared counter" % (sample.size // BLOCK, reps),
            "encode_MBps": round(rate_e, 1), "decode_MBps": round(rate_d, 1), "ratio": round(cb / (2 * BLOCK), 4),
            "usable_cpus": ncpu, "thread_sweep_pair_MBps": sweep,
            "per_core_MBps": {"encode": round(per_core_e, 1), "decode": round(per_core_d, 1)},
            "reference_asm": ("not timed: %s, but the upstream module github.com/minio/minlz is not on this box (no network)" % go_note) if go
                             else "not timed: `go` is not installed on this box (probed at run time)"}


def stream_leg(ctx, mz, synth, dist, dev, rank, world, total, level, workload, steps, warmup, tile_from=None):
    """BASELINE config 3: ONE stream (JSON-like, LevelBalanced by default), blocks in contiguous ranges over the ranks, every range
    encoded + framed on its GPU, runs gathered in order into rank 0's HBM (Writer side, writer.go:219-272); then the stream just
    written read back by all ranks (Reader side, reader.go:575-992).  Strong scaling: `total` is the whole stream.  Every rank
    calls this; rank 0 gets the result dict (the other ranks None).  tile_from: generate at most that many bytes per rank and
    repeat them on the device (blocks are independent: what a block compresses to does not depend on its neighbours)."""
    from minlz_amd import shard
    n_blocks = (total + BLOCK - 1) // BLOCK
    b0, b1 = shard.range_of(rank, world, n_blocks)
    lo, hi = min(b0 * BLOCK, total), min(b1 * BLOCK, total)
    gen = {"enwik": synth.enwik_like, "text": synth.text_like, "json": synth.json_like, "random": synth.random_bytes}[workload]
    span = hi - lo
    # Everything that can fail on ONE rank alone (its range's memory, the encoder's workspace for it) happens before the first collective, inside a
    # try, and the ranks then AGREE on having got through: a rank that failed while the others went on into all_gather / barrier would hang the
    # whole job instead of reporting (round-5 advisor finding).
    problem = None
    src = codec = None
    try:
        if tile_from and span > tile_from:
            base = torch.from_numpy(gen(tile_from, seed=100 + rank)).to(dev)
            src = base.repeat((span + tile_from - 1) // tile_from)[:span].contiguous()
            del base
        else:
            src = torch.from_numpy(gen(max(span, 1), seed=100 + rank)[:span]).to(dev)     # every rank generates only its own range
        codec = shard.HipTensorCodec(ctx)
        if dist is not None and span > 0:      # a local dry run of the rank's own encode (no collective inside): buffers and workspace at their full size
            lens = [min(BLOCK, span - o) for o in range(0, span, BLOCK)]
            codec.encode(src, lens, level)
            torch.cuda.synchronize(dev)
    except Exception as ex:  # noqa: BLE001
        problem = "%s: %s" % (type(ex).__name__, str(ex)[:200])
    if dist is not None:
        okt = torch.tensor([0 if problem else 1], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            raise RuntimeError("stream leg not run: a rank failed in its local set-up (%s)" % (problem or "another rank"))
    elif problem:
        raise RuntimeError(problem)

    def step():
        return shard.encode_stream_sharded_device(codec, src, total, BLOCK, level, rank, world)
    out = step()
    torch.cuda.synchronize(dev)
    clen = int(out.numel()) if out is not None else 0
    if rank == 0 and world == 1 and total <= (1 << 30):       # whole stream on this rank: check it (the N > 1 layout is covered by tests/test_dist_gloo.py)
        assert mz.stream_decode(out.cpu().numpy().tobytes(), ctx=ctx) == src.cpu().numpy().tobytes()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    ctx.set_option(12, 0xffffffff)
    ctx.set_option(mz.OPT_TIMING, 2)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    t_local = time.perf_counter() - t0          # this rank's encode + frame + its share of the gather (before the barrier)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern = dict(ctx.timers())
    ctx.set_option(mz.OPT_TIMING, 0)
    k_ms = sum(v for k, v in kern.items() if k.startswith("enc_") or k == "crc")
    # ---- the Reader side (reader.go:575-992): the stream just written, decoded by all ranks ----
    # Every rank holds the .mz stream in host memory (a file all ranks can read): chunk walk, upload of ITS span over its own
    # PCIe link, device decode + CRC check, output left sharded in HBM (shard.decode_stream_sharded_device).  PCIe-inclusive
    # by construction (the Reader's input is host bytes), so this is reported beside the encode rate, not as `value`.
    if dist is not None and dist.get_backend() == "nccl":
        # the stream travels as a device tensor (a pickled 1+ GB object through broadcast_object_list would dominate the leg's set-up)
        nb = torch.tensor([clen], dtype=torch.int64, device=dev)
        dist.broadcast(nb, src=0)
        dstream = out if rank == 0 else torch.empty(int(nb.item()), dtype=torch.uint8, device=dev)
        dist.broadcast(dstream, src=0)
        sbytes = torch.empty(int(nb.item()), dtype=torch.uint8, pin_memory=True)
        sbytes.copy_(dstream)
        torch.cuda.synchronize(dev)
        del dstream
    else:
        box = [out.cpu().numpy().tobytes() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(box, src=0)
        sbytes = torch.empty(len(box[0]), dtype=torch.uint8, pin_memory=True)
        sbytes.numpy()[:] = np.frombuffer(box[0], dtype=np.uint8)
        del box
    del out

    def dstep():
        return shard.decode_stream_sharded_device(codec, sbytes, rank, world, dev)
    local, (ulo, uhi), dtotal = dstep()
    torch.cuda.synchronize(dev)
    assert dtotal == total and (ulo, uhi) == (lo, hi) and torch.equal(local, src), "sharded stream decode mismatch"
    del local
    ctx.set_option(mz.OPT_TIMING, 2)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        dstep()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    d_elapsed = time.perf_counter() - t0
    dkern = dict(ctx.timers())
    ctx.set_option(mz.OPT_TIMING, 0)
    dk_ms = sum(v for k, v in dkern.items() if k.startswith("dec_") or k == "crc")
    per_rank, dper_rank, local_ms = [round(k_ms, 4)], [round(dk_ms, 4)], [round(t_local / steps * 1e3, 4)]
    if dist is not None:
        tt = torch.tensor([elapsed, k_ms, d_elapsed, dk_ms, t_local], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        elapsed = max(float(t[0]) for t in allt)
        per_rank = [round(float(t[1]), 4) for t in allt]
        d_elapsed = max(float(t[2]) for t in allt)
        dper_rank = [round(float(t[3]), 4) for t in allt]
        local_ms = [round(float(t[4]) / steps * 1e3, 4) for t in allt]
    if rank != 0:
        return None
    ms = elapsed / steps * 1e3
    dms = d_elapsed / steps * 1e3
    return {"workload": "%s stream of %d B in 8 MiB blocks (%d blocks), level %d, contiguous block ranges per rank, framed chunks (CRC32C on the device) "
                        "gathered in order into rank 0's HBM" % (workload, total, n_blocks, level),
            "scaling": "strong", "writer_MBps": round(total / 1e6 / (elapsed / steps), 1), "ms_per_step": round(ms, 4), "steps": steps,
            "stream_bytes": clen, "ratio": round(clen / max(total, 1), 4), "kernel_ms_per_rank": per_rank,
            # what a rank spends per step outside its kernels: the size all_gather, framing, and the payload gather into rank 0
            "rank_ms_before_barrier": local_ms,
            "gather_and_framing_ms": round(ms - max(per_rank), 4),
            "reader": {"what": "the same stream from host memory on every rank: chunk walk, H2D of the rank's span, device decode + CRC check, output left sharded",
                       "decode_MBps": round(total / 1e6 / (d_elapsed / steps), 1), "ms_per_step": round(dms, 4),
                       "kernel_ms_per_rank": dper_rank, "outside_kernels_ms": round(dms - max(dper_rank), 4)},
            "ranks": dist.get_world_size() if dist is not None else 1,
            "backend": ("rccl (torch nccl)" if dist.get_backend() == "nccl" else dist.get_backend()) if dist is not None else "none (single process)"}


def stream_mode(args, ctx, mz, synth, dist, dev, rank, world):
    """`--mode stream`: the config-3 leg as the whole line (strong scaling: --bytes is the whole stream)."""
    r = stream_leg(ctx, mz, synth, dist, dev, rank, world, args.bytes, args.level, args.workload, args.steps, args.warmup)
    if rank == 0:
        cfg = dict(r)
        value, ms = cfg.pop("writer_MBps"), cfg.pop("ms_per_step")
        cfg.pop("scaling"); cfg.pop("steps")
        cfg["outside_kernels_ms"] = cfg["gather_and_framing_ms"]
        cfg["device"] = ctx.device_name()
        if dist is not None and dist.get_backend() == "gloo":
            cfg["TEST_MODE"] = "MINLZ_BENCH_ONE_GPU: all ranks on one GPU over gloo — exercises the N > 1 code path, not a scaling measurement"
        print(json.dumps({"metric": "MB/s stream encode, 8MB blocks, one stream over N GPUs", "value": value, "unit": "MB/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": cfg}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def host_path_rates(mz, mctx, host, level, block, reps=5):
    """Pinned host -> mlz_encode_batch / mlz_decode_batch / mlz_stream_encode / mlz_stream_decode -> pinned host through ONE context (of one
    device or of several, mlz_init_devices): what a host-language caller of the C ABI sees, PCIe included.  Returns MB/s per direction
    (the median call of `reps`: a call now and then takes twice as long on the host side — page-locked first touches, thread wake-ups)."""
    from minlz_amd import _lib
    L = _lib.lib()
    vp, sz = C.c_void_p, C.c_size_t
    S = host.size
    nblk = (S + block - 1) // block
    blk_len = [min(block, S - i * block) for i in range(nblk)]
    psrc = torch.empty(S, dtype=torch.uint8, pin_memory=True); psrc.numpy()[:] = host
    penc = torch.empty(nblk * (block + 64), dtype=torch.uint8, pin_memory=True); penc.zero_()
    pdec = torch.empty(S, dtype=torch.uint8, pin_memory=True); pdec.zero_()
    sp = (vp * nblk)(*[psrc.data_ptr() + i * block for i in range(nblk)]); sl = (sz * nblk)(*blk_len)
    ep = (vp * nblk)(*[penc.data_ptr() + i * (block + 64) for i in range(nblk)]); ec = (sz * nblk)(*[block + 64] * nblk)
    ol = (C.c_int64 * nblk)()

    def med(fn):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    assert L.mlz_encode_batch(mctx.handle, level, nblk, sp, sl, ep, ec, ol) == 0
    te = med(lambda: L.mlz_encode_batch(mctx.handle, level, nblk, sp, sl, ep, ec, ol))
    cl = (sz * nblk)(*[ol[i] for i in range(nblk)])
    dp = (vp * nblk)(*[pdec.data_ptr() + i * block for i in range(nblk)]); dc = (sz * nblk)(*blk_len)
    dl = (C.c_int64 * nblk)()
    assert L.mlz_decode_batch(mctx.handle, nblk, ep, cl, dp, dc, dl) == 0
    td = med(lambda: L.mlz_decode_batch(mctx.handle, nblk, ep, cl, dp, dc, dl))
    assert bytes(pdec.numpy()) == host.tobytes(), "host-path batch round trip differs"
    # one framed stream (Writer / Reader in one call each, CRCs included)
    cap = L.mlz_stream_bound(S, block, 0)
    pst = torch.empty(cap, dtype=torch.uint8, pin_memory=True); pst.zero_()
    n = L.mlz_stream_encode(mctx.handle, level, block, 0, psrc.data_ptr(), S, pst.data_ptr(), cap)
    assert n > 0, n
    tse = med(lambda: L.mlz_stream_encode(mctx.handle, level, block, 0, psrc.data_ptr(), S, pst.data_ptr(), cap))
    pdec.zero_()
    assert L.mlz_stream_decode(mctx.handle, 0, pst.data_ptr(), n, pdec.data_ptr(), S) == S
    tsd = med(lambda: L.mlz_stream_decode(mctx.handle, 0, pst.data_ptr(), n, pdec.data_ptr(), S))
    assert bytes(pdec.numpy()) == host.tobytes(), "host-path stream round trip differs"
    mb = S / 1e6
    return {"encode": round(mb / te, 1), "decode": round(mb / td, 1), "pair": round(mb / (te + td), 1),
            "stream_encode": round(mb / tse, 1), "stream_decode": round(mb / tsd, 1), "stream_pair": round(mb / (tse + tsd), 1),
            "bytes": int(S), "blocks": int(nblk), "stream_bytes": int(n), "calls_timed": reps}


def single_process_mode(args):
    """`--gpus N --single-process`: ONE process, one context over N devices (mlz_init_devices) — the form a Go host binds.  Host-to-host
    rates through the C ABI's host-pointer calls (pinned memory on both sides), N x --bytes of input dealt to the devices by the library;
    the same stream through one device for comparison.  PCIe-inclusive: each device moves its share over its own link."""
    import minlz_amd as mz
    from minlz_amd import synth
    have = torch.cuda.device_count()
    one_gpu = bool(os.environ.get("MINLZ_BENCH_ONE_GPU"))
    if have < args.gpus and not (one_gpu and have >= 1):
        sys.exit("bench.py --gpus %d --single-process: only %d GPU(s) visible on this box" % (args.gpus, have))
    devices = [0] * args.gpus if one_gpu else list(range(args.gpus))
    S = args.bytes * args.gpus
    gen = {"enwik": lambda: synth.enwik_like(S, seed=1), "text": lambda: synth.text_like(S, seed=1),
           "json": lambda: synth.json_like(S, seed=77), "random": lambda: synth.random_bytes(S, seed=5)}[args.workload]
    host = gen()
    many = mz.Context(devices=devices)
    one = mz.Context(devices[0])
    try:
        for c in (many, one):
            c.set_option(mz.OPT_ENCODE_FAR, args.far)
        r_many = host_path_rates(mz, many, host, args.level, BLOCK, reps=max(3, args.steps // 4))
        r_one = host_path_rates(mz, one, host, args.level, BLOCK, reps=max(3, args.steps // 4))
        cfg = {"workload": "%s, %d x %d bytes in 8 MiB blocks, ONE process, one context over devices %s (mlz_init_devices); host-pointer calls, pinned memory both sides" %
                           (args.workload, args.gpus, args.bytes, devices),
               "devices": devices, "device": many.device_name(), "end_to_end_MBps": r_many, "one_device_same_input_MBps": r_one,
               "speedup_pair": round(r_many["pair"] / r_one["pair"], 3), "speedup_stream_pair": round(r_many["stream_pair"] / r_one["stream_pair"], 3),
               "level": args.level}
        if one_gpu:
            cfg["TEST_MODE"] = "MINLZ_BENCH_ONE_GPU: every context on cuda:0 — exercises the several-device path, not a scaling measurement"
        ms = host.size / 1e6 / r_many["pair"] * 1e3
        print(json.dumps({"metric": "MB/s encode+decode, 8MB blocks L%d, host to host through the C ABI, one process over N MI355X" % args.level,
                          "value": r_many["pair"], "unit": "MB/s", "n_gpus": args.gpus, "steps": max(3, args.steps // 4), "warmup": 1,
                          "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
                          "data": "synthetic", "config": cfg}), flush=True)
    finally:
        many.close(); one.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=100_000_000, help="uncompressed stream bytes per GPU (enwik8 = 1e8)")
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--far", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the foreign-stream, end-to-end and round-1 stand-in legs")
    ap.add_argument("--workload", default="enwik", choices=["enwik", "text", "json", "random"])
    ap.add_argument("--mode", default="blocks", choices=["blocks", "stream"],
                    help="blocks (default, the BASELINE metric): every rank encodes + decodes its own blocks, the compressed payload is gathered to rank 0; "
                         "stream (BASELINE config 3): ONE stream of --bytes total cut into contiguous block ranges over the ranks, framed and gathered "
                         "into rank 0's HBM (minlz_amd/shard.py), strong scaling")
    ap.add_argument("--strong-bytes", type=int, default=4 << 30,
                    help="N > 1, blocks mode: size of the ONE stream of the config-3 strong-scaling leg reported as config.config3_strong (0 = skip)")
    ap.add_argument("--file", default=os.environ.get("MINLZ_BENCH_FILE"), help="real input (e.g. enwik8); every rank reads its own --bytes slice, wrapping around")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process with one context over --gpus devices (mlz_init_devices), host-pointer calls: the form a Go host binds; prints end_to_end_MBps")
    args = ap.parse_args()

    if args.single_process:
        return single_process_mode(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, the same command the
        # driver uses) and pass rank 0's JSON line through.  Fails loudly when the box has fewer than N GPUs.
        have = torch.cuda.device_count()
        if have < args.gpus and not (os.environ.get("MINLZ_BENCH_ONE_GPU") and have >= 1):
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible on this box — not running a smaller world under the same label" % (args.gpus, have))
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MINLZ_BENCH_ONE_GPU=1 (a TEST mode, never a measurement: the line says so): all ranks share cuda:0, each with its own HIP
    # context, and talk through gloo with host-staged transfers (RCCL refuses two ranks on one device) — it runs the N > 1 code
    # path of this file on a 1-GPU box.
    one_gpu = bool(os.environ.get("MINLZ_BENCH_ONE_GPU"))
    if one_gpu:
        local = 0
    assert world == args.gpus, "bench.py --gpus %d was started with WORLD_SIZE=%d: the line would misreport n_gpus" % (args.gpus, world)
    if world > 1 or os.environ.get("MINLZ_BENCH_FORCE_DIST"):   # (the env switch runs the N > 1 code path with one rank: a smoke test)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)   # the ranks RCCL saw
        # RCCL prints its version banner to the C library's stdout when the communicator is made; behind a pipe that buffer is
        # flushed at exit, i.e. AFTER the JSON line.  Make the communicator now and flush, so that the JSON line is the last one.
        _t = torch.zeros(1, device="cpu" if one_gpu else torch.device("cuda", local)); dist.all_reduce(_t); torch.cuda.synchronize()
        C.CDLL(None).fflush(None)
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

    if args.mode == "stream":
        return stream_mode(args, ctx, mz, synth, dist, dev, rank, world)

    # ---- the stream of this rank (weak scaling: every rank has its own S bytes) ----
    S = args.bytes
    if args.file:
        raw = np.fromfile(args.file, dtype=np.uint8)
        assert raw.size > 0, "empty --file"
        start = (rank * S) % raw.size
        host = np.resize(np.roll(raw, -start), S) if raw.size < start + S else raw[start:start + S].copy()
        S = host.size
        wname, data_kind = "file %s" % os.path.basename(args.file), "file"
    else:
        gen = {"enwik": lambda: synth.enwik_like(S, seed=1 + rank), "text": lambda: synth.text_like(S, seed=1 + rank),
               "json": lambda: synth.json_like(S, seed=77 + rank), "random": lambda: synth.random_bytes(S, seed=5 + rank)}[args.workload]
        host = gen()
        wname = {"enwik": "enwik8-like synthetic text (the reference's L1 compresses it to ~0.47; enwik8 itself is not available offline)",
                 "text": "round-1 text-like synthetic stream (phrase table; L1 ~0.33)", "json": "JSON-like synthetic records",
                 "random": "incompressible bytes"}[args.workload]
        data_kind = "synthetic"
    stream = torch.cuda.current_stream(dev).cuda_stream

    class Leg:
        """HBM-resident encode + decode of one stream through the device-resident batch calls."""
        def __init__(self, data, level=None, block=BLOCK):
            """data: numpy uint8 (uploaded) or a uint8 tensor already on the device."""
            self.level = args.level if level is None else level
            self.block = block
            self.src = data if isinstance(data, torch.Tensor) else torch.from_numpy(data).to(dev)
            self.host = None if isinstance(data, torch.Tensor) else data
            self.S = int(self.src.numel())
            self.nblk = (self.S + block - 1) // block
            self.stride = block + 256
            self.enc = torch.empty(self.nblk * self.stride, dtype=torch.uint8, device=dev)
            self.dec = torch.empty(self.S + 256, dtype=torch.uint8, device=dev)
            self.enc_len = torch.zeros(self.nblk, dtype=torch.int64, device=dev)
            self.dec_len = torch.zeros(self.nblk, dtype=torch.int64, device=dev)
            self.blk_len = [min(block, self.S - i * block) for i in range(self.nblk)]
            self.e_desc = (BlockDesc * self.nblk)(*[BlockDesc(i * block, self.blk_len[i], i * self.stride, self.stride) for i in range(self.nblk)])
            self.d_desc = None
            self.clens = None

        def run_encode(self):
            ctx.encode_batch_device(stream, self.level, self.src.data_ptr(), self.enc.data_ptr(), self.e_desc, self.enc_len.data_ptr())

        def make_decode_desc(self, lens=None):
            lens = lens if lens is not None else self.enc_len.cpu().tolist()
            assert all(l > 0 for l in lens), lens
            self.d_desc = (BlockDesc * self.nblk)(*[BlockDesc(i * self.stride, lens[i], i * self.block, self.blk_len[i]) for i in range(self.nblk)])
            self.clens = lens

        def summary(self, steps, warmup=2):
            """check + timed -> the fields a config leg reports."""
            self.check()
            el, kk = self.timed(steps, warmup)
            e_ms = sum(v for k, v in kk.items() if k.startswith("enc_")); d_ms = sum(v for k, v in kk.items() if k.startswith("dec_"))
            return {"bytes": self.S, "blocks": self.nblk, "block_size": self.block, "level": self.level,
                    "value_MBps": round(self.S / 1e6 / (el / steps), 1), "ratio": round(sum(self.clens) / self.S, 4),
                    "encode_MBps": round(self.S / 1e6 / (e_ms / 1e3), 1) if e_ms else None,
                    "decode_MBps": round(self.S / 1e6 / (d_ms / 1e3), 1) if d_ms else None,
                    "kernel_ms": {k: round(v, 4) for k, v in kk.items()}}

        def run_decode(self):
            ctx.decode_batch_device(stream, self.enc.data_ptr(), self.dec.data_ptr(), self.d_desc, self.dec_len.data_ptr())

        def check(self):
            self.run_encode()
            torch.cuda.synchronize(dev)
            self.make_decode_desc()
            self.run_decode()
            torch.cuda.synchronize(dev)
            assert self.dec_len.cpu().tolist() == self.blk_len, "decode reported errors"
            assert torch.equal(self.dec[:self.S], self.src), "GPU decode(encode(x)) != x"

        def timed(self, steps, warmup, collective=None):
            """K steps bracketed by synchronize (+ barrier); returns (seconds, per-kernel HIP-event means in ms).
            Every HIP event pair leaves the device idle for ~10 us at a kernel boundary (rocprofv3 timeline), so the timed
            region records events around the dominant kernel only — its launch time over exactly the timed steps is what the
            roofline uses — and the full per-kernel breakdown comes from a short pass of its own, outside the clock."""
            from minlz_amd import _lib
            L = _lib.lib()
            names = [L.mlz_timer_name(i).decode() for i in range(16)]

            def step():
                self.run_encode()
                if collective is not None:
                    collective(self.enc_len)
                self.run_decode()
                if collective is not None:
                    collective.post()
            for _ in range(warmup):
                step()
            torch.cuda.synchronize(dev)
            # breakdown pass (all timers), also tells which kernel dominates
            ctx.set_option(12, 0xffffffff)
            ctx.set_option(mz.OPT_TIMING, 2)
            for _ in range(max(3, min(steps, 8))):
                step()
            if collective is not None and hasattr(collective, "drain"):
                collective.drain()
            torch.cuda.synchronize(dev)
            kern = dict(ctx.timers())
            ctx.set_option(mz.OPT_TIMING, 0)
            dom = max(kern, key=kern.get) if kern else None
            # timed region
            if dom is not None:
                ctx.set_option(12, 1 << names.index(dom))
                ctx.set_option(mz.OPT_TIMING, 2)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            if collective is not None and hasattr(collective, "drain"):
                collective.drain()
            torch.cuda.synchronize(dev)
            if dist is not None:
                dist.barrier()
            t1 = time.perf_counter()
            if dom is not None:
                kern[dom] = dict(ctx.timers()).get(dom, kern[dom])     # the dominant kernel's mean over the timed steps
            ctx.set_option(mz.OPT_TIMING, 0)
            ctx.set_option(12, 0xffffffff)
            return t1 - t0, kern

    main_leg = Leg(host)
    S, nblk = main_leg.S, main_leg.nblk
    main_leg.check()                       # correctness outside the timed region
    C_total = sum(main_leg.clens)

    gather_info = None
    if dist is not None:
        from minlz_amd import shard
        n_all = nblk * world
        pending = []

        h_len = torch.zeros(nblk, dtype=torch.int64, pin_memory=True)
        len_ready = torch.cuda.Event()

        def collective(enc_len):
            # The Writer's exchange (writer.go:219-272): every rank learns every block's compressed size (an RCCL all_gather of
            # nblk int64) and the compressed blocks travel, in stream order, into rank 0's HBM — isend / irecv of one compact
            # run per rank over xGMI, overlapped with this rank's decode (which reads its own copy).
            # begin (right after the encode launch): the sizes start their way to pinned host memory; nothing waits yet.
            h_len.copy_(enc_len, non_blocking=True)
            len_ready.record(torch.cuda.current_stream(dev))

        def post():
            # (called after the decode LAUNCH: the host waits for the encode's sizes while the device decodes, then posts the gather)
            while pending:
                shard.finish_gather(pending.pop())
            len_ready.synchronize()
            sizes = shard.gather_chunk_sizes(h_len.tolist(), n_all, rank, world, dev)
            mine = sizes[rank * nblk:(rank + 1) * nblk]
            run = torch.cat([main_leg.enc[i * main_leg.stride:i * main_leg.stride + l] for i, l in enumerate(mine)])   # one kernel
            out, works, payload = shard.start_gather(run, sizes, n_all, rank, world, 0)
            pending.append(works)
            collective.keep = (run, out)          # alive until the transfers are done
            collective.payload = payload
        collective.post = post

        def drain():                              # the last step's transfers belong to the timed region
            while pending:
                shard.finish_gather(pending.pop())
        collective.drain = drain
    else:
        collective = None

    elapsed, kavg = main_leg.timed(args.steps, args.warmup, collective)
    if dist is not None:
        while pending:
            shard.finish_gather(pending.pop())
        torch.cuda.synchronize(dev)
        gather_info = {"root": 0, "payload_bytes_per_step": int(collective.payload), "transport": "isend/irecv of one compact run per rank into rank 0's HBM, overlapped with decode"}
    if dist is not None:
        cdev = "cpu" if dist.get_backend() == "gloo" else dev
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ct = torch.tensor([C_total], dtype=torch.int64, device=cdev)
        dist.all_reduce(ct)
        C_all = int(ct.item())
    else:
        C_all = C_total

    # ---- N > 1: BASELINE config 3 as a STRONG-scaling leg in the same line (the headline above is weak scaling: every rank its own
    # stream): one JSON stream of --strong-bytes (default 4 GiB), LevelBalanced, written and read by all ranks (stream_leg) ----
    strong = None
    if dist is not None and args.strong_bytes > 0 and args.level == 1 and not args.file:
        del main_leg.enc, main_leg.dec
        torch.cuda.empty_cache()
        try:
            strong = stream_leg(ctx, mz, synth, dist, dev, rank, world, args.strong_bytes, 2, "json", 3, 1, tile_from=100_000_000)
        except Exception as ex:      # the headline stays valid without it; every rank fails alike (same sizes, same calls) or the barrier would hang
            strong = {"failed": "%s: %s" % (type(ex).__name__, str(ex)[:300])}

    # ---- N > 1: the same node through ONE process (mlz_init_devices over all N devices, host-pointer calls, pinned memory): the form a Go host binds.
    # Rank 0 starts `bench.py --gpus N --single-process` as a CHILD process and the other ranks wait at the barrier below (their GPUs are idle: the timed
    # legs are over).  A child, because that path has never run on more than one GPU: whatever it does on a real node, the line above it survives.
    # PCIe-inclusive, never `value`. ----
    single_process = None
    if dist is not None and world > 1 and args.level == 1 and not args.file:
        if rank == 0:
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                                        "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "MINLZ_BENCH_FORCE_DIST")}
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--single-process", "--steps", "12", "--bytes", str(S),
                                     "--workload", args.workload], env=env, capture_output=True, text=True, timeout=600)
                lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                if pr.returncode == 0 and lines:
                    cj = json.loads(lines[-1])
                    single_process = {"value_MBps": cj["value"], "what": cj["metric"], "end_to_end_MBps": cj["config"]["end_to_end_MBps"],
                                      "one_device_same_input_MBps": cj["config"]["one_device_same_input_MBps"], "devices": cj["config"]["devices"],
                                      "speedup_pair": cj["config"]["speedup_pair"], "speedup_stream_pair": cj["config"]["speedup_stream_pair"]}
                    if "TEST_MODE" in cj["config"]:
                        single_process["TEST_MODE"] = cj["config"]["TEST_MODE"]
                else:
                    single_process = {"failed": "exit code %d: %s" % (pr.returncode, (pr.stderr or "")[-300:])}
            except Exception as ex:  # noqa: BLE001  (the headline stays valid without it)
                single_process = {"failed": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    total_bytes = S * world
    value = total_bytes / 1e6 / (elapsed / args.steps)
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
            kpref = {"enc_tiles": "match_tiles_kernel<true",
                     "dec_exec": "dec_exec2_kernel", "enc_far_build": "far_build_kernel", "dec_parse": "dec_exit_kernel",
                     "enc_serialize": "serialize_pieces_kernel"}.get(dom)
            if kpref and tj.get("workload_bytes") == S and tj.get("workload", "text") == args.workload and not args.file:
                for kn, kv in tj.get("kernels", {}).items():
                    if kn.startswith(kpref):
                        traffic = kv["traffic"]
        step_traffic = None
        if traffic is not None:
            step_traffic = sum(kv["traffic"] for kv in tj.get("kernels", {}).values())
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(kavg[dom], 4),
                    "step_traffic": step_traffic, "step_algorithmic_bytes": 2 * alg,
                    "traffic_source": ("profiles/pmc_traffic.json: a committed rocprofv3 --pmc run of this workload (%s), not measured in this run.  Caveats: the x2 "
                                       "correction of the guide is calibrated for 16 B/lane streaming reads and is applied to every fetch; at 100 MB the step's working "
                                       "set sits in the 256 MiB Infinity Cache, so these are fabric-side bytes (L2 misses), an upper bound on HBM bytes%s"
                                       % (tj.get("commit", "commit not recorded"),
                                          ("; the same passes on a 1 GiB stream (beyond the cache): step traffic %.2f x algorithmic" % tj["large"]["step_traffic_over_algorithmic"])
                                          if isinstance(tj.get("large"), dict) and "step_traffic_over_algorithmic" in tj["large"] else "")) if traffic is not None else None}
        # the two directions as wholes (all kernels of a direction against the same N + C), so that the line names more than its dominant kernel
        roofline["by_direction"] = {
            d: {"ms": round(t, 4), "achieved": round(alg / 1e9 / (t / 1e3), 2), "frac": round(alg / 1e9 / (t / 1e3) / HBM_PEAK_GBS, 5)}
            for d, t in (("enc", enc_ms), ("dec", dec_ms)) if t}

    extras = {}
    if world == 1 and not args.no_extras:
        import oracle as O
        # ---- foreign streams: the same data encoded by the reference's algorithm (oracle L1); decode only ----
        lens = []
        henc = np.zeros(main_leg.nblk * main_leg.stride, dtype=np.uint8)
        for i in range(nblk):
            e = O.encode(host[i * BLOCK:i * BLOCK + main_leg.blk_len[i]], 1)
            henc[i * main_leg.stride:i * main_leg.stride + len(e)] = np.frombuffer(e, dtype=np.uint8)
            lens.append(len(e))
        main_leg.enc.copy_(torch.from_numpy(henc).to(dev))
        main_leg.make_decode_desc(lens)
        main_leg.run_decode()
        torch.cuda.synchronize(dev)
        assert main_leg.dec_len.cpu().tolist() == main_leg.blk_len and torch.equal(main_leg.dec[:S], main_leg.src), "decode of the reference-algorithm stream failed"
        extras["decode_foreign_general_blocks"] = ctx.general_blocks()
        for _ in range(2):
            main_leg.run_decode()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            main_leg.run_decode()
        torch.cuda.synchronize(dev)
        extras["decode_foreign_MBps"] = round(S / 1e6 / ((time.perf_counter() - t0) / 10), 1)
        extras["decode_foreign_ratio"] = round(sum(lens) / S, 4)
        # the kernel furthest below its roofline: the general-block pass on few large blocks (its own timer, a pass outside the clock above)
        ctx.set_option(12, 0xffffffff); ctx.set_option(mz.OPT_TIMING, 2)
        for _ in range(5):
            main_leg.run_decode()
        torch.cuda.synchronize(dev)
        fk = dict(ctx.timers()); ctx.set_option(mz.OPT_TIMING, 0)
        if roofline is not None and fk.get("dec_general"):
            falg = S + sum(lens)
            roofline["foreign"] = {"kernel": "dec_general", "what": "%d reference-made 8 MiB blocks (the oracle's L1 output of the bench stream), decode only" % nblk,
                                   "avg_launch_ms": round(fk["dec_general"], 4), "algorithmic_bytes_per_launch": falg,
                                   "achieved": round(falg / 1e9 / (fk["dec_general"] / 1e3), 2), "frac": round(falg / 1e9 / (fk["dec_general"] / 1e3) / HBM_PEAK_GBS, 5),
                                   "decode_kernel_ms": {k: round(v, 4) for k, v in fk.items() if k.startswith("dec_")}}
        # ... and cut into blocks of the reference Writer's default size (2 MiB, minlz.go:106): what a .mz file made with
        # default options holds
        from minlz_amd._lib import BlockDesc
        FB = 2 << 20
        fn = (S + FB - 1) // FB
        fstride = FB + 256
        fenc = np.zeros(fn * fstride, dtype=np.uint8)
        fl = []
        for i in range(fn):
            e = O.encode(host[i * FB:min(S, (i + 1) * FB)], 1)
            fenc[i * fstride:i * fstride + len(e)] = np.frombuffer(e, dtype=np.uint8)
            fl.append(len(e))
        d_fenc = torch.from_numpy(fenc).to(dev)
        fdesc = (BlockDesc * fn)(*[BlockDesc(i * fstride, fl[i], i * FB, min(FB, S - i * FB)) for i in range(fn)])
        d_flen = torch.zeros(fn, dtype=torch.int64, device=dev)
        fst = torch.cuda.current_stream(dev).cuda_stream
        main_leg.dec.zero_()
        for _ in range(3):
            ctx.decode_batch_device(fst, d_fenc.data_ptr(), main_leg.dec.data_ptr(), fdesc, d_flen.data_ptr())
        torch.cuda.synchronize(dev)
        assert torch.equal(main_leg.dec[:S], main_leg.src), "decode of the reference-algorithm stream (2 MiB blocks) failed"
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.decode_batch_device(fst, d_fenc.data_ptr(), main_leg.dec.data_ptr(), fdesc, d_flen.data_ptr())
        torch.cuda.synchronize(dev)
        extras["decode_foreign_2MiB_blocks_MBps"] = round(S / 1e6 / ((time.perf_counter() - t0) / 10), 1)
        del d_fenc
        # ---- masked CRC32C of the stream's blocks on the device (what the Writer / Reader add per block, minlz.go:133-140) ----
        cdesc = [BlockDesc(i * BLOCK, main_leg.blk_len[i], 0, 0) for i in range(nblk)]
        cout = torch.zeros(nblk, dtype=torch.int32, device=dev)
        for _ in range(3):
            ctx.crc_batch_device(stream, main_leg.src.data_ptr(), cdesc, cout.data_ptr())
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.crc_batch_device(stream, main_leg.src.data_ptr(), cdesc, cout.data_ptr())
        torch.cuda.synchronize(dev)
        tc = (time.perf_counter() - t0) / 20
        assert (int(cout[0].item()) & 0xffffffff) == O.crc(host[:main_leg.blk_len[0]]), "device CRC differs from the oracle's"
        extras["crc"] = {"ms": round(tc * 1e3, 4), "MBps": round(S / 1e6 / tc, 1),
                         "what": "mlz_crc_batch_device over the stream's %d blocks (one pass over N; the Writer / Reader legs of --mode stream include it)" % nblk}
        # ---- end to end through the host-pointer ABI, pinned memory on both sides (PCIe included) ----
        from minlz_amd import _lib
        L = _lib.lib()
        vp, sz = C.c_void_p, C.c_size_t
        psrc = torch.empty(S, dtype=torch.uint8, pin_memory=True); psrc.numpy()[:] = host
        penc = torch.empty(nblk * (BLOCK + 64), dtype=torch.uint8, pin_memory=True); penc.zero_()
        pdec = torch.empty(S, dtype=torch.uint8, pin_memory=True); pdec.zero_()
        sp = (vp * nblk)(*[psrc.data_ptr() + i * BLOCK for i in range(nblk)]); sl = (sz * nblk)(*main_leg.blk_len)
        ep = (vp * nblk)(*[penc.data_ptr() + i * (BLOCK + 64) for i in range(nblk)]); ec = (sz * nblk)(*[BLOCK + 64] * nblk)
        ol = (C.c_int64 * nblk)()
        assert L.mlz_encode_batch(ctx.handle, args.level, nblk, sp, sl, ep, ec, ol) == 0
        t0 = time.perf_counter()
        for _ in range(5):
            L.mlz_encode_batch(ctx.handle, args.level, nblk, sp, sl, ep, ec, ol)
        te = (time.perf_counter() - t0) / 5
        cl = (sz * nblk)(*[ol[i] for i in range(nblk)])
        dp = (vp * nblk)(*[pdec.data_ptr() + i * BLOCK for i in range(nblk)]); dc = (sz * nblk)(*main_leg.blk_len)
        dl = (C.c_int64 * nblk)()
        assert L.mlz_decode_batch(ctx.handle, nblk, ep, cl, dp, dc, dl) == 0
        t0 = time.perf_counter()
        for _ in range(5):
            L.mlz_decode_batch(ctx.handle, nblk, ep, cl, dp, dc, dl)
        td = (time.perf_counter() - t0) / 5
        assert bytes(pdec.numpy()) == host.tobytes()
        extras["end_to_end_MBps"] = {"encode": round(S / 1e6 / te, 1), "decode": round(S / 1e6 / td, 1), "pair": round(S / 1e6 / (te + td), 1),
                                     "note": "pinned host -> mlz_encode_batch / mlz_decode_batch -> pinned host, copies overlapped with kernels in 32 MiB groups"}
        del psrc, penc, pdec
        # ---- the same through ONE context over two per-device contexts on this GPU (mlz_init_devices {d, d}: the single-process fan-out a Go
        # host binds; on one GPU the second context overlaps its copies with the first one's kernels, the PCIe link is shared) ----
        try:
            two = mz.Context(devices=[local, local])
            two.set_option(mz.OPT_ENCODE_FAR, args.far)
            extras["end_to_end_two_contexts_MBps"] = host_path_rates(mz, two, host, args.level, BLOCK, reps=3)
            extras["end_to_end_two_contexts_MBps"]["note"] = "mlz_init_devices({%d, %d}): one process, two per-device contexts on ONE GPU; `python bench.py --gpus N --single-process` is the N-GPU form" % (local, local)
            two.close()
        except Exception as e:  # noqa: BLE001
            extras["end_to_end_two_contexts_MBps"] = {"error": repr(e)}
        # ---- the round-1 stand-in, for continuity with BENCH_r01 ----
        if args.workload == "enwik" and not args.file and args.level == 1:
            leg = Leg(synth.text_like(S, seed=1))
            r = leg.summary(max(3, args.steps // 2))
            r["workload"] = "synth.text_like (the round-1 bench stream)"
            extras["r01_standin"] = r
            del leg
        # ---- BASELINE configs 3, 4, 5 at single-GPU scale, and the smallest stream block sizes (short legs) ----
        if args.workload == "enwik" and not args.file and args.level == 1:
            nth = usable_cpus()
            # config 3: JSON-like stream, LevelBalanced, 8 MiB blocks (the 8-GPU form of it is --mode stream)
            jd = synth.json_like(S, seed=77)
            leg = Leg(jd, level=2)
            r = leg.summary(5)
            _, cb2 = O.bench_encode(jd, BLOCK, 2, nth, 1)
            r["workload"] = ("synth.json_like, LevelBalanced (BASELINE config 3 on one GPU, 12 blocks); since round 4 LevelBalanced writes blocks without tile levels "
                             "(MLZ_OPT_L2_FREE, the reference's ratio) which decode through the general-block path")
            r["oracle_L2_ratio"] = round(cb2 / jd.size, 4)
            r["ratio_vs_oracle_L2"] = round(r["ratio"] / (cb2 / jd.size), 4)
            r["general_blocks"] = ctx.general_blocks()
            ctx.set_option(mz.OPT_L2_FREE, 0)        # the leveled form of rounds 1-3, for comparison
            try:
                legl = Leg(jd, level=2)
                rl = legl.summary(3)
                r["with_tile_levels"] = {"ratio": rl["ratio"], "ratio_vs_oracle_L2": round(rl["ratio"] / (cb2 / jd.size), 4), "value_MBps": rl["value_MBps"],
                                         "encode_MBps": rl["encode_MBps"], "decode_MBps": rl["decode_MBps"]}
                del legl
            finally:
                ctx.set_option(mz.OPT_L2_FREE, 1)
            extras["config3_json_L2"] = r
            del leg
            # config 3 at its stated size on ONE GPU: 4 GiB of the JSON stream (the 100 MB generator output repeated on the device: blocks
            # are independent, so what a block compresses to does not depend on its neighbours), 512 blocks of 8 MiB, LevelBalanced:
            # block legs (encode + decode, HBM-resident), then ONE framed stream written (Writer side) and read back from pinned host
            # memory (Reader side: chunk walk, H2D, decode + CRC check), with the workspace the context held for it.
            try:
                G4 = 4 << 30
                base = torch.from_numpy(jd).to(dev)
                big = base.repeat((G4 + S - 1) // S)[:G4].contiguous()
                del base
                leg4 = Leg(big, level=2)
                r4 = leg4.summary(2, warmup=1)
                r4["general_blocks_in_last_group"] = ctx.general_blocks()   # (a device batch runs in internal groups of 512 MiB = 64 of these blocks)
                ws_e, ws_d = ctx.workspace_bytes()
                r4["workspace_bytes"] = {"encode": ws_e, "decode": ws_d}
                del leg4
                from minlz_amd import shard
                codec = shard.HipTensorCodec(ctx)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                out = shard.encode_stream_sharded_device(codec, big, G4, BLOCK, 2, 0, 1)
                torch.cuda.synchronize(dev)
                tw = time.perf_counter() - t0
                sbytes = torch.empty(int(out.numel()), dtype=torch.uint8, pin_memory=True)
                sbytes.copy_(out)
                del out
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                local, (ulo, uhi), dtotal = shard.decode_stream_sharded_device(codec, sbytes, 0, 1, dev)
                torch.cuda.synchronize(dev)
                tr = time.perf_counter() - t0
                assert dtotal == G4 and (ulo, uhi) == (0, G4) and torch.equal(local, big), "4 GiB stream round trip mismatch"
                r4["stream"] = {"stream_bytes": int(sbytes.numel()), "writer_MBps": round(G4 / 1e6 / tw, 1), "reader_MBps": round(G4 / 1e6 / tr, 1),
                                "note": "one call each (no warm-up): Writer = device-resident source -> framed .mz stream in HBM; Reader = stream in pinned host memory -> "
                                        "chunk walk, H2D, decode + CRC check, output left in HBM"}
                r4["workload"] = "4 GiB of synth.json_like (the 100 MB stream repeated), 512 x 8 MiB blocks, LevelBalanced, one GPU (BASELINE config 3 at its stated size)"
                extras["config3_json_L2_4GiB"] = r4
                del big, local, sbytes
            except RuntimeError as ex:      # (a box without the memory for it: say so instead of failing the line)
                extras["config3_json_L2_4GiB"] = {"skipped": str(ex)[:200]}
            del jd
            torch.cuda.empty_cache()
            # config 4: 1 GiB of incompressible bytes: every block must take the stored path (00 00 raw)
            g = torch.Generator(device=dev); g.manual_seed(4)
            rnd = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device=dev, generator=g)
            leg = Leg(rnd, level=1)
            r = leg.summary(3, warmup=1)
            assert all(c == b + 2 for c, b in zip(leg.clens, leg.blk_len)), "incompressible blocks were not stored"
            r["workload"] = "1 GiB of PRNG bytes (BASELINE config 4): all 128 blocks stored, N read + N written per direction"
            r["stored_blocks"] = leg.nblk
            extras["config4_incompressible_1GiB"] = r
            del leg, rnd
            # config 5: 64 KiB blocks encoded at LevelSmallest by the reference's algorithm on the CPU; device decode only
            td = synth.text_like(512 * 65536, seed=12)
            encs = [O.encode(td[i:i + 65536], 3) for i in range(0, td.size, 65536)]
            reps = 8
            st5 = 65536 + 256
            h5 = np.zeros(len(encs) * st5, dtype=np.uint8)
            for i, e in enumerate(encs):
                h5[i * st5:i * st5 + len(e)] = np.frombuffer(e, dtype=np.uint8)
            d5 = torch.from_numpy(h5).to(dev).repeat(reps)
            n5 = len(encs) * reps
            desc5 = (BlockDesc * n5)(*[BlockDesc(i * st5, len(encs[i % len(encs)]), i * 65536, 65536) for i in range(n5)])
            out5 = torch.empty(n5 * 65536 + 256, dtype=torch.uint8, device=dev)
            len5 = torch.zeros(n5, dtype=torch.int64, device=dev)
            ctx.decode_batch_device(stream, d5.data_ptr(), out5.data_ptr(), desc5, len5.data_ptr())
            torch.cuda.synchronize(dev)
            ref5 = torch.from_numpy(td).to(dev)
            assert bool((len5 == 65536).all()) and all(torch.equal(out5[k * td.size:(k + 1) * td.size], ref5) for k in (0, reps - 1)), "config-5 decode mismatch"
            ctx.set_option(12, 0xffffffff); ctx.set_option(mz.OPT_TIMING, 2)
            t0 = time.perf_counter()
            for _ in range(5):
                ctx.decode_batch_device(stream, d5.data_ptr(), out5.data_ptr(), desc5, len5.data_ptr())
            torch.cuda.synchronize(dev)
            t5 = (time.perf_counter() - t0) / 5
            k5 = dict(ctx.timers()); ctx.set_option(mz.OPT_TIMING, 0)
            extras["config5_L3_64KiB_decode"] = {"workload": "%d x 64 KiB blocks (512 distinct text blocks x %d) encoded by the oracle's LevelSmallest (encode_l3.go restatement) on the CPU; "
                                                             "device decode only (BASELINE config 5)" % (n5, reps),
                                                 "blocks": n5, "decode_MBps": round(n5 * 65536 / 1e6 / t5, 1), "ratio_L3": round(sum(len(e) for e in encs) / td.size, 4),
                                                 "general_blocks": ctx.general_blocks(),
                                                 "kernel_ms": {k: round(v, 4) for k, v in k5.items() if k.startswith("dec_")}}
            del d5, out5, ref5
            # the smallest stream block sizes (writer.go:1238-1246): encode + decode of the bench stream in 4 KiB and 16 KiB blocks
            sm = {}
            for bs in (4096, 16384):
                leg = Leg(host[:64 << 20], level=1, block=bs)
                r = leg.summary(3, warmup=1)
                _, cbs = O.bench_encode(host[:16 << 20], bs, 1, nth, 1)
                r["oracle_L1_ratio_same_blocks"] = round(cbs / (16 << 20), 4)
                sm["%dKiB" % (bs >> 10)] = r
                del leg
            extras["small_stream_blocks"] = sm

    cpu = None
    if not args.no_cpu and world == 1:   # the CPU leg runs on rank 0 of the single-GPU run only
        cpu = cpu_baseline(host)

    cfg = {"workload": "%s, %d B per GPU in 8 MiB blocks (%d blocks), level %d, far=%d; step = encode all blocks + decode all blocks, HBM-resident"
                       % (wname, S, nblk, args.level, args.far),
           "block_size": BLOCK, "bytes_per_gpu": S, "ratio": round(C_all / total_bytes, 4),
           "encode_MBps": round(S / 1e6 / (enc_ms / 1e3), 1) if enc_ms else None,
           "decode_MBps": round(S / 1e6 / (dec_ms / 1e3), 1) if dec_ms else None,
           "kernel_ms": {k: round(v, 4) for k, v in kavg.items()},
           "hbm_read_frac_north_star": round((S / 1e9 / ((enc_ms + dec_ms) / 1e3)) / HBM_PEAK_GBS, 5) if enc_ms + dec_ms else None,
           "ranks": dist.get_world_size() if dist is not None else 1,
           "backend": ("rccl (torch nccl)" if dist.get_backend() == "nccl" else dist.get_backend()) if dist is not None else "none (single process)",
           "device": ctx.device_name()}
    if dist is not None and dist.get_backend() == "gloo":
        cfg["TEST_MODE"] = "MINLZ_BENCH_ONE_GPU: all ranks on one GPU over gloo — exercises the N > 1 code path, not a scaling measurement"
    cfg.update(extras)
    if gather_info:
        cfg["gather"] = gather_info
    if strong is not None:
        cfg["config3_strong"] = strong
    if single_process is not None:
        cfg["single_process_all_devices"] = single_process
    out = {
        "metric": "MB/s encode+decode, 8MB blocks %s" % {1: "L1", 2: "L2 (LevelBalanced)", -1: "L0 (LevelSuperFast)", 0: "uncompressed"}.get(args.level, "level %d" % args.level),
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
        "data": data_kind,
        "config": cfg,
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
