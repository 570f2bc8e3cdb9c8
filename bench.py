#!/usr/bin/env python3
"""bench.py -- the hot path's headline measurement (BASELINE.json).

A "step" is one pass of the fused demodulator over one batch of synthetic input
that is already resident in HBM: `channels` independent WBFM channels x `blocks`
consecutive 262144-byte blocks of int8 IQ (2.048 MS/s) -> 8 kS/s int16 PCM.
Streams continue from step to step (per-channel state is carried on the device).

    python bench.py --gpus 1 --steps 200 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: channels are independent, so every rank owns its own `channels`
channels on its own GPU (weak scaling); there is no data-path collective.  The
ranks only meet for the barrier around the timed region and a MAX over elapsed
times.  Rank 0 prints ONE JSON line.

Extra objects in the JSON line:
  roofline      algorithmic HBM bytes per launch / mean duration of the dominant
                kernel, measured with HIP events on the launch stream during the
                timed region, against the 8 TB/s HBM3E peak
  cpu_baseline  the reference's own CPU chain (oracle/_ref, compiled from the
                reference sources) timed on this box's host cores on a bounded
                sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BLOCK = 262144
SETTLE_STEPS = 100        # untimed launches in front of the timed region, at least (clock settling)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_fm_batch(channels, blocks, device, first_channel=0):
    """FM test signal of SURVEY.md 8(d), generated on the GPU: carrier at -64 kHz,
    +-30 kHz deviation by a (300 + 100*(c mod 32)) Hz tone, amplitude 100, uniform
    noise in [-3,4].  (Same signal family as hackrfdiags_amd/synth.py; the noise
    comes from torch's generator here, so the bytes differ from the test vectors.)"""
    n = blocks * (BLOCK // 2)
    out = torch.empty((channels, blocks, BLOCK), dtype=torch.int8, device=device)
    k = torch.arange(n, dtype=torch.float64, device=device)
    gen = torch.Generator(device=device)
    for c in range(channels):
        ch = first_channel + c
        f_c = 300.0 + 100.0 * (ch % 32)
        beta = 30000.0 / f_c
        phi = (2.0 * np.pi * (-64000.0) / 2048000.0) * k - beta * (torch.cos((2.0 * np.pi * f_c / 2048000.0) * k) - 1.0)
        gen.manual_seed(12345 + ch)
        noise = torch.randint(-3, 5, (2, n), device=device, generator=gen, dtype=torch.int32)
        i = torch.round(100.0 * torch.cos(phi)).to(torch.int32) + noise[0]
        q = torch.round(100.0 * torch.sin(phi)).to(torch.int32) + noise[1]
        iq = torch.stack([i, q], dim=1).to(torch.int8)          # [n, 2] interleaved
        out[c] = iq.reshape(blocks, BLOCK)
    return out


def make_random_batch(channels, blocks, device, first_channel=0):
    gen = torch.Generator(device=device)
    gen.manual_seed(1 + first_channel)
    return torch.randint(-128, 128, (channels, blocks, BLOCK), dtype=torch.int8, device=device, generator=gen)


def cpu_baseline(seconds, threads):
    """Reference CPU chain (IqDataProcessor::acceptIqData in WBFM mode), one
    channel per thread, each looping over the same 8-block FM test signal."""
    from hackrfdiags_amd import synth
    from tests import reflib
    kind = "reference"
    try:
        eng = reflib.Ref()
    except (FileNotFoundError, OSError):
        eng = reflib.Oracle()          # our restatement, if the prebuilt reference did not travel
        kind = "port"
    nb = 8
    x = synth.make_input("fmtone", 0, nb).reshape(nb, BLOCK)
    handles = []
    for _ in range(threads):
        h = eng.rx()
        h.set_mode(reflib.WBFM)
        handles.append(h)
    counts = [0] * threads
    stop = time.perf_counter() + seconds

    def work(i):
        h = handles[i]
        n = 0
        while time.perf_counter() < stop:
            for b in range(nb):
                h.process(x[b])
            n += nb
        counts[i] = n

    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    blocks = sum(counts)
    return {
        "value": round(blocks * (BLOCK // 2) / dt / 1e6, 2),
        "unit": "MSamples/s",
        "cores": threads,
        "kind": kind,
        "sample": f"{blocks} blocks of 262144 B (FM test signal, WBFM mode) over {dt:.1f} s, "
                  f"{threads} thread(s), one channel per thread",
    }


def bench_ssbmod(args, api, device, rank, world, dist):
    """BASELINE config 5: `channels` SSB modulators, 512 PCM samples (64 ms) per block,
    `blocks` blocks per step -> int8 IQ at 2.048 MS/s.  Unit of work: one output IQ sample."""
    from hackrfdiags_amd import shard
    C, B = args.channels, args.blocks
    n = 512 * B
    gen = torch.Generator(device=device)
    gen.manual_seed(7 + rank)
    pcm = torch.randint(-32768, 32768, (C, n), dtype=torch.int16, device=device, generator=gen)
    out = torch.empty((C, 512 * n), dtype=torch.int8, device=device)
    torch.cuda.synchronize()                             # the PCM was generated on torch's stream
    kind = {"ssbmod": api.MOD_SSB, "ammod": api.MOD_AM, "fmmod": api.MOD_FM, "wbfmmod": api.MOD_WBFM}[args.workload]
    kname = {"ssbmod": "SSB", "ammod": "AM", "fmmod": "FM", "wbfmmod": "WBFM"}[args.workload]
    m = api.Mod(kind, C, device=device.index)
    stream = torch.cuda.Stream(device=device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record(stream)
        m.process_device(pcm.data_ptr(), n, out.data_ptr(), stream=stream.cuda_stream)
        if i is not None:
            ev[i][1].record(stream)

    for _ in range(max(0, SETTLE_STEPS - args.warmup) + args.warmup):   # see SETTLE_STEPS
        step()
    m.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    stream.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device)
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    samples = C * n * 256
    value = world * samples * args.steps / elapsed / 1e6
    algo_bytes = C * n * (2 + 512)
    mean_ms = float(np.mean(kernel_ms))
    achieved = algo_bytes / (mean_ms * 1e-3) / 1e9
    fill = None
    if rank == 0 and not args.no_extras:
        fill = stream_fill_gbs(device, out)
    if rank == 0:
        print(json.dumps({
            "metric": f"IQ MSamples/s modulated (8 kS/s PCM -> 2.048 MS/s int8 IQ, {kname}) per GPU; % HBM roofline",
            "value": round(value, 1), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int16 PCM -> Q15 int32 -> int8 IQ", "data": "synthetic",
            "config": {"workload": f"{C} {kname} modulator channels per GPU (BASELINE config 5" + ("" if (C, args.workload) == (1024, "ssbmod") else
                                   " is 1024 SSB channels") + f"), {B} blocks of 512 PCM "
                                   f"samples per step, 8-stage x256 half-band interpolator", "channels_per_gpu": C,
                       "blocks_per_step": B},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "measured_stream_fill_GBps": None if fill is None else round(fill, 1),
                         "frac_of_measured_fill": None if fill is None else round(achieved / fill, 4),
                         "kernel": {"ssbmod": "hrfd::k_mod<1>", "wbfmmod": "hrfd::k_mod<101> (x32 + Nco step), k_phase_scan, k_wb_rails, hrfd::k_mod<102> (x8)"}.get(
                             args.workload, "k_am_rails / k_fm_phase + k_fm_rails, then hrfd::k_mod<100>"),
                         "kernel_ms_mean": round(mean_ms, 4), "algorithmic_bytes_per_launch": algo_bytes},
        }), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_ingest(args, api, device, rank, world, dist):
    """PCIe-inclusive rate: host batches of [channels][blocks][262144] int8 in pinned memory through
    hrfd_ingest_* (H2D, WBFM demodulation, D2H of the PCM on three streams, two batches in
    flight).  Not the headline metric: the boundary hands over host buffers here."""
    from hackrfdiags_amd import shard
    C, B = args.channels, min(args.blocks, 4)
    rx = api.Rx(C, device=device.index)
    rx.set_mode(api.WBFM)
    ing = api.Ingest(rx, BLOCK, B, 2)
    x = make_fm_batch(C, B, device, first_channel=rank * C).cpu().numpy()
    for _ in range(2):                                   # both pinned slots hold valid IQ
        ing.acquire()[...] = x
        ing.submit(0)
    for _ in range(2):
        ing.collect()
    for _ in range(args.warmup):
        ing.acquire(); ing.submit(0); ing.collect()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ing.acquire(); ing.submit(0)
    for i in range(args.steps):
        if i + 1 < args.steps:
            ing.acquire(); ing.submit(0)                 # batch i+1 travels while batch i is demodulated
        ing.collect()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device)
    samples = C * B * (BLOCK // 2)
    value = world * samples * args.steps / elapsed / 1e6
    if rank == 0:
        print(json.dumps({
            "metric": "IQ MSamples/s demodulated, host to host (pinned batches over PCIe, WBFM) per GPU",
            "value": round(value, 1), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8 IQ -> int16 PCM", "data": "synthetic",
            "config": {"workload": f"{C} WBFM channels, batches of {B} blocks from pinned host memory, 2 in flight "
                                   f"(hrfd_ingest_*: H2D + kernels + D2H overlapped)", "channels_per_gpu": C,
                       "blocks_per_step": B},
            "pcie_GBps": round(C * B * BLOCK * args.steps / elapsed / 1e9, 2),
            "replayed_batches": ing.replayed(),
            **({"invalid": "batches were replayed on the exact path"} if ing.replayed() else {}),
        }), flush=True)
    ing.close()
    if dist is not None:
        dist.destroy_process_group()


def kernel_source_tag():
    """sha256 (first 16 hex digits) of the kernel sources: ties a committed PMC summary to the code it measured"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hackrfdiags_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic_bytes(args, C, B):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/profile_round.sh
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own runs of this same command; the summary
    is committed under profiles/).  gfx950: FETCH_SIZE counts 64 B per 128-B request for wide
    coalesced reads, hence the factor 2 (MI355X_MICROARCH.md).  None when the workload differs
    from the profiled one OR when the kernel sources have changed since that profile (the summary
    carries the source tag it was taken with)."""
    if args.workload != "wbfm" or (C, B) != (256, 16):
        return None, None
    path = os.path.join(ROOT, "profiles", "latest_pmc_traffic_wbfm256x16.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("kernel_source_tag") != kernel_source_tag():
            return None, t.get("kernel_source_tag")
        return int(t["FETCH_SIZE"]["mean"] * 1024 * 2 + t["WRITE_SIZE"]["mean"] * 1024), t.get("kernel_source_tag")
    except (OSError, KeyError, ValueError):
        return None, None


def stream_copy_gbs(device):
    """Second denominator: what a plain device-to-device copy of 1 GiB reaches on this GPU in this
    run, read + write bytes over the time of the copy kernel (HIP events, 10 copies after 3 warm-ups)."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.int8, device=device)
    b = torch.empty(n, dtype=torch.int8, device=device)
    a.random_(0, 127)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    del a, b
    return 2 * n / (ms * 1e-3) / 1e9


def stream_fill_gbs(device, out):
    """The write-side counterpart: what the library's fill of the modulators' own output buffer reaches on this GPU in
    this run (bytes written over the time of the fill kernel, HIP events, 10 fills after 3 warm-ups)."""
    for _ in range(3):
        out.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out.fill_(1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return out.numel() * out.element_size() / (ms * 1e-3) / 1e9


def end_to_end(api, device, C):
    """Host to host: pinned batches of [C][4][262144] through hrfd_ingest_* -- H2D, the kernels and
    the D2H of the PCM on three streams, two batches in flight.  Never the headline value."""
    rx = api.Rx(C, device=device.index)
    rx.set_mode(api.WBFM)
    B = 4
    ing = api.Ingest(rx, BLOCK, B, 2)
    x = make_fm_batch(C, B, device).cpu().numpy()
    for _ in range(2):
        ing.acquire()[...] = x
        ing.submit(0)
    for _ in range(2):
        ing.collect()
    steps = 6
    t0 = time.perf_counter()
    ing.acquire(); ing.submit(0)
    for i in range(steps):
        if i + 1 < steps:
            ing.acquire(); ing.submit(0)
        ing.collect()
    dt = time.perf_counter() - t0
    out = {"MSamples/s": round(C * B * (BLOCK // 2) * steps / dt / 1e6, 1), "pcie_GBps": round(C * B * BLOCK * steps / dt / 1e9, 2),
           "batch": f"{C} channels x {B} blocks from pinned host memory, 2 batches in flight", "replayed_batches": ing.replayed()}
    ing.close()
    return out


def single_block_latency_ms(api, device):
    """The reference's real cadence: ONE channel, ONE 262144-byte block (64 ms of signal) per call, host buffer in,
    PCM out, through hrfd_rx_process_block (what the IqDataProcessor shim's acceptIqData does).  Median of 30."""
    from hackrfdiags_amd import synth
    rx = api.Rx(1, device=device.index)
    rx.set_mode(api.WBFM)
    x = synth.make_input("fmtone", 0, 4).reshape(1, 4, BLOCK)
    ts = []
    for i in range(34):
        t0 = time.perf_counter()
        rx.process_block(x[:, i % 4:i % 4 + 1], 1)
        ts.append(time.perf_counter() - t0)
    return round(1e3 * float(np.median(ts[4:])), 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps; the clock governor needs ~25 ms of this load to settle (DESIGN.md 5)")
    ap.add_argument("--channels", type=int, default=0,
                    help="channels per GPU; default: 256 on one GPU (BASELINE config 2), 512 per GPU on several "
                         "(config 4: 4096 channels over 8 GPUs), 1024 for the modulator workloads (config 5)")
    ap.add_argument("--blocks", type=int, default=16, help="262144-byte blocks per channel per step")
    ap.add_argument("--signal", choices=["fmtone", "random"], default="fmtone")
    ap.add_argument("--workload", choices=["wbfm", "mixed", "am", "fm", "ssb", "ssbmod", "ammod", "fmmod", "wbfmmod", "ingest"], default="wbfm",
                    help="wbfm = BASELINE config 2 (the headline); mixed = config 3 (AM+FM+WBFM+SSB bank, "
                         "per-mode dispatch); ssbmod = config 5 (SSB modulator, 8-stage x256 interpolator)")
    ap.add_argument("--scatter", action="store_true",
                    help="N > 1 only: all IQ starts on rank 0 and is scattered over RCCL inside the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="leave out the end-to-end and single-block latency figures (profiling runs: only the headline "
                         "launches in the kernel statistics)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.channels <= 0:
        args.channels = 1024 if args.workload in ("ssbmod", "ammod", "fmmod", "wbfmmod") else (256 if world == 1 else 512)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libhrfd has no CPU path)")
    # HRFD_BENCH_REHEARSE=1: N ranks on however many GPUs there are, over gloo -- a dry run of the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device).  Never set by the driver; the line it prints says so.
    rehearse = os.environ.get("HRFD_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)    # nccl == RCCL on ROCm

    from hackrfdiags_amd import api, shard

    if args.workload in ("ssbmod", "ammod", "fmmod", "wbfmmod"):
        return bench_ssbmod(args, api, device, rank, world, dist)
    if args.workload == "ingest":
        return bench_ingest(args, api, device, rank, world, dist)

    C, B = args.channels, args.blocks
    gen = make_fm_batch if args.signal == "fmtone" else make_random_batch
    iq = gen(C, B, device, first_channel=rank * C)
    pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=device)
    n_pcm = torch.zeros((C, B), dtype=torch.int32, device=device)
    torch.cuda.synchronize()                             # inputs and outputs were written on torch's stream
    rx = api.Rx(C, device=local_rank)
    rx.set_mode(api.WBFM)
    if args.workload in ("am", "fm", "ssb"):
        rx.set_mode({"am": api.AM, "fm": api.FM, "ssb": api.LSB}[args.workload])     # a bank of one of the other modes
    if args.workload == "mixed":
        # BASELINE config 3: equal quarters of AM, FM, WBFM and SSB channels
        for c in range(C):
            rx.set_mode([api.AM, api.FM, api.WBFM, api.LSB][(4 * c) // C], channel=c)
    stream = torch.cuda.Stream(device=device)
    iq_root = None
    scatter = args.scatter and world > 1
    if scatter and rank == 0:
        # the north star's "per-channel scatter": every rank's IQ starts on rank 0, in ONE source buffer built once
        iq_root = torch.cat([gen(C, B, device, first_channel=r * C) for r in range(world)], dim=0).contiguous()
    rx.debug_enable_timing(max(args.steps, 1))

    def step():
        if scatter:
            with torch.cuda.stream(stream):
                shard.scatter_iq(iq_root, iq, world * C)     # one group of sends out of rank 0, straight into `iq`
        rx.process_device(iq.data_ptr(), B * BLOCK, BLOCK, B, pcm.data_ptr(), d_n_pcm=n_pcm.data_ptr(),
                          stream=stream.cuda_stream)

    # The clock governor needs ~25 ms of this load before it holds its clock (profiles/README.md): when
    # the caller asks for a short warm-up, untimed "settle" launches come first so that the K timed
    # steps are measured at the clock a long-running stream sees, whatever W is.
    settle = max(0, SETTLE_STEPS - args.warmup)
    for _ in range(settle + args.warmup):
        step()
    rx.sync()
    rx.debug_enable_timing(max(args.steps, 1))          # restart the slot counter

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    stream.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, device)

    rx.sync()
    counters = rx.debug_counters()
    kernel_ms = [rx.debug_kernel_ms(i) for i in range(args.steps)]
    produced = int(n_pcm.sum().item())
    assert produced == C * B * 512, f"PCM samples produced {produced} != {C * B * 512}"

    traffic, ptag = pmc_traffic_bytes(args, C, B)
    copy_gbs = stream_copy_gbs(device)
    samples_per_step = C * B * (BLOCK // 2)
    value = world * samples_per_step * args.steps / elapsed / 1e6            # MSamples/s, whole job
    algo_bytes = C * B * (BLOCK + 1024 + 4)                                   # 2.0078 B / IQ sample
    mean_ms = float(np.mean(kernel_ms))
    achieved = algo_bytes / (mean_ms * 1e-3) / 1e9

    if rank == 0:
        line = {
            "metric": "IQ MSamples/s demodulated (2.048 MS/s->8 kS/s WBFM) per GPU; % HBM roofline",
            "value": round(value, 1),
            "unit": "MSamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8 IQ -> Q15 int16/int32 + f32 recurrence -> int16 PCM",
            "data": "synthetic",
            "config": {
                "workload": (f"{C} concurrent WBFM channels per GPU at 2.048 MS/s (BASELINE config 2), "
                             if args.workload == "wbfm" else
                             f"{C} concurrent {args.workload.upper()} demodulator channels per GPU (not a BASELINE config), "
                             if args.workload in ("am", "fm", "ssb") else
                             f"mixed-mode bank {C // 4} AM + {C // 4} FM + {C // 4} WBFM + {C // 4} SSB per GPU "
                             f"(BASELINE config 3), ") +
                            (f"= {world * C} channels over {world} GPUs (BASELINE config 4 asks for 4096 over 8), "
                             if world > 1 and args.workload == "wbfm" else "") +
                            f"{B} blocks of 262144 B per channel per step, input resident in HBM"
                            + (", IQ scattered from rank 0 over RCCL each step" if args.scatter and world > 1 else ""),
                "channels_per_gpu": C, "blocks_per_step": B, "signal": args.signal,
                "parallelism": f"channels sharded, {world} rank(s), no data-path collective",
            },
            "per_gpu_value": round(value / world, 1),
            "realtime_channels_per_gpu": int(value / world / 2.048),
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": (f"profiles/latest_pmc_traffic_wbfm256x16.json, kernel sources {ptag}" if traffic is not None else
                                   ("no PMC summary for this workload" if ptag is None else
                                    f"PMC summary is of kernel sources {ptag}, this run is {kernel_source_tag()}: not reported")),
                "measured_stream_copy_GBps": round(copy_gbs, 1),
                "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                "kernel": ("hrfd::k_rx_wbfm_flow<4> (one persistent workgroup per CU, LDS ring, first-octant-table atan2)"
                           if args.workload == "wbfm" else "all demodulator kernels of a step: first kernel's start to the later of the two streams' last kernel end (HIP events on both)"),
                "kernel_ms_mean": round(mean_ms, 4),
                "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                "kernel_ms_median": round(float(np.median(kernel_ms)), 4),
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "verification": {"uncommitted_launches": counters[5], "tiles_repaired_in_place": counters[4],
                             "launches": counters[6]},
        }
        if counters[5] != 0:
            # a launch that did not commit means later launches started from a stale state and the batch path was
            # not what ran: the number is not a measurement of it
            line["invalid"] = f"{counters[5]} launch(es) were not committed"
        if world == 1 and args.workload == "wbfm" and not args.no_extras:
            line["end_to_end"] = end_to_end(api, device, C)
            line["single_block_latency_ms"] = single_block_latency_ms(api, device)
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, os.cpu_count() or 1)
            # also the reference on ONE host thread (config 1's shape), a short sample
            one = cpu_baseline(min(3.0, args.cpu_seconds), 1)
            line["cpu_baseline"]["single_thread_value"] = one["value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
