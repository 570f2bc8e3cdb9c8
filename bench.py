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
# HIP maps a process's streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4) plus one per stream with a
# CU mask (the WBFM modulator holds two).  With six or more queues alive the cross-queue event hops of the sliced WBFM
# modulator take ~50 us instead of ~10 (measured: 5.2 ms per step against 4.7 in the same process);
# nothing here needs more than two.  Read by the runtime when it starts, reported in the line as `runtime`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
if "--serial-modes" in sys.argv or any(a.startswith("--force-") for a in sys.argv):
    os.environ.setdefault("HRFD_DEBUG_HOOKS", "1")      # these flags use test hooks of include/hrfd_debug.h (inert otherwise)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BLOCK = 262144
# untimed launches in front of the timed region, at least (clock settling); HRFD_BENCH_SETTLE=0 for counter passes,
# where every dispatch is serialized by the profiler and the clock is not what is measured
# (round 5: 160, was 100 -- profiles/r5_clock_ramp.txt: consecutive regions of 20 launches from a cold start run at 0.253,
#  0.251, 0.227, 0.215, 0.210, 0.207 and from the seventh on 0.2057 +- 0.0007 ms per launch: the governor takes ~140
#  launches, 30 ms of this load, and a timed region that starts at launch 100 still carries 0.7 % of the ramp)
SETTLE_STEPS = int(os.environ.get("HRFD_BENCH_SETTLE", "160"))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


from hackrfdiags_amd.synth_torch import make_fm_batch, make_random_batch  # noqa: E402


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited / unknown"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def host_core_counts():
    """(hardware threads this process may run on, physical cores among them, CPU quota of the container or None)
    from the affinity mask, /sys topology and the cgroup; physical == threads when the topology cannot be read."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    quota = cgroup_cpu_quota()
    cores = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id") as f:
                core = f.read().strip()
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id") as f:
                pkg = f.read().strip()
            cores.add((pkg, core))
        except OSError:
            return len(cpus), len(cpus), quota
    return len(cpus), len(cores), quota


def cpu_baseline(seconds):
    """The reference's own CPU chain (IqDataProcessor::acceptIqData in WBFM mode, oracle/_ref compiled from the
    reference sources), one channel per host thread as DataConsumer.cc:319-351 runs it: std::threads inside the
    harness, one IqDataProcessor + demodulators per thread, no allocation and no interpreter in the loop
    (ref_bench_rx).  One run on every hardware thread of the box, one on ONE thread (BASELINE config 1's shape)."""
    from hackrfdiags_amd import synth
    from tests import reflib
    nb = 8
    x = synth.make_input("fmtone", 0, nb).reshape(nb, BLOCK)
    hw_threads, physical, quota = host_core_counts()
    # one thread per CPU the box actually grants: a container with a CPU quota below its visible CPUs (a one-GPU box:
    # 16 of 256) only thrashes when every visible hardware thread gets a busy thread
    threads = hw_threads if quota is None else max(1, min(hw_threads, int(quota)))
    cores_granted = min(physical, threads)
    try:
        eng = reflib.Ref()
    except (FileNotFoundError, OSError):
        return cpu_baseline_port(seconds, threads, x)
    n1, dt1, _ = eng.bench_rx(reflib.WBFM, 1, min(3.0, seconds), x)
    single = n1 * (BLOCK // 2) / dt1 / 1e6
    n, dt, pcm = eng.bench_rx(reflib.WBFM, threads, seconds, x)
    assert pcm == n * 512, "the reference chain did not produce 512 PCM samples per block"
    value = n * (BLOCK // 2) / dt / 1e6
    return {
        "value": round(value, 2),
        "unit": "MSamples/s",
        "cores": threads,
        "threads": threads,
        "physical_cores": physical,
        "hardware_threads_visible": hw_threads,
        "cgroup_cpu_quota": quota,
        "kind": "reference",
        "single_thread_value": round(single, 2),
        "scaling_vs_single_thread": round(value / single, 1),
        "efficiency_vs_cores_used": round(value / (single * cores_granted), 3),
        "sample": f"{n} blocks of 262144 B (FM test signal, WBFM mode) over {dt:.1f} s on {threads} std::thread(s), "
                  f"one IqDataProcessor + demodulators per thread; single thread: {n1} blocks over {dt1:.1f} s",
    }


def cpu_baseline_port(seconds, threads, x):
    """Fallback when the prebuilt reference did not travel: our CPU restatement (oracle/, plain C, it allocates per
    call), one channel per Python thread (ctypes releases the GIL inside the call)."""
    from tests import reflib
    eng = reflib.Oracle()
    nb = x.shape[0]
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
        "value": round(blocks * (BLOCK // 2) / dt / 1e6, 2), "unit": "MSamples/s", "cores": threads, "kind": "port",
        "sample": f"{blocks} blocks of 262144 B (FM test signal, WBFM mode) over {dt:.1f} s, {threads} Python "
                  f"thread(s) around the C restatement, one channel per thread",
    }


def spin_until_done(event, limit_s=120.0):
    """Polls the event that closes the timed region until the GPU has passed it, so that the synchronize calls behind it
    (the contract's bracket) return at once: a blocking wait adds the host's wake-up latency -- tens of microseconds, ~1 %
    of a 20-step region of 0.21 ms steps -- to every region, and that is the host's scheduler, not the path measured
    (round 6; `ms_per_step` is still wall clock between the two synchronized points)."""
    t_end = time.perf_counter() + limit_s
    while not event.query():
        if time.perf_counter() > t_end:
            break


MOD_KINDS = {"ssbmod": ("MOD_SSB", "SSB"), "ammod": ("MOD_AM", "AM"), "fmmod": ("MOD_FM", "FM"), "wbfmmod": ("MOD_WBFM", "WBFM")}
MOD_KERNELS = {"ssbmod": "hrfd::k_mod<1>",
               "wbfmmod": "hrfd::k_mod<101> (x32 + Nco step), k_phase_rows8 (the serial Nco recurrence, nine time slices), hrfd::k_wb_tail (Nco lookup + x8)"}


def measure_mod(api, shard, device, dist, workload, C, B, steps, warmup, settle, rank, world, extras=True):
    """`C` modulators of one kind, `B` blocks of 512 PCM samples (64 ms) each per step -> int8 IQ at 2.048 MS/s
    (BASELINE config 5 for 1024 SSB channels).  Unit of work: one output IQ sample."""
    n = 512 * B
    gen = torch.Generator(device=device)
    gen.manual_seed(7 + rank)
    pcm = torch.randint(-32768, 32768, (C, n), dtype=torch.int16, device=device, generator=gen)
    out = torch.empty((C, 512 * n), dtype=torch.int8, device=device)
    torch.cuda.synchronize()                             # the PCM was generated on torch's stream
    kind, kname = MOD_KINDS[workload]
    m = api.Mod(getattr(api, kind), C, device=device.index)
    stream = torch.cuda.Stream(device=device)

    def step():
        m.process_device(pcm.data_ptr(), n, out.data_ptr(), stream=stream.cuda_stream)

    for _ in range(settle + warmup):
        step()
    m.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # (round 6, as on the receive side since round 5: ONE event pair on the launch stream around all calls of the timed
    #  region prices the roofline; single calls are bracketed BEHIND the region for min / median)
    ev_region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    ev_region[0].record(stream)
    for _ in range(steps):
        step()
    ev_region[1].record(stream)
    spin_until_done(ev_region[1])
    stream.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device)
    m.sync()                                             # raises if a k_phase_scan wait expired
    region_ms = ev_region[0].elapsed_time(ev_region[1]) / steps
    n_samp = min(4, steps)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_samp)]
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    stream.synchronize()
    m.sync()
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    samples = C * n * 256
    algo_bytes = C * n * (2 + 512)
    achieved = algo_bytes / (region_ms * 1e-3) / 1e9
    fill = stream_gbs(device, 1, out) if (rank == 0 and extras) else None
    if fill is not None:
        assert under_profiler() or achieved <= fill, f"the modulator writes faster ({achieved:.0f} GB/s) than a kernel that does nothing else ({fill:.0f}): a denominator is wrong"
    m.close()
    del out, pcm
    torch.cuda.empty_cache()                             # the next workload of the line starts from a clean allocator
    return {
        "kname": kname, "value": world * samples * steps / elapsed / 1e6, "ms_per_step": 1e3 * elapsed / steps,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "measured_stream_write_GBps": None if fill is None else round(fill, 1),
                     "frac_of_measured_write": None if fill is None else round(achieved / fill, 4),
                     "kernel": MOD_KERNELS.get(workload, "k_am_rails / k_fm_step + k_phase_rows + k_fm_rails, then hrfd::k_mod<100>"),
                     "region_ms_per_launch": round(region_ms, 4), "priced_with": "region_ms_per_launch (one HIP event pair around the timed region / steps)",
                     "kernel_ms_min": round(float(np.min(kernel_ms)), 4), "kernel_ms_median": round(float(np.median(kernel_ms)), 4),
                     "kernel_launches_sampled": n_samp, "algorithmic_bytes_per_launch": algo_bytes},
    }


def bench_mod(args, api, device, rank, world, dist):
    from hackrfdiags_amd import shard
    C, B = args.channels, args.blocks
    settle = max(0, SETTLE_STEPS - args.warmup)
    r = measure_mod(api, shard, device, dist, args.workload, C, B, args.steps, args.warmup, settle, rank, world,
                    extras=not args.no_extras)
    kname = r["kname"]
    t, src = pmc_traffic(f"{args.workload}_{C}x{B}")
    r["roofline"]["traffic"] = t
    r["roofline"]["traffic_source"] = src
    r["roofline"]["traffic_over_algorithmic"] = None if t is None else round(t / r["roofline"]["algorithmic_bytes_per_launch"], 4)
    if rank == 0:
        print(json.dumps({
            "metric": f"IQ MSamples/s modulated (8 kS/s PCM -> 2.048 MS/s int8 IQ, {kname}) per GPU; % HBM roofline",
            "value": round(r["value"], 1), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "settle_steps": settle, "ms_per_step": round(r["ms_per_step"], 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int16 PCM -> Q15 int32 -> int8 IQ", "data": "synthetic",
            "config": {"workload": f"{C} {kname} modulator channels per GPU (BASELINE config 5" + ("" if (C, args.workload) == (1024, "ssbmod") else
                                   " is 1024 SSB channels") + f"), {B} blocks of 512 PCM "
                                   f"samples per step, 8-stage x256 half-band interpolator", "channels_per_gpu": C,
                       "blocks_per_step": B},
            "roofline": r["roofline"],
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


def bench_fanout(args, api, device, rank, world):
    """BASELINE config 4's shape in ONE process (hrfd_fanout_*: what a C++ host that owns all the radios links against):
    `channels` WBFM channels in `shards` contiguous shards, the IQ of the whole bank resident on device 0, scattered to
    the shards' devices (peer copies; a copy on the device itself where a shard sits on the source device), every shard
    demodulated, the PCM gathered back.  With fewer devices than shards the shards share devices round robin -- on a
    one-GPU box all of them sit on device 0 and the line says so: the scatter is then device-local copies, NOT xGMI."""
    C, B, S = args.channels, args.blocks, args.shards
    n_dev = torch.cuda.device_count()
    devices = [g % n_dev for g in range(S)]
    fo = api.Fanout(C, devices)
    fo.set_mode(api.WBFM)
    base = make_fm_batch(32, B, device)                  # 32 distinct channels (the tone depends on c mod 32 anyway)
    idx = torch.arange(C, device=device) % 32
    x = base[idx]                                        # [C][B][262144] on device 0
    del base
    pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=device)
    n_pcm = torch.zeros((C, B), dtype=torch.int32, device=device)
    torch.cuda.synchronize()

    def sync_all():
        for d in sorted(set(devices) | {device.index}):
            torch.cuda.synchronize(d)

    def step(times=None):
        t0 = time.perf_counter()
        fo.scatter(device.index, x.data_ptr(), BLOCK, B)
        sync_all()
        t1 = time.perf_counter()
        fo.process(0)
        sync_all()
        t2 = time.perf_counter()
        replayed = fo.collect(device.index, pcm.data_ptr(), n_pcm.data_ptr())
        sync_all()
        t3 = time.perf_counter()
        if times is not None:
            times.append((t1 - t0, t2 - t1, t3 - t2, replayed))

    for _ in range(args.warmup):
        step()
    times = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(times)
    elapsed = time.perf_counter() - t0
    sc, pr, co = (1e3 * float(np.mean([t[i] for t in times])) for i in range(3))
    samples = C * B * (BLOCK // 2)
    algo = C * B * (BLOCK + 1024 + 4)
    produced = int(n_pcm.sum().item())
    line = {
        "metric": "IQ MSamples/s demodulated (2.048 MS/s->8 kS/s WBFM), one process over several shards (hrfd_fanout_*)",
        "value": round(samples * args.steps / elapsed / 1e6, 1), "unit": "MSamples/s", "n_gpus": len(set(devices)),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "int8 IQ -> Q15 int16/int32 + f32 recurrence -> int16 PCM", "data": "synthetic",
        "config": {"workload": f"{C} WBFM channels in {S} shards of {C // S} (BASELINE config 4's shape), {B} blocks per channel per step, "
                               f"IQ of the whole bank resident on device {device.index}; shards on devices {devices}"
                               + ("" if len(set(devices)) == S else
                                  f" -- {len(set(devices))} device(s) for {S} shards: the scatter is device-local copies here, "
                                  "NOT xGMI; unmeasured over xGMI until an 8-GPU node runs this"),
                   "channels": C, "shards": S, "blocks_per_step": B},
        "phases_ms": {"scatter": round(sc, 4), "process": round(pr, 4), "collect": round(co, 4),
                      "note": "host clock, every phase closed with a device synchronize; in a pipeline the scatter of batch "
                              "k+1 runs beside the kernels of batch k (separate streams per shard)"},
        "scatter_GBps": round(C * B * BLOCK / (sc * 1e-3) / 1e9, 1),
        "roofline": {"bound": "hbm", "achieved": round(algo / (pr * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS * len(set(devices)),
                     "unit": "GB/s", "frac": round(algo / (pr * 1e-3) / 1e9 / (HBM_PEAK_GBS * len(set(devices))), 4), "traffic": None,
                     "kernel": "hrfd::k_rx_wbfm_flow<4, false, false, WBFM>, one launch per shard (process phase, host clock)"},
        "verification": {"channels_replayed": int(sum(t[3] for t in times)), "pcm_samples": produced, "pcm_expected": C * B * 512},
    }
    if produced != C * B * 512 or line["verification"]["channels_replayed"]:
        line["invalid"] = "PCM count or replays"
    print(json.dumps(line), flush=True)
    fo.close()


def kernel_code_tag():
    """sha256 (first 16 hex digits) of the DEVICE CODE libhrfd.so carries -- the .hip_fatbin section, i.e. the compiled
    gfx950 code objects -- so that a committed PMC summary is tied to the code it measured and to nothing else (until
    round 4 the tag hashed the sources: a comment edit invalidated every summary)."""
    import hashlib
    import struct
    from hackrfdiags_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        elf = f.read()
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "libhrfd.so: not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    def sec(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = sec(shstrndx)
    names = elf[stroff:stroff + strsize]
    for i in range(shnum):
        name, off, size = sec(i)
        if names[name:names.index(b"\0", name)] == b".hip_fatbin":
            return hashlib.sha256(elf[off:off + size]).hexdigest()[:16]
    raise RuntimeError("libhrfd.so has no .hip_fatbin section")


def kernel_source_tag():
    """(kept as a second field of the line: sha256 of the kernel sources, comments included)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hackrfdiags_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


_PMC = None


def pmc_traffic(name):
    """(HBM bytes per launch, note) of workload `name` from the PMC passes of tools/pmc_round.sh (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, each in a run of its own of this same bench command; the summary is committed as
    profiles/latest_pmc_traffic.json).  gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at
    64 bytes -- doubled here, as MI355X_MICROARCH.md prescribes; WRITE_SIZE is exact for 16-byte-per-lane streaming
    stores.  Both are in KiB.  Reported only when the summary was taken with THIS device code (kernel_code_tag)."""
    global _PMC
    if _PMC is None:
        try:
            with open(os.path.join(ROOT, "profiles", "latest_pmc_traffic.json")) as f:
                _PMC = json.load(f)
        except (OSError, ValueError):
            _PMC = {}
    if not _PMC:
        return None, "no PMC summary"
    if _PMC.get("kernel_code_tag") != kernel_code_tag():
        return None, f"the PMC summary is of device code {_PMC.get('kernel_code_tag')}, this run is {kernel_code_tag()}: not reported"
    w = _PMC.get("workloads", {}).get(name)
    if not w:
        return None, "no PMC summary for this workload"
    return int(w["FETCH_SIZE_KiB"] * 1024 * 2 + w["WRITE_SIZE_KiB"] * 1024), f"profiles/latest_pmc_traffic.json[{name}], device code {_PMC['kernel_code_tag']}"


def issue_summary():
    """what the headline kernel's SIMDs were doing (tools/valu_probe.sh: rocprofv3's derived VALUBusy / SALUBusy /
    MemUnitStalled and raw SQ counters, PMC-only passes; committed as profiles/latest_valu_probe.json) -- quoted only when it
    was taken with THIS device code.  The contract prices the kernel against HBM; these say what it is bound by."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_valu_probe.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if d.get("kernel_code_tag") != kernel_code_tag():
        return None
    c = d.get("counters", {})
    return {"valu_busy_pct": c.get("VALUBusy"), "valu_lane_utilization_pct": c.get("VALUUtilization"), "salu_busy_pct": c.get("SALUBusy"),
            "mem_unit_stalled_pct": c.get("MemUnitStalled"), "lds_bank_conflict_pct": c.get("LDSBankConflict"),
            "valu_instructions_per_launch": c.get("SQ_INSTS_VALU"), "source": "profiles/latest_valu_probe.json (" + d.get("how", "") + ")"}


def under_profiler():
    """rocprofv3 preloads its tool library into the process: counter passes serialize every dispatch and change the
    clocks, so timings taken there are not comparable with each other (a 32768-workgroup stream kernel suffers more than a
    256-workgroup persistent one) -- the bandwidth assertion below is for plain runs."""
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("HSA_TOOLS_LIB", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    return "rocprof" in pre.lower() or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ)


def stream_gbs(device, kind, buf=None):
    """The measured denominators: what a kernel that does nothing but READ (kind 0) or nothing but WRITE
    (kind 1) 1 GiB reaches on this GPU in this run -- libhrfd's own two plain stream kernels (csrc/hrfd_membw.hip:
    16 bytes per lane, 32 KiB per workgroup, every XCD one contiguous eighth), HIP events on the launch stream, 10
    launches after 3 warm-ups.  (Until round 4 this was a torch copy_, i.e. read + write, compared with a read-only
    kernel: a fraction above 1 said the denominator was wrong.)"""
    import ctypes
    from hackrfdiags_amd import _lib
    L = _lib.load()
    n = 1 << 30
    own = buf is None
    if own:
        buf = torch.empty(n, dtype=torch.int8, device=device)
        buf.random_(0, 127)
    nbytes = (buf.numel() * buf.element_size()) // (1 << 18) * (1 << 18)
    st = torch.cuda.current_stream(device)
    torch.cuda.synchronize()
    for _ in range(3):
        assert L.hrfd_debug_membw(kind, ctypes.c_void_p(buf.data_ptr()), nbytes, None, ctypes.c_void_p(st.cuda_stream)) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        L.hrfd_debug_membw(kind, ctypes.c_void_p(buf.data_ptr()), nbytes, None, ctypes.c_void_p(st.cuda_stream))
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    if own:
        del buf
    return nbytes / (ms * 1e-3) / 1e9


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


def realtime_cadence(api, device, C, kind="wbfm", batches=48):
    """The north star's target as stated: C concurrent channels IN REAL TIME.  The reference's cadence is one 262144-byte
    block per channel per 64 ms (hackRf/hackrf.c:100-101 -> DataConsumer::acceptData, DataConsumer.cc:219-262), so a
    batch is ONE block of every channel, handed over in pinned host memory (hrfd_ingest_*: H2D, the per-block kernels,
    D2H of PCM / magnitude / gate), PCM back on the host, and it must be through before the next one arrives.
    Latency = submit -> results on the host, one batch in flight.  `paced`: a batch every 64 ms, i.e. the GPU idles
    in between and every batch meets the cold clock (what a real-time host lives in); `unpaced`: back to back."""
    rx = api.Rx(C, device=device.index)
    if kind == "mixed":
        for c in range(C):
            rx.set_mode([api.AM, api.FM, api.WBFM, api.LSB][(4 * c) // C], channel=c)
    else:
        rx.set_mode(api.WBFM)
    ing = api.Ingest(rx, BLOCK, 1, 2)
    x = make_fm_batch(C, 2, device).cpu().numpy()        # two consecutive blocks of every channel
    torch.cuda.synchronize()
    fill = []
    for k in range(2):                                   # both pinned slots hold valid IQ from here on
        slot = ing.acquire()
        t0 = time.perf_counter()
        slot[...] = x[:, k:k + 1]
        fill.append(time.perf_counter() - t0)
        ing.submit(0)
        ing.collect()
    for _ in range(4):
        ing.acquire(); ing.submit(0); ing.collect()

    def run(paced):
        lat, produced = [], 0
        t_next = time.perf_counter()
        for _ in range(batches):
            if paced:
                t_next += 0.064
                while time.perf_counter() < t_next:
                    time.sleep(0.0005)
            ing.acquire()
            t0 = time.perf_counter()
            ing.submit(0)
            out = ing.collect()
            lat.append(time.perf_counter() - t0)
            produced += int(out[1].sum())
        lat_ms = 1e3 * np.sort(np.array(lat))
        p50, p99, worst = float(np.percentile(lat_ms, 50)), float(np.percentile(lat_ms, 99)), float(lat_ms[-1])
        return {"batches": batches, "p50_ms": round(p50, 3), "p99_ms": round(p99, 3), "max_ms": round(worst, 3),
                "budget_used_p99": round(p99 / 64.0, 4), "realtime_headroom": round(64.0 / p99, 1),
                "host_to_host_GBps_p50": round(C * BLOCK / (p50 * 1e-3) / 1e9, 1),
                "pcm_samples": produced, "pcm_expected": batches * C * 512}

    out = {"workload": (f"{C} concurrent {'WBFM' if kind == 'wbfm' else 'AM/FM/WBFM/SSB (a quarter each)'} channels, ONE 262144-byte "
                        f"block (64 ms of signal) per channel per batch = {C * BLOCK / 2**20:.0f} MiB from pinned host memory through "
                        f"hrfd_ingest_*, PCM back on the host; latency = submit -> results, one batch in flight"),
           "paced_64ms": run(True), "unpaced": run(False),
           "producer_fill_ms": round(1e3 * float(np.mean(fill)), 2),
           "producer_fill_note": "one host thread copying the batch into the pinned slot (numpy); DataConsumer::acceptData's memcpy, "
                                 "spread over the receive threads in a real host",
           "replayed_batches": ing.replayed()}
    for k in ("paced_64ms", "unpaced"):
        if out[k]["pcm_samples"] != out[k]["pcm_expected"]:
            out["invalid"] = f"{k}: PCM samples {out[k]['pcm_samples']} != {out[k]['pcm_expected']}"
    ing.close()
    rx.close()
    torch.cuda.empty_cache()
    return out


def host_replay_cost(api, device):
    """A batch of more than 64 blocks with closed squelch gates is the one case the device does not repair by itself
    (the gated pass holds a run's block list in 64 LDS words): hrfd_rx_process_block replays the channels concerned on
    the host's side, block by block through the exact per-block kernel.  64 WBFM channels x 80 blocks from host memory
    through the blocking entry, a quarter of the channels silent: with the threshold at its default (no gate can close)
    and at -30 dBFS (their gates close)."""
    C, B = 64, 80
    x = make_fm_batch(C, B, device)
    x[::4] = make_quiet_batch(C // 4, B, device)
    xh = x.cpu().numpy()
    del x
    res = {}
    for name, thr in (("gates_open", None), ("gates_closing", -30)):
        rx = api.Rx(C, device=device.index)
        rx.set_mode(api.WBFM)
        if thr is not None:
            rx.set_threshold(thr)
        rx.process_block(xh, B)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = rx.process_block(xh, B)
            ts.append(time.perf_counter() - t0)
        res[name + "_ms"] = round(1e3 * min(ts), 2)
        res[name + "_pcm_blocks"] = int((out[1] > 0).sum())
        rx.close()
    res["workload"] = (f"{C} WBFM channels x {B} blocks (> 64) through hrfd_rx_process_block from pageable host memory "
                       f"({C * B * BLOCK / 2**30:.2f} GiB per call), a quarter of the channels silent; best of 3 calls")
    res["replay_cost_ms"] = round(res["gates_closing_ms"] - res["gates_open_ms"], 2)
    torch.cuda.empty_cache()
    return res


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


def make_quiet_batch(channels, blocks, device, first_channel=0):
    """A channel with no signal: uniform noise in [-1, 1] (block-mean magnitude 1 -> about -42 dBFS)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(99 + first_channel)
    return torch.randint(-1, 2, (channels, blocks, BLOCK), dtype=torch.int8, device=device, generator=gen)


def rx_workload_name(workload, C, B, signal="fmtone", quiet_fraction=0.0, iqdump=False):
    """the key of a workload in the line's `also` object and in profiles/latest_pmc_traffic.json"""
    name = f"{workload}_{C}x{B}"
    if signal != "fmtone":
        name += "_" + signal
    if quiet_fraction > 0:
        name += f"_quiet{int(round(100 * quiet_fraction))}"
    if iqdump:
        name += "_iqdump"
    return name


def rx_workload_text(workload, C, B, world, signal, scatter, quiet_fraction=0.0, threshold=None, iqdump=False):
    if workload == "wbfm":
        if world > 1 and world * C == 4096:
            head = f"{world * C} WBFM channels sharded over {world} GPUs, {C} per GPU (BASELINE config 4), "
        elif world > 1:
            head = (f"{C} concurrent WBFM channels per GPU at 2.048 MS/s = {world * C} channels over {world} GPUs "
                    f"(WEAK scaling of BASELINE config 2's per-GPU load: the same {C} channels per GPU at every N; config 4 is in `config4`), ")
        elif C == 256:
            head = f"{C} concurrent WBFM channels per GPU at 2.048 MS/s (BASELINE config 2), "
        else:
            head = f"{C} concurrent WBFM channels per GPU at 2.048 MS/s (BASELINE config 2 is 256; the north star's target is >= 1000), "
    elif workload in ("am", "fm", "ssb"):
        head = f"{C} concurrent {workload.upper()} demodulator channels per GPU (not a BASELINE config), "
    else:
        head = f"mixed-mode bank {C // 4} AM + {C // 4} FM + {C // 4} WBFM + {C // 4} SSB per GPU (BASELINE config 3), "
    tail = f"{B} blocks of 262144 B per channel per step, input resident in HBM"
    if quiet_fraction > 0:
        tail += f", {quiet_fraction:.0%} of the channels carry no signal under a squelch threshold of {threshold} dBFS (their gates close)"
    if iqdump:
        tail += ", the 256 kS/s stream of `enable iqdump` written out as well"
    if scatter and world > 1:
        tail += ", IQ scattered from rank 0 over RCCL each step"
    return head + tail


def verify_against_oracle(iq, pcm, modes, launches, B, n, threshold, seed=0):
    """`--verify` (on by default for the receive workloads): AFTER the timed region, `n` channels of the bench's own batch
    -- channel 0, the last one and seeded-random ones from the whole range -- go through the sequential CPU oracle
    (tests/reflib.Oracle: the checker, never the thing measured), from a fresh state through every launch the handle has
    seen (settle + warm-up + timed steps + the single launches sampled behind the region, all over the same resident batch,
    the streams continuing), and the PCM the handle's LAST launch left in the output buffer -- the last of the sampled
    launches behind the timed region -- must be the oracle's, sample for sample.  One Python thread per channel (ctypes
    releases the GIL inside the oracle).  With N > 1 only rank 0 verifies (seconds; the other ranks wait for it in the
    first collective of multi_gpu_legs)."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import reflib
    orc = reflib.Oracle()
    C = iq.shape[0]
    rng = np.random.default_rng(seed)
    sel = {0, C - 1}
    while len(sel) < min(n, C):
        sel.add(int(rng.integers(0, C)))
    sel = sorted(sel)
    t0 = time.perf_counter()
    tsel = torch.tensor(sel, device=iq.device)
    x = iq[tsel][:, :B * BLOCK].reshape(len(sel), B, BLOCK).cpu().numpy()
    got = pcm[tsel].cpu().numpy()

    def work(i):
        o = orc.rx()
        o.set_mode(reflib.WBFM if modes is None else modes[sel[i]])
        if threshold is not None:
            o.set_threshold(int(threshold))
        bad = 0
        for launch in range(launches):
            for b in range(B):
                p = o.process(x[i, b])[0]
                if launch == launches - 1:
                    bad += int(len(p) != 512 or not (got[i, b] == p).all())
        return bad

    with ThreadPoolExecutor(max_workers=min(16, len(sel))) as ex:
        bad = list(ex.map(work, range(len(sel))))
    res = {"oracle_channels_checked": len(sel), "channels": sel, "launches_replayed_by_the_oracle": launches,
           "pcm_blocks_compared": len(sel) * B, "pcm_blocks_mismatching": int(sum(bad)), "tolerance_lsb": 0,
           "seconds": round(time.perf_counter() - t0, 1)}
    assert sum(bad) == 0, f"bench --verify: PCM of channels {[c for c, k in zip(sel, bad) if k]} differs from the oracle"
    return res


def measure_rx(api, shard, device, dist, *, workload, C, B, signal, steps, warmup, settle, rank, world,
               scatter=False, quiet_fraction=0.0, threshold=None, iqdump=False, idle_s=0.0, serial_modes=False, stride_pad=0,
               blk=BLOCK, verify=0):
    """K timed steps of the receive path over one resident batch [C][B][262144]; returns the figures of a bench line.
    The dominant kernels' time comes from HIP events the library records on its launch stream(s) around the
    demodulator kernels of every launch (hrfd_rx_debug_enable_timing)."""
    gen = make_fm_batch if signal == "fmtone" else make_random_batch
    iq = gen(C, B, device, first_channel=rank * C)
    n_quiet = int(round(C * quiet_fraction))
    quiet = []
    if n_quiet:
        # every (C / n_quiet)-th channel is silent
        quiet = [int(i * C / n_quiet) for i in range(n_quiet)]
        q = make_quiet_batch(n_quiet, B, device, first_channel=rank * C)
        iq[torch.tensor(quiet, device=device)] = q
        del q
    if blk != BLOCK:                                     # a shorter block: the front of every 262144-byte block
        iq = iq[:, :, :blk].contiguous()
    stride = B * blk + stride_pad
    if stride_pad:
        padded = torch.zeros((C, stride), dtype=torch.int8, device=device)
        padded[:, :B * blk] = iq.reshape(C, B * blk)
        iq = padded
    pcm = torch.zeros((C, B, blk // 512), dtype=torch.int16, device=device)
    n_pcm = torch.zeros((C, B), dtype=torch.int32, device=device)
    iq256 = torch.zeros((C, B, blk // 8), dtype=torch.int8, device=device) if iqdump else None
    torch.cuda.synchronize()                             # inputs and outputs were written on torch's stream
    rx = api.Rx(C, device=device.index)
    rx.set_mode(api.WBFM)
    if workload in ("am", "fm", "ssb"):
        rx.set_mode({"am": api.AM, "fm": api.FM, "ssb": api.LSB}[workload])     # a bank of one of the other modes
    if workload == "mixed":
        # BASELINE config 3: equal quarters of AM, FM, WBFM and SSB channels
        for c in range(C):
            rx.set_mode([api.AM, api.FM, api.WBFM, api.LSB][(4 * c) // C], channel=c)
    if threshold is not None:
        rx.set_threshold(int(threshold))
    if serial_modes:
        rx.debug_set_fir_flow(0)
    stream = torch.cuda.Stream(device=device)
    iq_root = None
    scatter = scatter and world > 1
    if scatter and rank == 0:
        # the north star's "per-channel scatter": every rank's IQ starts on rank 0, in ONE source buffer built once
        iq_root = torch.cat([gen(C, B, device, first_channel=r * C) for r in range(world)], dim=0).contiguous()
    # Kernel time (round 5): ONE pair of HIP events on the launch stream around ALL `steps` launches of the timed region --
    # their distance / steps is the mean launch duration over every launch of the region (the kernels of consecutive
    # steps follow each other on the stream without a gap).  Until round 4 the library bracketed every launch (rounds
    # 1-3: an event record is a queue packet of ~3 us, two per launch put 6-7 us between kernels that otherwise follow
    # each other with none) or every fourth one (round 4: 5 samples of 20 steps, and the gaps of the sampled ones still
    # inside the step time).  Single launches are still sampled -- min / median -- but BEHIND the timed region
    # (`n_samp` extra launches, hrfd_rx_debug_enable_timing): no event packet stands inside the region.
    n_samp = min(8, steps)
    rx.debug_enable_timing(0)
    def step():
        if scatter:
            with torch.cuda.stream(stream):
                shard.scatter_iq(iq_root, iq, world * C)     # one group of sends out of rank 0, straight into `iq`
        rx.process_device(iq.data_ptr(), stride, blk, B, pcm.data_ptr(), d_n_pcm=n_pcm.data_ptr(),
                          d_iq256=None if iq256 is None else iq256.data_ptr(), stream=stream.cuda_stream)

    if idle_s > 0:
        torch.cuda.synchronize()
        time.sleep(idle_s)                               # the clock governor falls back to its idle state
    for _ in range(settle + warmup):
        step()
    rx.sync()

    ev_region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_region[0].record(stream)
    for _ in range(steps):
        step()
    ev_region[1].record(stream)
    spin_until_done(ev_region[1])
    stream.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device)

    rx.sync()
    region_ms = ev_region[0].elapsed_time(ev_region[1]) / steps
    rx.debug_enable_timing(n_samp)
    rx.debug_timing_every(1)
    for _ in range(n_samp):
        step()
    stream.synchronize()
    rx.sync()
    counters = rx.debug_counters()
    kernel_ms = [rx.debug_kernel_ms(i) for i in range(n_samp)]
    produced = int(n_pcm.sum().item())
    # a closed gate produces no PCM (the tracker lets one "tail" block through after a signal: none here, the quiet
    # channels are quiet from the start)
    expect = (C - n_quiet) * B * (blk // 512)
    samples_per_step = C * B * (blk // 2)
    algo_bytes = C * B * (blk + blk // 256 + 4) + (C * B * (blk // 8) if iqdump else 0)   # 2.0078 B / IQ sample
    # the launch duration the roofline is priced with: all `steps` launches of the timed region (with the scatter on the
    # stream the region holds the copies too: then the sampled kernels alone)
    mean_ms = float(np.mean(kernel_ms)) if scatter else region_ms
    achieved = algo_bytes / (mean_ms * 1e-3) / 1e9
    out = {
        "value": world * samples_per_step * steps / elapsed / 1e6,            # MSamples/s, whole job
        "ms_per_step": 1e3 * elapsed / steps, "mean_ms": mean_ms, "achieved": achieved,
        "kernel_ms_min": float(np.min(kernel_ms)), "kernel_ms_median": float(np.median(kernel_ms)),
        "kernel_ms_sampled_mean": float(np.mean(kernel_ms)), "launches_sampled": n_samp,
        "algo_bytes": algo_bytes, "counters": counters, "quiet_channels": n_quiet, "launches_timed": steps,
        "pcm_produced": produced, "pcm_expected": expect,
    }
    if verify and not (iqdump or n_quiet or stride_pad or blk != BLOCK or scatter):
        modes = None
        if workload == "mixed":
            modes = [[api.AM, api.FM, api.WBFM, api.LSB][(4 * c) // C] for c in range(C)]
        elif workload in ("am", "fm", "ssb"):
            modes = [{"am": api.AM, "fm": api.FM, "ssb": api.LSB}[workload]] * C
        try:
            out["verification"] = verify_against_oracle(iq, pcm, modes, settle + warmup + steps + n_samp, B, verify, threshold, seed=C + steps)
        except AssertionError:
            raise                                        # a PCM mismatch: no line
        except Exception as e:                           # noqa: BLE001  (the checker broke, not the library: the measurements stand, the line says so)
            out["verification"] = {"oracle_channels_checked": 0, "error": repr(e)[:200]}
    rx.close()
    del iq, pcm, n_pcm, iq256, iq_root
    torch.cuda.empty_cache()                             # the next workload of the line starts from a clean allocator
    return out


def brief(r, text, extra=None):
    """one entry of the verbose `also` object (--extras-verbose)"""
    d = {"workload": text, "ms_per_step": round(r["ms_per_step"], 4), "value_MSamples_per_s": round(r["value"], 1)}
    if "roofline" in r:                                  # modulators
        d["region_ms_per_launch"] = r["roofline"]["region_ms_per_launch"]
        d["kernel_ms_min"] = r["roofline"]["kernel_ms_min"]
        d["roofline_frac"] = r["roofline"]["frac"]
        d["algorithmic_bytes_per_launch"] = r["roofline"]["algorithmic_bytes_per_launch"]
    else:
        d["region_ms_per_launch"] = round(r["mean_ms"], 4)
        d["kernel_ms_min"] = round(r["kernel_ms_min"], 4)
        d["roofline_frac"] = round(r["achieved"] / HBM_PEAK_GBS, 4)
        d["algorithmic_bytes_per_launch"] = r["algo_bytes"]
        d["uncommitted_launches"] = r["counters"][5]
        if r["counters"][5] != 0:
            d["invalid"] = f"{r['counters'][5]} launch(es) were not committed"
        elif r["pcm_produced"] != r["pcm_expected"]:
            d["invalid"] = f"PCM samples produced {r['pcm_produced']} != {r['pcm_expected']}"
    if extra:
        d.update(extra)
    if "name" in d:
        t, src = pmc_traffic(d.pop("name"))
        d["traffic"] = t
        d["traffic_over_algorithmic"] = None if t is None else round(t / d["algorithmic_bytes_per_launch"], 4)
        if t is None:
            d["traffic_source"] = src
    return d


def compact(d):
    """the same entry as the driver-visible line carries it (`configs`): ms per step, fraction of the 8 TB/s roofline priced
    with the region event pair, PMC traffic over algorithmic bytes -- everything else of the entry is in `also`
    (--extras-verbose) and in DESIGN.md"""
    c = {"ms": d.get("ms_per_step"), "frac": d.get("roofline_frac"), "traffic_x": d.get("traffic_over_algorithmic")}
    if d.get("region_ms_per_launch") is not None:
        c["kernel_ms"] = d["region_ms_per_launch"]
    v = d.get("verification")
    if v:
        c["oracle_channels_ok"] = v.get("oracle_channels_checked") if not v.get("error") and v.get("pcm_blocks_mismatching", 0) == 0 else 0
        if v.get("error"):
            c["verify_error"] = v["error"][:80]
    if "invalid" in d:
        c["invalid"] = d["invalid"]
    return c


def also_lines(api, shard, device, args):
    """The other BASELINE configurations that fit one GPU, the >= 1000-channel target, the worst-case input and the
    cold-clock figure, each a short measurement of its own (N = 1 only; SURVEY 8(d), BASELINE.json configs 3 and 5)."""
    K, W = 40, 10
    settle = max(0, SETTLE_STEPS - W)
    common = dict(steps=K, warmup=W, settle=settle, rank=0, world=1)
    out = {}

    def rx(name, **kw):
        text_kw = {k: kw[k] for k in ("quiet_fraction", "threshold", "iqdump") if k in kw}
        r = measure_rx(api, shard, device, None, **{**common, **kw})
        out[name] = brief(r, rx_workload_text(kw["workload"], kw["C"], kw["B"], 1, kw["signal"], False, **text_kw),
                          {"steps": kw.get("steps", K), "warmup": kw.get("warmup", W), "settle_steps": kw.get("settle", settle),
                           "name": rx_workload_name(kw["workload"], kw["C"], kw["B"], kw["signal"], kw.get("quiet_fraction", 0.0),
                                                    kw.get("iqdump", False))})
        return r

    r = rx("wbfm_1024x16", workload="wbfm", C=1024, B=16, signal="fmtone", verify=args.verify)
    out["wbfm_1024x16"]["verification"] = r.get("verification")
    r = rx("mixed_256x16", workload="mixed", C=256, B=16, signal="fmtone", verify=args.verify)
    out["mixed_256x16"]["verification"] = r.get("verification")
    # config 4's whole bank on ONE GPU (16 GiB of IQ per launch): the N = 1 point of its strong-scaling curve
    rx("wbfm_4096x16", workload="wbfm", C=4096, B=16, signal="fmtone", steps=10, warmup=3, settle=10)
    # the headline's 256 channels with 64 blocks per launch (4 GiB; SURVEY 8d: "B >= 16 ... time over >= 64 blocks"): what
    # the per-launch costs of the 16-block step -- launch hand-over, the XCDs' spread, the service tail -- are worth
    rx("wbfm_256x64", workload="wbfm", C=256, B=64, signal="fmtone", steps=20, warmup=5, settle=35)
    rx("wbfm_256x16_random", workload="wbfm", C=256, B=16, signal="random")
    # what a caller sees that launches into an idle GPU (the reference's cadence is one block per 64 ms): the driver's
    # own warm-up, no settling launches, after a second of idleness
    rx("wbfm_256x16_unsettled", workload="wbfm", C=256, B=16, signal="fmtone", steps=min(args.steps, 20),
       warmup=min(args.warmup, 5), settle=0, idle_s=1.0)
    rx("wbfm_256x16_quiet25", workload="wbfm", C=256, B=16, signal="fmtone", quiet_fraction=0.25, threshold=-30)
    rx("wbfm_256x16_iqdump", workload="wbfm", C=256, B=16, signal="fmtone", iqdump=True)
    for name, wl, C in (("ssbmod_1024x16", "ssbmod", 1024), ("ammod_1024x16", "ammod", 1024), ("fmmod_1024x16", "fmmod", 1024),
                        ("wbfmmod_1024x16", "wbfmmod", 1024)):
        r = measure_mod(api, shard, device, None, wl, C, 16, K, W, settle, 0, 1, extras=False)
        out[name] = brief(r, f"{C} {r['kname']} modulator channels, 16 blocks of 512 PCM samples per step"
                          + (" (BASELINE config 5)" if wl == "ssbmod" else ""), {"steps": K, "warmup": W, "settle_steps": settle, "name": name})
    # The WBFM modulator's phase recurrence is serial per channel and costs the same 3.2 ms for 64 or 8192 channels
    # (DESIGN.md 3.3): 1024 channels is the WORST point to quote its per-GPU throughput at.  What a chip-filling bank
    # reaches (32 GiB of IQ per step; the recurrence runs two waves per SIMD, the passes around it are what takes the time):
    r = measure_mod(api, shard, device, None, "wbfmmod", 8192, 16, 6, 2, 4, 0, 1, extras=False)
    out["wbfmmod_8192x16"] = brief(r, "8192 WBFM modulator channels, 16 blocks of 512 PCM samples per step (the largest bank "
                                      "k_phase_rows takes: 32 GiB of IQ out per step)", {"steps": 6, "warmup": 2, "settle_steps": 4, "name": "wbfmmod_8192x16"})
    # What the fallbacks cost: shapes the flow kernels do not take run on the round-1/2 block kernels
    # (one workgroup per channel-block, k_rx_finish behind them).
    r = measure_rx(api, shard, device, None, **{**common, "workload": "am", "C": 32, "B": 16, "signal": "fmtone"})
    out["fallback_am_32x16"] = brief(r, "32 AM channels x 16 blocks: a FIR-mode bank under 48 channels runs on the block kernels "
                                        "k_rx_fir<14> + k_rx_post<14> + k_rx_finish (a whole-CU workgroup per channel would leave 7/8 of the chip idle)",
                                     {"steps": K, "warmup": W, "settle_steps": settle})
    r = measure_rx(api, shard, device, None, **{**common, "workload": "wbfm", "C": 256, "B": 16, "signal": "fmtone", "blk": 258048})
    out["fallback_wbfm_256x16_blk258048"] = brief(r, "256 WBFM channels x 16 blocks of 258048 bytes (31.5 units of 8 KiB: not whole units): "
                                                     "the block kernel k_rx_wbfm + k_rx_finish", {"steps": K, "warmup": W, "settle_steps": settle})
    out["fallback_wbfm_64x80_host_replay"] = host_replay_cost(api, device)
    # the north star's target at its own cadence (>= 1000 channels in real time: one block per channel per 64 ms)
    out["realtime_1024x1"] = realtime_cadence(api, device, 1024, "wbfm")
    out["realtime_mixed_1024x1"] = realtime_cadence(api, device, 1024, "mixed", batches=24)
    return out


def rccl_block(shard, dist, device, rank, world):
    """Proof that `world` ranks ran on `world` DEVICES over RCCL: every rank's device identity (PCI bus id, uuid)
    all-gathered, the backend and its version, and one all-reduce of device tensors through it."""
    ident = [None] * world
    dist.all_gather_object(ident, {"rank": rank, **shard.device_identity(device.index)})
    t = torch.tensor([rank + 1], dtype=torch.int64, device=device)
    dist.all_reduce(t)
    keys = {(d.get("host"), d.get("pci_bus_id") or d.get("uuid") or d.get("local_device")) for d in ident}
    backend = dist.get_backend()
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                               # noqa: BLE001
            ver = repr(e)
    return {"backend": backend + (" (= RCCL on ROCm)" if backend == "nccl" else " (REHEARSAL: host-staged, ranks may share a GPU)"),
            "rccl_version": ver, "world_size": world, "devices": ident, "distinct_devices": len(keys),
            "one_device_per_rank": len(keys) == world,
            "all_reduce_of_device_tensors_ok": int(t.item()) == world * (world + 1) // 2}


def measure_scatter(shard, device, dist, C, B, steps, rank, world):
    """The scatter alone: all IQ of the job on rank 0, ONE group of point-to-point sends per step (shard.scatter_iq =
    ncclGroupStart .. ncclGroupEnd under RCCL) into every rank's resident input tensor; K steps between barriers,
    MAX over ranks.  Returns ms per step."""
    mine = torch.zeros((C, B, BLOCK), dtype=torch.int8, device=device)
    root = torch.zeros((world * C, B, BLOCK), dtype=torch.int8, device=device) if rank == 0 else None
    for _ in range(2):
        shard.scatter_iq(root, mine, world * C)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        shard.scatter_iq(root, mine, world * C)
    torch.cuda.synchronize()
    dist.barrier()
    ms = 1e3 * shard.max_over_ranks(time.perf_counter() - t0, device) / steps
    del mine, root
    torch.cuda.empty_cache()
    return ms


def multi_gpu_legs(api, shard, device, dist, args, r_excl, C, B, settle, rank, world):
    """N > 1, in the SAME invocation as the headline ("MS/s and roofline % at G = 1, 2, 4, 8 with scatter
    included and excluded"): the headline's shape again with the per-channel scatter from rank 0 inside the timed
    region, the scatter alone, config 4 (4096 channels over the N GPUs, strong scaling) both ways, and the `rccl` block."""
    common = dict(workload=args.workload, B=B, signal=args.signal, steps=args.steps, warmup=args.warmup, settle=settle,
                  rank=rank, world=world)
    out = {"rccl": rccl_block(shard, dist, device, rank, world)}

    def legs(Cg, r_ex):
        r_in = measure_rx(api, shard, device, dist, C=Cg, scatter=True, **common)
        sc_ms = measure_scatter(shard, device, dist, Cg, B, max(4, min(args.steps, 20)), rank, world)
        per_link = Cg * B * BLOCK                            # bytes that leave rank 0 on EACH of its world - 1 links per step
        return {"channels_per_gpu": Cg, "channels_total": world * Cg,
                "excluded_ms": round(r_ex["ms_per_step"], 4), "included_ms": round(r_in["ms_per_step"], 4),
                "value_excluded_MSamples_per_s": round(r_ex["value"], 1), "value_included_MSamples_per_s": round(r_in["value"], 1),
                "roofline_frac_excluded": round(r_ex["achieved"] / HBM_PEAK_GBS, 4),
                "scatter_alone_ms": round(sc_ms, 4), "bytes_per_link_per_step": per_link, "links_out_of_rank0": world - 1,
                "GBps_per_link": round(per_link / (sc_ms * 1e-3) / 1e9, 1),
                "uncommitted_launches": r_ex["counters"][5] + r_in["counters"][5]}

    if args.scatter:
        # the headline itself included the scatter: measure the leg without it here
        r_plain = measure_rx(api, shard, device, dist, C=C, scatter=False, **common)
        out["scatter"] = legs(C, r_plain)
    else:
        out["scatter"] = legs(C, r_excl)
    if not args.no_config4 and args.workload == "wbfm" and 4096 % world == 0:
        C4 = 4096 // world
        r4 = measure_rx(api, shard, device, dist, C=C4, scatter=False, **common)
        out["config4"] = {"workload": f"BASELINE config 4: 4096 WBFM channels over {world} GPU(s), {C4} per GPU x {B} blocks "
                                      "(STRONG scaling: the N = 1 point is `configs.wbfm_4096x16` of the one-GPU line)",
                          "scaling": "strong", "kernel_ms_mean": round(r4["mean_ms"], 4), **legs(C4, r4)}
    return out


def headline_kernel_name(args):
    """the dominant kernel of the headline workload, with the template arguments it really runs with"""
    if args.workload == "wbfm":
        return ("hrfd::k_rx_wbfm_flow<SVC=4 (flow_body runs WBFM with 6 service + 10 stream waves), GATED=false, DUMP=%s, MODE=3 WBFM>%s"
                % ("true" if args.iqdump else "false", " + the gated pass behind it" if args.quiet_fraction > 0 else ""))
    if args.workload == "mixed" and not args.serial_modes:
        return "hrfd::k_rx_flow_bank<4> (several modes in one launch, the mode read per workgroup)"
    return "hrfd::k_rx_wbfm_flow<4, .., MODE> per mode (--serial-modes: k_rx_fir / k_rx_post / k_rx_finish)"


def libm_variant_name():
    """which build of glibc's sinf / cosf the host's libm is, as libhrfd probed it (visible in the line): the FM
    modulator, Nco::run and the pm / fm generators follow it bit for bit; "unknown" = neither probed build, +-1 LSB there"""
    from hackrfdiags_amd import _lib
    L = _lib.load()
    if not hasattr(L, "hrfd_libm_variant"):
        return None
    return {0: "glibc sincosf without FMA", 1: "glibc sincosf, FMA build", -1: "unknown libm (device follows the FMA build; FM / pm / Nco::run expect +-1 LSB)"}.get(int(L.hrfd_libm_variant()), "?")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps; the clock governor needs ~25 ms of this load to settle (DESIGN.md 5)")
    ap.add_argument("--channels", type=int, default=0,
                    help="channels per GPU; default: 256 for EVERY N (BASELINE config 2's load per GPU: `value` at "
                         "N = 1, 2, 4, 8 is one weak-scaling curve; config 4 -- 4096 channels over the N GPUs -- is measured "
                         "in the same run and reported as `config4`), 1024 for the modulator workloads (config 5)")
    ap.add_argument("--blocks", type=int, default=16, help="262144-byte blocks per channel per step")
    ap.add_argument("--signal", choices=["fmtone", "random"], default="fmtone")
    ap.add_argument("--workload", choices=["wbfm", "mixed", "am", "fm", "ssb", "ssbmod", "ammod", "fmmod", "wbfmmod", "ingest", "fanout"], default="wbfm",
                    help="wbfm = BASELINE config 2 (the headline); mixed = config 3 (AM+FM+WBFM+SSB bank, "
                         "per-mode dispatch); ssbmod = config 5 (SSB modulator, 8-stage x256 interpolator)")
    ap.add_argument("--scatter", action="store_true",
                    help="N > 1 only: the HEADLINE's timed region includes the scatter of all IQ from rank 0 over RCCL "
                         "(without this flag both legs are still measured and reported as `scatter`: `value` excludes it)")
    ap.add_argument("--verify", type=int, default=16,
                    help="receive workloads: after the timed region, this many channels of the bench's own batch go through the "
                         "sequential CPU oracle and must match bit for bit (0: off)")
    ap.add_argument("--no-config4", action="store_true", help="N > 1: leave out the config 4 legs (4096 channels over the N GPUs)")
    ap.add_argument("--quiet-fraction", type=float, default=0.0,
                    help="this fraction of the channels carries no signal (with --threshold: their squelch gates close)")
    ap.add_argument("--threshold", type=int, default=None, help="squelch threshold in dBFS (setSignalDetectThreshold)")
    ap.add_argument("--iqdump", action="store_true", help="also write the 256 kS/s stream (`enable iqdump`)")
    ap.add_argument("--serial-modes", action="store_true",
                    help="mixed bank: one kernel per mode, one after the other (test hook; default: ONE launch, the mode read per workgroup)")
    ap.add_argument("--stride-pad", type=int, default=0,
                    help="experiment: extra bytes between the channels' input buffers (channel_stride = blocks * 262144 + pad)")
    ap.add_argument("--shards", type=int, default=8, help="--workload fanout: shards of the bank (config 4: 8)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extras-verbose", action="store_true",
                    help="print the full `also` object (every entry with its workload text, byte counts, verification record) beside "
                         "the compact `configs`; the default line stays under 8 KB (the driver keeps a line's tail)")
    ap.add_argument("--no-extras", action="store_true",
                    help="leave out the end-to-end and single-block latency figures and the `also` measurements "
                         "(profiling runs: only the headline launches in the kernel statistics)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.channels <= 0:
        args.channels = 1024 if args.workload in MOD_KINDS else 4096 if args.workload == "fanout" else 256
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
            dist.init_process_group(backend="gloo")         # (device shards are staged through the host: shard.scatter_iq)
        else:
            dist.init_process_group(backend="nccl", device_id=device)    # nccl == RCCL on ROCm

    from hackrfdiags_amd import api, shard

    if rank == 0 and args.verify and args.workload not in MOD_KINDS and args.workload not in ("ingest", "fanout"):
        # the checker is built and loaded BEFORE anything is measured (tests/reflib runs `make -C oracle` when the library is
        # stale): a broken checker must not cost the line its measurements
        try:
            from tests import reflib
            reflib.Oracle()
        except Exception as e:                               # noqa: BLE001
            print(f"bench.py: --verify switched off, the oracle does not load: {e!r}", file=sys.stderr)
            args.verify = 0

    if args.workload in MOD_KINDS:
        return bench_mod(args, api, device, rank, world, dist)
    if args.workload == "ingest":
        return bench_ingest(args, api, device, rank, world, dist)
    if args.workload == "fanout":
        if world != 1:
            raise SystemExit("--workload fanout is the one-process path: run it without torchrun")
        return bench_fanout(args, api, device, rank, world)

    C, B = args.channels, args.blocks
    # The clock governor needs ~25 ms of this load before it holds its clock (profiles/README.md): when
    # the caller asks for a short warm-up, untimed "settle" launches come first so that the K timed
    # steps are measured at the clock a long-running stream sees, whatever W is.  (`configs.wbfm_256x16_unsettled`
    # is the figure without them.)
    settle = max(0, SETTLE_STEPS - args.warmup)
    r = measure_rx(api, shard, device, dist, workload=args.workload, C=C, B=B, signal=args.signal, steps=args.steps,
                   warmup=args.warmup, settle=settle, rank=rank, world=world, scatter=args.scatter,
                   quiet_fraction=args.quiet_fraction, threshold=args.threshold, iqdump=args.iqdump,
                   serial_modes=args.serial_modes, stride_pad=args.stride_pad, verify=args.verify if rank == 0 else 0)
    multi = multi_gpu_legs(api, shard, device, dist, args, r, C, B, settle, rank, world) if world > 1 else None
    counters = r["counters"]
    assert r["pcm_produced"] == r["pcm_expected"], f"PCM samples produced {r['pcm_produced']} != {r['pcm_expected']}"
    wname = rx_workload_name(args.workload, C, B, args.signal, args.quiet_fraction, args.iqdump)
    traffic, traffic_src = pmc_traffic(wname)
    read_gbs = stream_gbs(device, 0)
    write_gbs = stream_gbs(device, 1)
    # a read-only kernel cannot read faster than the kernel that does nothing else: if it does, a denominator is wrong
    assert under_profiler() or rehearse or r["achieved"] <= read_gbs, \
        f"{r['achieved']:.0f} GB/s algorithmic > {read_gbs:.0f} GB/s of the plain read kernel"
    value, achieved = r["value"], r["achieved"]

    if rank == 0:
        line = {
            "metric": "IQ MSamples/s demodulated (2.048 MS/s->8 kS/s WBFM) per GPU; % HBM roofline",
            "value": round(value, 1),
            "unit": "MSamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle,
            "ms_per_step": round(r["ms_per_step"], 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8 IQ -> Q15 int16/int32 + f32 recurrence -> int16 PCM",
            "data": "synthetic",
            "config": {
                "workload": rx_workload_text(args.workload, C, B, world, args.signal, args.scatter,
                                             args.quiet_fraction, args.threshold, args.iqdump),
                "channels_per_gpu": C, "blocks_per_step": B, "signal": args.signal,
                "parallelism": f"channels sharded, {world} rank(s), no data-path collective"
                               + (" (REHEARSAL over gloo on shared GPUs: not a measurement)" if rehearse and world > 1 else ""),
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
                "traffic_over_algorithmic": None if traffic is None else round(traffic / r["algo_bytes"], 4),
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": r["algo_bytes"],
                "kernel": headline_kernel_name(args),
                "region_ms_per_launch": round(r["mean_ms"], 4),
                "priced_with": "region_ms_per_launch: one HIP event pair on the launch stream around the %d launches of the timed region / steps "
                               "(a throughput reciprocal: back-to-back launches); kernel_ms_*: single launches bracketed BEHIND the region" % r["launches_timed"],
                "kernel_ms_min": round(r["kernel_ms_min"], 4),
                "kernel_ms_median": round(r["kernel_ms_median"], 4),
                "kernel_ms_sampled_mean": round(r["kernel_ms_sampled_mean"], 4),
                "kernel_launches_sampled": r["launches_sampled"],
                "measured_stream_read_GBps": round(read_gbs, 1),
                "frac_of_measured_read": round(achieved / read_gbs, 4),
                "measured_stream_write_GBps": round(write_gbs, 1),
                "issue": issue_summary() if (args.workload == "wbfm" and C == 256 and B == 16) else None,
            },
            "verification": {"uncommitted_launches": counters[5], "tiles_repaired_in_place": counters[4],
                             "launches": counters[6], **r.get("verification", {"oracle_channels_checked": 0})},
            "kernel_code_tag": kernel_code_tag(),
            "kernel_source_tag": kernel_source_tag(),
            "libm_variant": libm_variant_name(),
            "runtime": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "under_profiler": under_profiler()},
        }
        if multi is not None:
            line.update(multi)
        if counters[5] != 0:
            # a launch that did not commit means later launches started from a stale state and the batch path was
            # not what ran: the number is not a measurement of it
            line["invalid"] = f"{counters[5]} launch(es) were not committed"
        default_headline = (world == 1 and args.workload == "wbfm" and (C, B) == (256, 16) and args.signal == "fmtone"
                            and args.quiet_fraction == 0 and not args.iqdump)
        if world == 1 and args.workload == "wbfm" and not args.no_extras:
            line["end_to_end"] = end_to_end(api, device, C)
            line["single_block_latency_ms"] = single_block_latency_ms(api, device)
        if default_headline and not args.no_extras:
            also = also_lines(api, shard, device, args)
            if args.extras_verbose:
                line["also"] = also
            # The driver keeps a line's TAIL: every other BASELINE configuration, the north star's shape and the robustness
            # cases as one compact object right in front of cpu_baseline (ms per step, fraction of 8 TB/s priced with the
            # region event pair, PMC traffic over algorithmic bytes); the headline repeated as its first entry.
            head = {"ms": line["ms_per_step"], "frac": line["roofline"]["frac"], "traffic_x": line["roofline"]["traffic_over_algorithmic"],
                    "kernel_ms": line["roofline"]["region_ms_per_launch"]}
            cfg = {"wbfm_256x16": head}
            for k, d in also.items():
                if k.startswith("realtime"):
                    cfg[k] = {"p50_ms": d["paced_64ms"]["p50_ms"], "p99_ms": d["paced_64ms"]["p99_ms"], "budget_used_p99": d["paced_64ms"]["budget_used_p99"],
                              **({"invalid": d["invalid"]} if "invalid" in d else {})}
                elif k == "fallback_wbfm_64x80_host_replay":
                    cfg[k] = {"gates_open_ms": d["gates_open_ms"], "gates_closing_ms": d["gates_closing_ms"], "replay_cost_ms": d["replay_cost_ms"]}
                else:
                    cfg[k] = compact(d)
            line["configs"] = cfg
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        text = json.dumps(line)
        if len(text) > 8000 and not args.extras_verbose:
            print(f"bench.py: the line is {len(text)} bytes (> 8000: the driver keeps a tail)", file=sys.stderr)
        print(text, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
