#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[2] -- TestSource-shaped 16-bit IQ streams,
decimate by 16 centred (Decimators::decimate16_cen over IntHalfbandFilterEO1) + UDPSinkFEC
framing + CM256 128+32 encode -- as 8 independent streams per GPU (configs[4] is exactly this
at 8 GPUs: 64 streams, 8 per GPU), 2^25 device-rate samples per stream and step, i.e. 2^28
samples = 1 GiB of int16 IQ per GPU and step, resident in HBM before the timed region.
A step = one sdrhip_rx_process() call over that batch (decimate -> frame -> encode); the
streams are continuous across steps (filter state and partial frames carry over).

Before the W warm-up steps the same step is run untimed for --preroll-seconds (default 0.25 s):
the GPU's clocks ramp over the first ~60 ms of load (measured: 0.69 ms/step right after start,
0.59 ms/step from ~60 ms on, flat over 2000 steps), which a small W would otherwise put into
the timed region.  The timed region is exactly K steps between barrier + synchronize pairs.

One JSON line on rank 0; `value` = whole-job M input samples / s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

STREAMS_PER_GPU = 8
LOG2DECIM, NB_FEC = 4, 32
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per input sample
BYTES_DECIM = 4.0 + 4.0 / 16.0                      # kernel K1: read int16 IQ, write 1/16 of it
BYTES_CONFIG3 = 4.0 + 160.0 * 512.0 / 258064.0      # whole pipe incl. the 160 x 512 B frames


def make_input(device, n, seed, kind):
    g = torch.Generator(device=device).manual_seed(seed)
    if kind == "noise":  # uniform full-scale int16 (the stress input of BASELINE.md 3.4, worst case for toggling)
        return torch.randint(-32768, 32768, (n, 2), generator=g, device=device, dtype=torch.int16)
    # TestSource-like CW (TestSource.cpp:395-422 shape, double precision) + 6 LSB of dither
    k = torch.arange(n, device=device, dtype=torch.float64)
    ph = 2.0 * np.pi * k * ((100e3 + 1e3 * seed) / 10e6)
    a = 3276.8
    x = torch.stack([torch.round(a * torch.cos(ph)), torch.round(a * torch.sin(ph))], dim=1)
    x += torch.randint(-3, 4, (n, 2), generator=g, device=device).to(torch.float64)
    return x.to(torch.int16)


def pmc_traffic(samples_per_launch):
    """HBM bytes per launch of the decimator kernel from the committed PMC passes (profiles/traffic.json,
    collected with tools/prof.sh on this very command); None when the launch geometry differs."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)["decim_kernel<4,2,true>"]
        if int(t["samples_per_launch"]) != int(samples_per_launch):
            return None
        return float(t["fetch_size_kb"]) * 1024.0 * float(t["fetch_correction"]) + float(t["write_size_kb"]) * 1024.0
    except Exception:
        return None


VALU_PEAK_TLANEOPS = 1024 * 16 * 2.4e9 / 1e12  # 1024 SIMDs x 16 lanes / clk x 2.4 GHz: the issue rate of the multiplier-class ops
# (mad / dot2 / perm, 62 % of K1's mix; plain add / xor / shift issue faster: tools/valu_peak.hip)


def pmc_valu_lane_ops(samples_per_launch):
    """integer VALU lane-ops per launch of the decimator kernel (SQ_INSTS_VALU x 64, committed PMC pass)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)["decim_kernel<4,2,true>"]
        if int(t["samples_per_launch"]) != int(samples_per_launch):
            return None
        return float(t["valu_wave_insts"]) * 64.0
    except Exception:
        return None


def cpu_baseline(budget_s):
    """Same pipe on ONE host core: the real reference decimator (oracle/_ref, EO1 build of
    Decimators::decimate16_cen) + the oracle's pshufb CM256 encoder (cm256cc itself is absent:
    that leg is a port).  Bounded sample, scaled to M samples / s."""
    import signals
    from oracle_lib import Oracle, Reference

    orc = Oracle()
    n = 1 << 22
    x = signals.noise(n, 4242)
    use_ref = Reference.available("eo1")
    if use_ref:
        dec = Reference("eo1").decimators()
        kind = "reference"
        run = lambda reps: dec.decimate_repeat(4, 2, 16, x, reps)  # noqa: E731
    else:  # pragma: no cover - oracle/_ref travels with the repo
        dec = orc.decimators(0)
        kind = "port"
        run = lambda reps: [dec.decimate(4, 2, 16, x) for _ in range(reps)][-1][0]  # noqa: E731
    y = run(1)  # warm-up pass
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s * 0.8:
        y = run(4)
        reps += 4
    t_dec = (time.perf_counter() - t0) / reps  # seconds per 2^22 samples
    # FEC leg on the frames that 2^22 input samples produce (16 frames + change): time 16 frames
    y = np.ascontiguousarray(y[:16 * 16129])
    frames = orc.framer(nb_fec_blocks=NB_FEC).write(y)
    t1 = time.perf_counter()
    nfr = 0
    while time.perf_counter() - t1 < budget_s * 0.2:
        for f in range(frames.shape[0]):
            orc.frame_encode(frames[f], NB_FEC)
        nfr += frames.shape[0]
    t_fec = (time.perf_counter() - t1) / nfr  # seconds per frame = per 258064 input samples
    sec_per_sample = t_dec / n + t_fec / 258064.0
    return {
        "value": round(1e-6 / sec_per_sample, 3), "unit": "Msamples/s", "cores": 1, "kind": kind,
        "sample": "%d x 2^22 random full-scale samples through %s decimate16_cen (%.1f Msamples/s alone) + %d frames "
                  "through the oracle's SSSE3 cm256 128+32 encoder (port; %.2f ms/frame), one thread, serialised" %
                  (reps, "the reference's compiled" if use_ref else "the oracle's", 1e-6 * n / t_dec, nfr, 1e3 * t_fec),
        "decimate_only_msps": round(1e-6 * n / t_dec, 3),
    }


def cpu_all_cores(budget_s):
    """the reference decimator with one independent stream per hardware thread
    (the reference itself runs one decimator thread per process).  Decimation leg only -- it is 95 % of the
    CPU pipe's time; reported next to cpu_baseline, never used as the denominator of anything."""
    import concurrent.futures as cf

    import signals
    from oracle_lib import Reference

    if not Reference.available("eo1"):
        return None
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ref = Reference("eo1")
    n = 1 << 20
    xs = [signals.noise(n, 5000 + t) for t in range(ncpu)]
    decs = [ref.decimators() for _ in range(ncpu)]
    reps = 8

    def work(t):
        k = 0
        t_end = time.perf_counter() + budget_s
        while time.perf_counter() < t_end:
            decs[t].decimate_repeat(4, 2, 16, xs[t], reps)  # ctypes releases the GIL for the call
            k += reps
        return k

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(ncpu) as ex:
        total = sum(ex.map(work, range(ncpu)))
    dt = time.perf_counter() - t0
    return {"value": round(total * n / dt / 1e6, 1), "unit": "Msamples/s", "cores": ncpu, "kind": "reference",
            "sample": "%d threads x reference decimate16_cen on 2^20-sample blocks for %.1f s (decimation leg only)" % (ncpu, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2-samples", type=int, default=25, help="samples per stream per step (default 2^25)")
    ap.add_argument("--input", choices=["noise", "testsource"], default="noise")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--preroll-seconds", type=float, default=0.25,
                    help="untimed run-in of the same step before the W warm-up steps (the GPU's clocks ramp over the first "
                         "~60 ms of load: with a small W the timed steps would measure that ramp)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="diagnostic: leave the per-kernel HIP events out of the timed region (roofline fields become null)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to dry-run the N > 1 path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            local %= torch.cuda.device_count()  # dry run: ranks may share a GPU
            torch.cuda.set_device(local)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # nccl == RCCL on ROCm
    else:
        dist = None
        torch.cuda.set_device(local)
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N > 1)"

    import sdrdaemon_amd as sd
    from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_ENCODE

    dev = torch.device("cuda", local)
    ctx = sd.Context(local)
    S, n = STREAMS_PER_GPU, 1 << args.log2_samples
    # streams are sharded one-per-stream across ranks: global stream id = rank * S + s
    from sdrdaemon_amd import sharding

    x = torch.stack([make_input(dev, n, 1000 + sid, args.input) for sid in sharding.stream_ids(rank, world, S)])
    rx = sd.RxPipe(ctx, S, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                   center_frequency_khz=435000, sample_rate=625000)

    def step(i):
        # finished frames stay in the library's frame area (zero-copy view, valid until the next call),
        # the way transmitUDP sends straight out of m_txBlocks (UDPSinkFEC.cpp:259-282)
        return rx.process_view(x, tv_sec=i, tv_usec=0)

    preroll = 0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preroll_seconds:
        for _ in range(10):
            step(0)
        torch.cuda.synchronize()
        preroll += 10
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    ctx.kernel_timing(not args.no_kernel_timing)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = 0
    for i in range(args.steps):
        frames += step(args.warmup + i).shape[1]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dec_ms, dec_n = ctx.kernel_timing_read(K_DECIMATE)
    fec_ms, fec_n = ctx.kernel_timing_read(K_FEC_ENCODE)
    ctx.kernel_timing(False)
    # the only collectives of the job: MAX of the elapsed time, SUM of the samples (8 bytes each, reporting only)
    elapsed, total_samples = sharding.aggregate(elapsed, float(S) * n * args.steps, dist,
                                                dev if args.backend == "nccl" else torch.device("cpu"))

    if rank == 0:
        value = total_samples / elapsed / 1e6
        per_launch_samples = float(S) * n
        avg_ms = dec_ms / max(dec_n, 1) if dec_n else float("nan")
        achieved = BYTES_DECIM * per_launch_samples / (avg_ms * 1e-3) / 1e9
        res = {
            "metric": "IQ Msamples/s through decim+FEC-encode pipe; bit-exact vs CPU ref",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "preroll_steps": preroll,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32",
            "data": "synthetic: %s, %d streams/GPU (stream id = rank*%d + s), HBM-resident before the timed region" %
                    ("uniform random full-scale int16 IQ" if args.input == "noise" else "TestSource-like CW A=0.1 + dither", S, S),
            "config": {"workload": "configs[2] x %d streams/GPU: 10 Msps-shaped int16 IQ, decimate16_cen (EO1) + UDPSinkFEC framing + "
                                   "CM256 128+32 encode" % S,
                       "streams_per_gpu": S, "samples_per_stream_per_step": n, "log2decim": LOG2DECIM, "fcpos": "cen",
                       "nb_fec": NB_FEC, "hb_variant": "EO1", "frames_per_stream_per_step": frames // max(args.steps, 1), "output": "zero-copy view of the frame area",
                       "parallelism": "stream-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(per_launch_samples),
                         "kernel": "decim_kernel<L=4,FC=cen,PACK16>", "launches": dec_n, "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_bytes_per_launch": BYTES_DECIM * per_launch_samples,
                         "pipe_gbps_config3": round(BYTES_CONFIG3 * value * 1e6 / 1e9 / world, 1),
                         "fec_encode_avg_launch_ms": round(fec_ms / max(fec_n, 1), 4), "fec_encode_launches": fec_n},
        }
        lane_ops = pmc_valu_lane_ops(per_launch_samples)
        if lane_ops and dec_n:
            # secondary figure of SURVEY.md 8(d): integer VALU issue, every op counted at the 16-lane / clk rate
            tl = lane_ops / (avg_ms * 1e-3) / 1e12
            res["roofline"]["valu"] = {"achieved": round(tl, 2), "peak": round(VALU_PEAK_TLANEOPS, 2), "unit": "T lane-ops/s",
                                       "frac": round(tl / VALU_PEAK_TLANEOPS, 4), "lane_ops_per_sample": round(lane_ops / per_launch_samples, 2)}
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            res["gpu_over_cpu_1core"] = round(value / res["cpu_baseline"]["value"], 1)
            allc = cpu_all_cores(min(4.0, args.cpu_seconds / 3.0))
            if allc:
                res["cpu_baseline_all_cores"] = allc
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
