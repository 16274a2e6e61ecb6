#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or as the
     plain command above: without WORLD_SIZE in the environment it launches the N ranks itself)

Workload (config.workload): BASELINE.json configs[2] -- TestSource-shaped 16-bit IQ streams,
decimate by 16 centred (Decimators::decimate16_cen over IntHalfbandFilterEO1) + UDPSinkFEC
framing + CM256 128+32 encode -- as 8 independent streams per GPU (configs[4] is exactly this
at 8 GPUs: 64 streams, 8 per GPU), 2^25 device-rate samples per stream and step, i.e. 2^28
samples = 1 GiB of int16 IQ per GPU and step, resident in HBM before the timed region.
A step = one sdrhip_rx_process() call over that batch (decimate -> frame -> encode); the
streams are continuous across steps (filter state and partial frames carry over).

--streams N fixes the TOTAL number of streams of the job (the same 64 streams at
2 / 4 / 8 GPUs, stream s on rank s mod G): "scaling" is then "strong"; that is the default for
--gpus > 1 (64 streams = configs[4]); --streams 0 gives every rank 8 streams ("weak"), which is
also what the one-GPU headline run does.  At N = 1 the line also carries `configs`: the other single-GPU
configurations of BASELINE.json measured in the same run (configs[1] decimate16_cen alone,
configs[2] as ONE stream of 2^27 samples, configs[3] the Tx pipe with a different random
24-erasure pattern in every frame), each with its own roofline object.

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
PCIE_PEAK_GBS = 63.0   # host link, PCIe Gen5 x16 per direction (same guide): the bound of the host-fed (drop-in) mode
# algorithmic bytes per input sample
BYTES_DECIM = 4.0 + 4.0 / 16.0                      # kernel K1: read int16 IQ, write 1/16 of it
BYTES_CONFIG3 = 4.0 + 160.0 * 512.0 / 258064.0      # whole pipe incl. the 160 x 512 B frames


class BoxState:
    """Socket power and shader clock of the GPU while the benchmark runs (both dominant kernels run at the board's
    power cap and boxes differ by ~10 % in what they make of it -- a line without these cannot tell a slow kernel from a hot box).
    A thread polls amdsmi (in-process, ~1 ms per sample; fallback: the amdgpu hwmon files) between start() and stop();
    summary() -> {"power_w": mean, "power_w_max", "sclk_mhz": mean, "sclk_mhz_min", "samples", "power_cap_w", "source"} or None."""

    def __init__(self, index=0, period_s=0.002):
        import threading

        self.index, self.period = index, period_s
        self.rows, self._stop, self._thr = [], threading.Event(), None
        self.cap_w, self.source, self._read = None, None, None
        try:
            import amdsmi

            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[index]

            def read():
                p = amdsmi.amdsmi_get_power_info(h)
                w = p.get("current_socket_power") or p.get("socket_power") or p.get("average_socket_power")
                c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                return float(w), float(c.get("clk") or c.get("cur_clk"))

            read()
            self._read, self.source = read, "amdsmi (current_socket_power, GFX clk)"
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(h).get("power_cap")
                self.cap_w = float(cap) / (1e6 if cap and cap > 1e5 else 1.0)
            except Exception:
                pass
        except Exception:
            self._read = self._hwmon_reader()

    def _hwmon_reader(self):
        import glob

        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        if self.index >= len(cards):
            return None
        d = cards[self.index]
        pw = [f for f in (d + "/power1_input", d + "/power1_average") if os.path.exists(f)]
        fq = d + "/freq1_input"
        if not pw or not os.path.exists(fq):
            return None
        self.source = "amdgpu hwmon (%s, freq1_input)" % os.path.basename(pw[0])

        def read():
            with open(pw[0]) as f:
                w = float(f.read()) / 1e6
            with open(fq) as f:
                c = float(f.read()) / 1e6
            return w, c
        try:
            read()
        except Exception:
            return None
        return read

    def start(self):
        import threading

        if self._read is None or self._thr is not None:
            return self
        self.rows = []
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                try:
                    self.rows.append(self._read())
                except Exception:
                    pass
                self._stop.wait(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join()
            self._thr = None
        return self.summary()

    def summary(self):
        rows = [r for r in self.rows if r[0] > 0]
        if not rows:
            return None
        pw, ck = [r[0] for r in rows], [r[1] for r in rows]
        return {"power_w": round(sum(pw) / len(pw), 1), "power_w_max": round(max(pw), 1), "sclk_mhz": round(sum(ck) / len(ck), 1),
                "sclk_mhz_min": round(min(ck), 1), "samples": len(rows), "power_cap_w": self.cap_w, "source": self.source}


def make_input(ctx, device, n, seeds, kind):
    """-> (len(seeds), n, 2) int16 on the device.  noise: uniform full-scale int16 (the stress input of BASELINE.md 3.4,
    worst case for toggling); testsource: the library's TestSource bank (10 Msps, -20 dB CW at +100 kHz + 1 kHz per
    stream id: TestSource.cpp:59-215 semantics, README.md:362 signal), generated on the device."""
    if kind == "hash":
        # counter-based full-scale uniform noise (tests/signals.py): the stream the committed whole-output digests of the
        # compiled reference were made from (tests/golden/headline_golden.json) -- see verify_step()
        import signals

        return torch.stack([signals.hash_noise_torch(n, seed, device) for seed in seeds])
    if kind == "noise":
        out = []
        for seed in seeds:
            g = torch.Generator(device=device).manual_seed(seed)
            out.append(torch.randint(-32768, 32768, (n, 2), generator=g, device=device, dtype=torch.int16))
        return torch.stack(out)
    import sdrdaemon_amd as sd

    import headline_inputs as hi

    ts = sd.TestSource(ctx, len(seeds))
    for s, seed in enumerate(seeds):
        assert ts.configure(hi.ts_config_string(seed), s), ts.error()
    x = ts.read(n)
    ctx.synchronize()
    return x.reshape(len(seeds), n, 2).contiguous()


def decim_kernel_name(plan):
    """the kernel the library actually launched for decimate16_cen (sdrhip_decimators_last_plan / sdrhip_rx_last_plan)"""
    return {"valu": "decim_kernel<4,2,true>", "mfma": "decim_mfma_kernel<4,true>"}.get(plan["path"], "none")


def pmc_traffic(samples_per_launch, kernel):
    """HBM bytes per launch of the decimator kernel from the committed PMC passes (profiles/traffic.json,
    collected with tools/prof.sh on this very command); None when the launch geometry differs."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)[kernel]
        if int(t["samples_per_launch"]) != int(samples_per_launch):
            return None
        return float(t["fetch_size_kb"]) * 1024.0 * float(t["fetch_correction"]) + float(t["write_size_kb"]) * 1024.0
    except Exception:
        return None


VALU_PEAK_TLANEOPS = 1024 * 16 * 2.4e9 / 1e12  # 1024 SIMDs x 16 lanes / clk x 2.4 GHz: the issue rate of the multiplier-class ops
# (mad / dot2 / perm, 62 % of K1's mix; plain add / xor / shift issue faster: tools/valu_peak.hip)


def pmc_valu_lane_ops(samples_per_launch, kernel):
    """integer VALU lane-ops per launch of the decimator kernel (SQ_INSTS_VALU x 64, committed PMC pass)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)[kernel]
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


def timed_steps(ctx, fn, classes, steps=40, preroll_s=0.15):
    """-> (wall ms per step, {kernel class: avg launch ms}) of fn() after a clock run-in"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preroll_s:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    ctx.set_option("ktime_stride", 4)  # (every 4th launch of a class carries the event pair, see main())
    ctx.kernel_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    per = {}
    for c in classes:
        ms, cnt = ctx.kernel_timing_read(c)
        per[c] = ms / max(cnt, 1)
    ctx.kernel_timing(False)
    ctx.set_option("ktime_stride", 1)
    return wall, per


def pmc_traffic_units(kernel, units_key, units):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/traffic.json), when the entry was collected on
    the same number of units (outputs / samples) per launch; else None"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)[kernel]
        if int(t[units_key]) != int(units):
            return None
        return float(t["fetch_size_kb"]) * 1024.0 * float(t["fetch_correction"]) + float(t["write_size_kb"]) * 1024.0
    except Exception:
        return None


def roof(bytes_per_launch, launch_ms, kernel, traffic=None):
    ach = bytes_per_launch / (launch_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
         "kernel": kernel, "avg_launch_ms": round(launch_ms, 4), "traffic": traffic}
    if traffic is not None:
        r["traffic_source"] = "profiles/traffic.json (committed PMC passes of the same launch geometry; not measured by this run)"
    return r


def extra_configs(ctx, dev, x, kind, ids):
    """The other single-GPU configurations of BASELINE.json, same process, same input tensors."""
    import sdrdaemon_amd as sd
    from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_DECODE, K_INTERPOLATE

    out = []
    S, n = x.shape[0], x.shape[1]
    # configs[1]: decimate16_cen alone, FEC off
    d = sd.Decimators(ctx, S, sd.HB_EO1)
    y = torch.empty((S, n >> LOG2DECIM, 2), dtype=torch.int16, device=dev)
    wall, per = timed_steps(ctx, lambda: d.decimate(LOG2DECIM, sd.FC_CEN, 16, x, out=y), [K_DECIMATE])
    out.append({"config": "configs[1]: %d streams x 2^%d samples, decimate16_cen (EO1), FEC off" % (S, n.bit_length() - 1),
                "ms_per_step": round(wall, 4), "value": round(S * n / wall / 1e3, 1), "unit": "Msamples/s (input)",
                "roofline": roof(BYTES_DECIM * S * n, per[K_DECIMATE], decim_kernel_name(d.last_plan()),
                                 pmc_traffic(float(S) * n, decim_kernel_name(d.last_plan()))),
                "verified": verify_decim(ctx, x, ids, kind)})
    del y, d
    # the headline step on the input every BASELINE config names: TestSource (10 Msps CW; the bank's integer NCO on the device,
    # tests/headline_inputs.py), verified against digests made in the build container by the compiled reference decimator over the
    # oracle's restatement of that NCO (headline_golden.json ts_bank8).  A CW carrier 20 dB under full scale toggles far fewer bits
    # than full-scale noise: on kernels that sit at the board's power cap the bit pattern is part of the result.
    if kind != "testsource":
        xt = make_input(ctx, dev, n, [1000 + sid for sid in ids], "testsource")
        rxt = sd.RxPipe(ctx, S, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                        center_frequency_khz=435000, sample_rate=625000)
        from sdrdaemon_amd.engine import K_FEC_ENCODE as _KFE
        wall, per = timed_steps(ctx, lambda: rxt.process_view(xt, tv_sec=1, tv_usec=0), [K_DECIMATE, _KFE])
        plan_t = rxt.last_plan()
        del rxt
        import headline_inputs as _hi
        out.append({"config": "configs[2] x %d streams on TestSource input: %s ... (the bank's integer NCO, generated on the device), decimate16_cen + framing + "
                              "CM256 128+32 -- the headline step, other bit pattern" % (S, _hi.ts_config_string(1000 + ids[0])),
                    "ms_per_step": round(wall, 4), "value": round(S * n / wall / 1e3, 1), "unit": "Msamples/s (input)",
                    "fec_encode_avg_launch_ms": round(per[_KFE], 4),
                    "roofline": roof(BYTES_DECIM * S * n, per[K_DECIMATE], decim_kernel_name(plan_t)),
                    "path_frac": round(BYTES_CONFIG3 * S * n / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "verified": verify_step(ctx, xt, ids, "testsource")})
        del xt
    # the headline workload in the round-4 ARRANGEMENT (rx_direct = 0): the matrix-core decimator stores in stream order, K2 and the
    # encoder's fused copy frame it (three roles, two launches).  Same frames; the decimator alone is faster there (its writes are
    # 67 MB that stay in the Infinity Cache), the encoder slower (it carries the copy), the step about the same, the traffic 134 MB more.
    from sdrdaemon_amd.engine import K_FEC_ENCODE
    ctx.set_option("rx_direct", 0)
    try:
        rxs = sd.RxPipe(ctx, S, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                        center_frequency_khz=435000, sample_rate=625000)
        wall, per = timed_steps(ctx, lambda: rxs.process_view(x, tv_sec=1, tv_usec=0), [K_DECIMATE, K_FEC_ENCODE])
        del rxs
        out.append({"config": "configs[2] x %d streams, stream-order arrangement (rx_direct = 0): decimator output in stream order, framing by K2 + the "
                              "encoder's fused copy (gf_encode128_fft_pack_kernel)" % S,
                    "ms_per_step": round(wall, 4), "value": round(S * n / wall / 1e3, 1), "unit": "Msamples/s (input)",
                    "fec_encode_avg_launch_ms": round(per[K_FEC_ENCODE], 4),
                    "roofline": roof(BYTES_DECIM * S * n, per[K_DECIMATE], "decim_mfma_kernel<4,true> (stream-order stores)"),
                    "verified": verify_step(ctx, x, ids, kind)})
    finally:
        ctx.set_option("rx_direct", 1)
    # the headline workload through the pipelined plumbing (sdrhip_rx_set_pipelined: frames delivered one call late).  Two variants of
    # where the waiting encode runs: inside the decimator launch of the next call (rx_fused_kernel), or as its own launch on the
    # context's second stream BESIDE that decimator (LDS-DMA ring of depth 3 so that two encoder workgroups fit on every CU).
    # Steady state: every timed step holds one decimation and one encode.
    for fused, label, kern in ((1, "CM256 encoder workgroups inside the decimator's launch (rx_fused_kernel)", "rx_fused_kernel<4,true>"),
                               (3, "CM256 encoder on the second stream beside the decimator (two streams, ring depth 3)", "decim_mfma_kernel<4,true> (ring 3) || gf_encode128_fft_kernel")):
        ctx.set_option("rx_fused", fused)
        try:
            rxp = sd.RxPipe(ctx, S, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                            center_frequency_khz=435000, sample_rate=625000, pipelined=True)
            wall, per = timed_steps(ctx, lambda: rxp.process_view(x, tv_sec=1, tv_usec=0), [K_DECIMATE])
            rxp.flush_view()
            del rxp
            out.append({"config": "configs[2] x %d streams, pipelined plumbing: frames delivered one call late, %s" % (S, label),
                        "ms_per_step": round(wall, 4), "value": round(S * n / wall / 1e3, 1), "unit": "Msamples/s (input)",
                        "roofline": roof(BYTES_CONFIG3 * S * n, wall if fused == 3 else per[K_DECIMATE],
                                         kern + (" (decimator + encoder of the previous call: config-3 algorithmic bytes, 4.317 B per sample%s)" %
                                                 ("; the two launches overlap, the step's wall time is the launch time" if fused == 3 else "")),
                                         pmc_traffic(float(S) * n, kern) if fused == 1 else None),
                        "verified": verify_step(ctx, x, ids, kind, pipelined=True)})
        finally:
            ctx.set_option("rx_fused", 1)
    # configs[2] literally: ONE stream (2^27 samples per step) through the fused Rx pipe
    n1 = 1 << 27
    x1 = make_input(ctx, dev, n1, [4000], kind)
    rx1 = sd.RxPipe(ctx, 1, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC)
    wall, per = timed_steps(ctx, lambda: rx1.process_view(x1, tv_sec=1, tv_usec=0), [K_DECIMATE])
    out.append({"config": "configs[2] as one stream: 2^27 samples per step, decimate16_cen + framing + CM256 128+32",
                "ms_per_step": round(wall, 4), "value": round(n1 / wall / 1e3, 1), "unit": "Msamples/s (input)",
                "roofline": roof(BYTES_DECIM * n1, per[K_DECIMATE], decim_kernel_name(rx1.last_plan()),
                                 pmc_traffic(float(n1), decim_kernel_name(rx1.last_plan()) + "@%d" % n1)),
                "verified": verify_one_stream(ctx, x1, kind)})
    del x1, rx1
    # configs[3]: Tx pipe.  The received frames are config 3's OUTPUT: the first 128 frames of every stream of this very bank
    # through the Rx pipe, 24 of each frame's 160 blocks lost (a DIFFERENT random set in every frame, tests/headline_inputs.py),
    # frames resident on the device (block indices read from the headers by the planning kernel), decode + interpolate by 16.
    import headline_inputs as hi

    meta = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": NB_FEC}
    rxf, keep = hi.tx_received_frames(ctx, x, meta)
    Stx, F = rxf.shape[0], rxf.shape[1]
    tx = sd.TxPipe(ctx, Stx, hi.TX_LOG2_INTERP)
    ctx.set_option("dec_max_rows", NB_FEC)  # the sender's fecblk (it is in every frame's meta block): no frame carries more recovery blocks
    try:
        wall, per = timed_steps(ctx, lambda: tx.process(rxf), [K_FEC_DECODE, K_INTERPOLATE])
        del tx
        tx_verified = verify_tx_step(ctx, rxf, kind, n)
    finally:
        ctx.set_option("dec_max_rows", 128)
    nout = Stx * F * 16129 << hi.TX_LOG2_INTERP
    out.append({"config": "configs[3]: %d streams x %d frames per step (config 3's frames of this bank), UDPSourceFEC decode 128+32 with 24 erased "
                          "blocks (a distinct random pattern per frame, %d distinct) + interpolate16_cen" % (Stx, F, len({k.tobytes() for k in keep})),
                "ms_per_step": round(wall, 4), "value": round(nout / wall / 1e3, 1), "unit": "Msamples/s (output)",
                "decode_ms_per_step": round(per[K_FEC_DECODE], 4),
                "roofline": roof((4.0 + 4.0 / 16.0) * nout, per[K_INTERPOLATE], interp_kernel_name(ctx),
                                 pmc_traffic_units(interp_kernel_name(ctx), "outputs_per_launch", nout)),
                "pipe_gbps_config4": round((4.0 + 128.0 * 512.0 / 258064.0) * nout / (wall * 1e-3) / 1e9, 1),
                "verified": tx_verified})
    # configs[3] through the pipelined plumbing (sdrhip_tx_set_pipelined): the decode of batch N on the second stream beside the
    # interpolator of batch N - 1, samples delivered one call late
    ctx.set_option("dec_max_rows", NB_FEC)
    try:
        txp = sd.TxPipe(ctx, Stx, hi.TX_LOG2_INTERP, pipelined=True)
        wall, per = timed_steps(ctx, lambda: txp.process(rxf), [K_INTERPOLATE])
        txp.flush(device=rxf.device)
        del txp
        txp_verified = verify_tx_step(ctx, rxf, kind, n, pipelined=True)
    finally:
        ctx.set_option("dec_max_rows", 128)
    out.append({"config": "configs[3], pipelined plumbing: %d streams x %d frames per step, decode of this batch on the second stream beside the "
                          "interpolator of the previous batch, samples delivered one call late" % (Stx, F),
                "ms_per_step": round(wall, 4), "value": round(nout / wall / 1e3, 1), "unit": "Msamples/s (output)",
                "roofline": roof((4.0 + 128.0 * 512.0 / 258064.0) * nout, wall, interp_kernel_name(ctx) + " || gf_decode128_fft_kernel (config-4 algorithmic "
                                 "bytes, 4.254 B per output; the launches overlap, the step's wall time is the launch time)"),
                "verified": txp_verified})
    out.extend(drop_in_lines(ctx))
    return out


def drop_in_lines(ctx):
    """The drop-in (host-fed) mode as sdrdaemonrx drives it (VERDICT r5 #3b): one TestSource block of 65 536 samples per call
    (TestSource.h:33) handed over as a HOST buffer (sdrdaemonrx.cpp:590,640) -- PCIe-inclusive, host to host, through the Python
    mirror of the adapters (ctypes adds ~2 us per call over the C++ headers).  Three shapes: Downsampler::process (decimate16_cen
    alone, synchronous), the Rx pipe synchronous (decimate + framing + CM256 128+32), and the asynchronous entry
    (sdrhip_rx_submit / sdrhip_rx_collect) with 16 blocks per upload + launch + download, three batches in flight.
    Bound: the host link (4 bytes in per input sample + 0.317 out)."""
    import sdrdaemon_amd as sd

    nblk = 65536
    rng = np.random.default_rng(77)
    x = rng.integers(-32768, 32768, (nblk, 2), dtype=np.int16)
    lines = []

    def line(config, us_per_block, nbytes, extra=None, ok=None):
        gbs = nbytes / (us_per_block * 1e-6) / 1e9
        d = {"config": config, "verified": {"ok": bool(ok), "what": "every byte the host-pointer calls returned for a run of blocks",
                                            "against": "the device-pointer entry of this library on the same samples (its parity with the oracle / the "
                                                       "reference digests: the lines above and tests/test_gpu_pipes.py)"}, "us_per_block": round(us_per_block, 2), "value": round(nblk / us_per_block, 1), "unit": "Msamples/s (input, host to host)",
             "roofline": {"bound": "pcie", "achieved": round(gbs, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / PCIE_PEAK_GBS, 4),
                          "peak_source": "PCIe Gen5 x16, 63 GB/s per direction (MI355X_MICROARCH.md, host link)", "traffic": None}}
        if extra:
            d.update(extra)
        lines.append(d)

    # (a) Downsampler::process: decimate16_cen of one block, synchronous
    d = sd.Decimators(ctx, 1, sd.HB_EO1)
    for _ in range(100):
        d.decimate(LOG2DECIM, sd.FC_CEN, 16, x)
    K = 500
    t0 = time.perf_counter()
    for _ in range(K):
        d.decimate(LOG2DECIM, sd.FC_CEN, 16, x)
    us = (time.perf_counter() - t0) / K * 1e6
    # check: 6 more blocks through host pointers on fresh handles against the same 6 blocks as one device-memory bank call
    chk = rng.integers(-32768, 32768, (6 * nblk, 2), dtype=np.int16)
    dh, dd = sd.Decimators(ctx, 1, sd.HB_EO1), sd.Decimators(ctx, 1, sd.HB_EO1)
    yh = np.concatenate([dh.decimate(LOG2DECIM, sd.FC_CEN, 16, chk[b * nblk:(b + 1) * nblk])[0] for b in range(6)])
    yd = dd.decimate(LOG2DECIM, sd.FC_CEN, 16, torch.from_numpy(chk).cuda())[0]
    ctx.synchronize()
    line("drop-in, configs[1] shape: Downsampler::process on one 65 536-sample host block per call (decimate16_cen, synchronous, host pointers)", us,
         nblk * BYTES_DECIM, {"kernel_path": d.last_plan()["path"]}, ok=np.array_equal(yh.reshape(-1, 2), yd.reshape(-1, 2).cpu().numpy()))
    del d, dh, dd
    # (b) the Rx pipe on one block per call, synchronous
    rx = sd.RxPipe(ctx, 1, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                   center_frequency_khz=435000, sample_rate=625000)
    for _ in range(100):
        rx.process(x, 1, 2)
    t0 = time.perf_counter()
    for _ in range(K):
        rx.process(x, 1, 2)
    us = (time.perf_counter() - t0) / K * 1e6

    def pipe():
        return sd.RxPipe(ctx, 1, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                         center_frequency_khz=435000, sample_rate=625000)

    rh, rd = pipe(), pipe()
    fh = np.concatenate([rh.process(chk[b * nblk:(b + 1) * nblk], 1, 2) for b in range(6)])
    fd = np.concatenate([rd.process(torch.from_numpy(chk[b * nblk:(b + 1) * nblk]).cuda(), 1, 2).cpu().numpy() for b in range(6)])
    line("drop-in, configs[2] shape: sdrhip_rx_process on one 65 536-sample host block per call (decimate16_cen + framing + CM256 128+32, synchronous)", us,
         nblk * BYTES_CONFIG3, ok=fh.shape[0] == 1 and np.array_equal(fh, fd))
    del rx, rh, rd
    # (c) the asynchronous entry: 16 blocks per batch from pageable memory, the collector one batch behind
    blocks, nb = 16, 16 * 16
    src = rng.integers(-32768, 32768, (1, nb * nblk, 2), dtype=np.int16)
    rx = sd.RxPipe(ctx, 1, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                   center_frequency_khz=435000, sample_rate=625000)
    rx.set_async(depth=4, blocks=blocks)
    blks = [src[:, b * nblk:(b + 1) * nblk] for b in range(nb)]
    maxf = blocks * nblk // (16129 * 16) + 2

    def run(rounds, keep=None):
        inflight, frames = 0, 0

        def take():
            f = rx.collect(wait=True, max_frames=maxf)
            if keep is not None:
                keep.append(f[0].copy())
            return f.shape[1]

        for _ in range(rounds):
            for b in range(nb):
                rx.submit(blks[b], 1, 2)
                if (b + 1) % blocks == 0:
                    inflight += 1
                    if inflight == 3:
                        frames += take()
                        inflight -= 1
        while inflight:
            frames += take()
            inflight -= 1
        return frames

    first = []
    run(1, first)
    # (the first round's frames against the same samples through the device-pointer pipe, batch by batch: same stamps, same frames)
    rd = pipe()
    fd = np.concatenate([rd.process(torch.from_numpy(src[:, b * blocks * nblk:(b + 1) * blocks * nblk]).cuda(), 1, 2)[0].cpu().numpy() for b in range(nb // blocks)])
    ok_async = np.array_equal(np.concatenate(first), fd)
    del rd
    R = 6
    t0 = time.perf_counter()
    frames = run(R)
    us = (time.perf_counter() - t0) / (R * nb) * 1e6
    ok_async = ok_async and abs(frames - R * nb * nblk // (16 * 16129)) <= 1  # (every submitted sample came back framed)
    line("drop-in, configs[2] shape, asynchronous entry: sdrhip_rx_submit / sdrhip_rx_collect, 65 536-sample host blocks (pageable), %d blocks per "
         "upload + launch + download, 3 batches in flight" % blocks, us, nblk * BYTES_CONFIG3, {"frames_collected": int(frames)}, ok=ok_async)
    del rx
    return lines


def verify_one_stream(ctx, x1, kind):
    """configs[2] as one stream behind its timed region: the frames of one step on fresh handles against the committed digest
    (headline_golden.json one27: compiled reference decimator + framer / encoder restatement), else against the VALU path"""
    import hashlib

    import sdrdaemon_amd as sd

    def digest(path):
        ctx.set_option("decim_path", path)
        try:
            rx = sd.RxPipe(ctx, 1, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                           center_frequency_khz=435000, sample_rate=625000)
            fr = rx.process_view(x1, tv_sec=1, tv_usec=0).torch()
            ctx.synchronize()
            return hashlib.sha256(fr[0].contiguous().cpu().numpy().tobytes()).hexdigest()
        finally:
            ctx.set_option("decim_path", "auto")

    got = digest("auto")
    what = "sha256 of the stream's whole frame stream of one step"
    if kind == "hash":
        try:
            with open(os.path.join(ROOT, "tests", "golden", "headline_golden.json")) as f:
                b = json.load(f)["one27"]
            if (1 << b["log2n"]) == x1.shape[1] and b["seeds"] == [4000]:
                return {"ok": got == b["frames_sha256"][0], "what": what,
                        "against": "tests/golden/headline_golden.json one27 (compiled reference decimate16_cen + framer / CM256 restatement)"}
        except Exception:
            pass
    return {"ok": got == digest("valu"), "what": what, "against": "the VALU kernel path of this library on the same input"}


def strong_layout_at_one_gpu(ctx, dev, n, kind, total=64):
    """The job `bench.py --gpus N` runs for every N > 1 -- SURVEY 8e's fixed bank of 64 streams, stream s on rank s mod N -- at
    N = 1 (the N = 1 point of a scaling curve must be the same job as N = 2 .. 8; the headline step above is its
    per-GPU share at N = 8).  Same call, 64 streams on the one GPU; verified against the committed reference digests of the
    64 x 2^25 bank (headline_golden.json bank64_25) when the geometry matches."""
    import sdrdaemon_amd as sd
    from sdrdaemon_amd import sharding
    from sdrdaemon_amd.engine import K_DECIMATE

    ids = sharding.stream_ids_strong(0, 1, total)
    x = make_input(ctx, dev, n, [1000 + sid for sid in ids], kind)
    rx = sd.RxPipe(ctx, total, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                   center_frequency_khz=435000, sample_rate=625000)
    wall, per = timed_steps(ctx, lambda: rx.process_view(x, tv_sec=1, tv_usec=0), [K_DECIMATE], steps=12)
    plan = rx.last_plan()
    del rx
    line = {"config": "configs[4]'s job at N = 1: the strong layout's %d streams x 2^%d samples on one GPU (what --gpus 2 / 4 / 8 shard s mod N)" % (total, n.bit_length() - 1),
            "layout": "strong", "streams_total": total, "stream_ids": ids,
            "ms_per_step": round(wall, 4), "value": round(total * n / wall / 1e3, 1), "unit": "Msamples/s (input)",
            "roofline": roof(BYTES_DECIM * total * n, per[K_DECIMATE], decim_kernel_name(plan)),
            "verified": verify_step(ctx, x, ids, kind)}
    del x
    return line


def interp_kernel_name(ctx):
    """interpolate16_cen: K5w (interp_wave.h) unless the CONTEXT was told otherwise (set_option or the SDRHIP_INTERP_PATH it read
    when it was created: valu = K5)"""
    return "interp_kernel<4>" if ctx.option("interp_path", "auto") == "valu" else "interp_wave_kernel<4, 4>"


def verify_tx_step(ctx, rxf, kind, n, pipelined=False):
    """Behind the timed Tx region: the SAME call (dec_max_rows still at the sender's fecblk) on a fresh handle, whole-output
    digests per stream.  With the default `hash` input and bench.py's own geometry the expected digests are the committed ones
    (tests/golden/headline_golden.json "tx_bank8": oracle cm256_decode + the compiled reference's interpolate16_cen); otherwise
    the output must equal that of the dense decoder path of this library."""
    import hashlib

    import sdrdaemon_amd as sd
    import headline_inputs as hi

    S, F = rxf.shape[0], rxf.shape[1]

    def digests():
        tx = sd.TxPipe(ctx, S, hi.TX_LOG2_INTERP, pipelined=pipelined)
        iq = tx.process(rxf)
        if pipelined:  # (the first call delivers nothing; the batch comes out one call -- here: a flush -- late)
            assert iq.shape[1] == 0
            iq = tx.flush(device=rxf.device)
        ctx.synchronize()
        return [hashlib.sha256(iq[s].contiguous().cpu().numpy().tobytes()).hexdigest() for s in range(S)]

    c0 = ctx.counter("dec_rows_exceeded")
    got = digests()
    exceeded = ctx.counter("dec_rows_exceeded") - c0
    gold = None
    if kind == "hash":
        try:
            with open(os.path.join(ROOT, "tests", "golden", "headline_golden.json")) as f:
                T = json.load(f)["tx_bank8"]
            if (1 << T["log2n"]) == n and S == len(T["seeds"]) and F == T["frames"] and T["keep_seed"] == hi.TX_KEEP_SEED:
                gold = T["iq_sha256"]
        except Exception:
            gold = None
    what = "sha256 of every stream's whole interpolated output of one step"
    if gold is not None:
        return {"ok": got == gold and exceeded == 0, "streams": S, "what": what, "dec_rows_exceeded": exceeded,
                "against": "tests/golden/headline_golden.json tx_bank8 (oracle cm256_decode + compiled reference interpolate16_cen)"}
    ctx.set_option("dec_path", "dense")
    try:
        exp = digests()
    finally:
        ctx.set_option("dec_path", "syndrome")
    return {"ok": got == exp and exceeded == 0, "streams": S, "what": what, "dec_rows_exceeded": exceeded,
            "against": "the dense decoder path of this library on the same input (no committed digest for this geometry)"}


def _gold_bank(n, ids, kind="hash"):
    """the committed reference digests (tests/golden/headline_golden.json) of the bank that holds exactly these streams of this input
    kind (hash: counter-based noise; testsource: the TestSource bank's NCO, made by the oracle's restatement of it), or None"""
    if kind not in ("hash", "testsource"):
        return None, None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "headline_golden.json")) as f:
            H = json.load(f)
        for name in (("bank8", "bank64_25", "bank64") if kind == "hash" else ("ts_bank8",)):
            b = H[name]
            if (1 << b["log2n"]) == n and all((1000 + sid) in b["seeds"] for sid in ids):
                return b, [b["seeds"].index(1000 + sid) for sid in ids]
    except Exception:
        pass
    return None, None


def verify_decim(ctx, x, ids, kind):
    """configs[1] behind its timed region: decimate16_cen of the bank on fresh handles, whole-output digest per stream against the
    committed digests of the compiled reference (hash input, bench.py's geometry), else against the VALU kernel path."""
    import hashlib

    import sdrdaemon_amd as sd

    S, n = x.shape[0], x.shape[1]

    def digests(path):
        ctx.set_option("decim_path", path)
        try:
            d = sd.Decimators(ctx, S, sd.HB_EO1)
            y, _ = d.decimate(LOG2DECIM, sd.FC_CEN, 16, x)
            ctx.synchronize()
            return [hashlib.sha256(y[s].contiguous().cpu().numpy().tobytes()).hexdigest() for s in range(S)], d.last_plan()["path"]
        finally:
            ctx.set_option("decim_path", "auto")

    got, path = digests("auto")
    b, idx = _gold_bank(n, ids, kind)
    what = "sha256 of every stream's whole decimated output of one step"
    if b is not None:
        return {"ok": got == [b["dec_sha256"][i] for i in idx], "streams": S, "kernel_path": path, "what": what,
                "against": "tests/golden/headline_golden.json dec_sha256 (the compiled reference's decimate16_cen)"}
    exp, _ = digests("valu")
    return {"ok": got == exp, "streams": S, "kernel_path": path, "what": what,
            "against": "the VALU kernel path of this library on the same input (no committed digest for this geometry)"}


def verify_step(ctx, x, ids, kind, pipelined=False):
    """Behind the timed region: the SAME call on fresh handles (streams restart from the constructor state), whole output
    digests.  With the default `hash` input and bench.py's own geometry the expected digests are the committed ones made
    from the compiled reference decimator + the framer / encoder restatement (tests/golden/headline_golden.json); otherwise
    the frames must equal those of the VALU kernel path.  -> dict for the `verified` key."""
    import hashlib

    import sdrdaemon_amd as sd

    S, n = x.shape[0], x.shape[1]

    def frames_digests(path):
        ctx.set_option("decim_path", path)
        try:
            rx = sd.RxPipe(ctx, S, log2decim=LOG2DECIM, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=NB_FEC,
                           center_frequency_khz=435000, sample_rate=625000, pipelined=pipelined)
            fr = rx.process_view(x, tv_sec=1, tv_usec=0).torch()
            if pipelined:  # (the first call delivers nothing; the SECOND one delivers the first one's frames)
                assert fr.shape[1] == 0
                fr = rx.process_view(x, tv_sec=7, tv_usec=9).torch()
            ctx.synchronize()
            return [hashlib.sha256(fr[s].contiguous().cpu().numpy().tobytes()).hexdigest() for s in range(S)], rx.last_plan()["path"]
        finally:
            ctx.set_option("decim_path", "auto")

    got, path = frames_digests("auto")
    gold = None
    b, idx = _gold_bank(n, ids, kind)
    if b is not None:
        gold = [b["frames_sha256"][i] for i in idx]
    if gold is not None:
        return {"ok": got == gold, "against": "tests/golden/headline_golden.json %s(compiled reference decimate16_cen + framer / CM256 restatement)" %
                                              ("ts_bank8: the TestSource NCO samples by the oracle " if kind == "testsource" else ""),
                "streams": S, "kernel_path": path, "what": "sha256 of every stream's whole frame stream of one step"}
    exp, _ = frames_digests("valu")
    return {"ok": got == exp, "against": "the VALU kernel path of this library on the same input (no committed digest for this geometry)",
            "streams": S, "kernel_path": path, "what": "sha256 of every stream's whole frame stream of one step"}


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-execute the same command line under
    torch.distributed.run, one rank per GPU on this node, rendezvous on 127.0.0.1 (the container's host name may not resolve).
    nccl (= RCCL over xGMI) needs N visible GPUs; --backend gloo is the explicit dry run of the N > 1 path on a smaller box."""
    import socket
    import subprocess

    if args.backend == "nccl" and torch.cuda.device_count() < args.gpus:
        print("bench: --gpus %d but %d GPU(s) visible; pass --backend gloo to dry-run the N > 1 path with ranks sharing GPUs"
              % (args.gpus, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2-samples", type=int, default=25, help="samples per stream per step (default 2^25)")
    ap.add_argument("--streams", type=int, default=int(os.environ["SDRHIP_BENCH_STREAMS"]) if "SDRHIP_BENCH_STREAMS" in os.environ else None,
                    help="total streams of the job, sharded s -> rank s mod G (strong scaling).  Default: 64 = BASELINE configs[4] "
                         "whenever --gpus > 1 (SURVEY.md 8e: the same 64 streams at 2 / 4 / 8 GPUs), 8 streams on the one GPU at "
                         "--gpus 1 (the headline step); 0 = 8 streams per GPU whatever N (weak scaling); the environment variable "
                         "SDRHIP_BENCH_STREAMS sets the default for a driver that cannot add flags")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra single-GPU configurations of the `configs` key")
    ap.add_argument("--input", choices=["hash", "noise", "testsource"], default="hash",
                    help="hash: counter-based full-scale uniform noise (tests/signals.py) -- the input of the committed reference digests, "
                         "so the run can verify its own output; noise: torch.randint; testsource: the library's GPU TestSource bank")
    ap.add_argument("--no-verify", action="store_true", help="skip the output check behind the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--preroll-seconds", type=float, default=0.25,
                    help="untimed run-in of the same step before the W warm-up steps (the GPU's clocks ramp over the first "
                         "~60 ms of load: with a small W the timed steps would measure that ramp)")
    ap.add_argument("--timer-stride", type=int, default=0,
                    help="the HIP-event timers of the ROOFLINE kernel (the decimator) bracket every N-th step of the timed region (default: 4, "
                         "and 1 -- every launch -- when --steps < 50, so that the kernel average of a short run rests on all of its launches); "
                         "the encoder's timers stay on every 4th step (an event pair costs the stream ~3.5 us: tools/experiments_r06/timer_cost.py)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="diagnostic: leave the per-kernel HIP events out of the timed region (roofline fields become null)")
    ap.add_argument("--no-box-state", action="store_true", help="do not sample socket power / shader clock (amdsmi) beside the timed region")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: go through the N > 1 code path anyway (init_process_group, barriers, device-tensor all_reduce) -- "
                         "executes the nccl = RCCL branch on a one-GPU box")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to dry-run the N > 1 path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()
    if args.timer_stride <= 0:
        args.timer_stride = 1 if args.steps < 50 else 4
    if args.streams is None:
        # SDRHIP_BENCH_SCALE=1: this run is one point of a 1 / 2 / 4 / 8 sweep -- the N = 1 point is then the SAME job as the
        # others (SURVEY 8e's bank of 64 streams); without it the one-GPU run is the headline step (8 streams, config 5's per-GPU
        # share) and reports the 64-stream job as an extra `configs` line
        args.streams = 64 if (args.gpus > 1 or os.environ.get("SDRHIP_BENCH_SCALE") == "1") else 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain process (the way the driver starts the 1-GPU run): become the launcher of one rank per GPU
        sys.exit(relaunch_under_torchrun(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    assert world == args.gpus, "--gpus must equal WORLD_SIZE"

    import sdrdaemon_amd as sd
    from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_ENCODE

    dev = torch.device("cuda", local)
    ctx = sd.Context(local)
    n = 1 << args.log2_samples
    # streams are sharded one-per-stream across ranks, no data-path collective
    from sdrdaemon_amd import sharding

    if args.streams:
        assert args.streams >= world, "--streams must be >= the number of GPUs"
        ids = sharding.stream_ids_strong(rank, world, args.streams)  # stream s on rank s mod G
    else:
        ids = sharding.stream_ids(rank, world, STREAMS_PER_GPU)      # rank * 8 + s
    S = len(ids)
    ids_by_rank = [ids]
    if dist is not None:  # (reporting only, outside the timed region)
        ids_by_rank = [None] * world
        dist.all_gather_object(ids_by_rank, ids)
    x = make_input(ctx, dev, n, [1000 + sid for sid in ids], args.input)
    if rank == 0:
        print("bench: %d rank(s), layout %s" % (world, ("strong: %d streams in total, stream s on rank s mod %d (SURVEY 8e)" % (args.streams, world))
                                                if args.streams else "weak: %d streams per rank, stream id = rank * %d + s" % (S, S)), file=sys.stderr)
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
    # kernel-class timers (HIP events on the library's stream) over the timed region: every 4th step's launches -- an event pair
    # costs the stream ~2.5 us, four of them per step took 3 % off the step (tools/bench_rx_modes.py times the same step without)
    ctx.set_option("ktime_stride", max(4, args.timer_stride))
    ctx.set_option("ktime_stride_class", "%d:%d" % (K_DECIMATE, args.timer_stride))
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
    ctx.set_option("ktime_stride", 1)
    # the state of the box: the same step for 0.8 s more, UNTIMED, with the power / clock sampler beside it.  Not inside the timed
    # region: the SMU's socket power is a filtered value that takes ~0.4 s of load to settle (a 60-ms timed region reads the ramp:
    # 680 W on a box that settles at 1370 W), and the queries themselves should not sit beside the measurement.
    box_state = None
    if rank == 0 and not args.no_box_state:
        box = BoxState(local, period_s=0.01).start()
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < 0.8:
            for _ in range(10):
                step(0)
            torch.cuda.synchronize()
        box.stop()
        box.rows = box.rows[len(box.rows) // 2:]  # the settled half
        box_state = box.summary()
        if box_state is not None:
            box_state["over"] = "the second half of 0.8 s of the same step, back to back, right behind the timed region (untimed)"
    plan = rx.last_plan()
    kname = decim_kernel_name(plan)
    verified = None if args.no_verify else verify_step(ctx, x, ids, args.input)
    # the only collectives of the job: MAX of the elapsed time, SUM of the samples (8 bytes each, reporting only)
    elapsed, total_samples = sharding.aggregate(elapsed, float(S) * n * args.steps, dist,
                                                dev if args.backend == "nccl" else torch.device("cpu"), force=args.force_dist)

    if rank == 0:
        value = total_samples / elapsed / 1e6
        per_launch_samples = float(S) * n
        avg_ms = dec_ms / max(dec_n, 1) if dec_n else float("nan")
        achieved = BYTES_DECIM * per_launch_samples / (avg_ms * 1e-3) / 1e9
        res = {
            "metric": "IQ Msamples/s through decim+FEC-encode pipe; bit-exact vs CPU ref",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "preroll_steps": preroll,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong" if args.streams else "weak",
            "vs_baseline": None, "dtype": "int32",
            "collectives": ("%s: barrier x2 + all_reduce(MAX time, SUM samples) on %s tensors, world %d" %
                            (args.backend, "device" if args.backend == "nccl" else "host", world)) if dist is not None else "none (one process)",
            "data": "synthetic: %s, %s, HBM-resident before the timed region" %
                    ("uniform random full-scale int16 IQ (counter-based hash, tests/signals.py)" if args.input == "hash" else
                     "uniform random full-scale int16 IQ (torch.randint)" if args.input == "noise" else "GPU TestSource bank: 10 Msps CW, -20 dB, +100 kHz + 1 kHz x stream id",
                     ("%d streams in total, stream s on rank s mod %d" % (args.streams, world)) if args.streams else
                     ("%d streams/GPU (stream id = rank*%d + s)" % (S, S))),
            "config": {"workload": "configs[2] x %s: 10 Msps-shaped int16 IQ, decimate16_cen (EO1) + UDPSinkFEC framing + "
                                   "CM256 128+32 encode" % (("%d streams over %d GPU(s) (configs[4] when 64 over 8)" % (args.streams, world))
                                                            if args.streams else "%d streams/GPU" % S),
                       "streams_per_gpu": S, "streams_total": args.streams if args.streams else S * world,
                       "layout": ("strong: the fixed bank of %d streams, stream s on rank s mod %d (SURVEY 8e; the same job at every N)" % (args.streams, world))
                                 if args.streams else "weak: %d streams per rank, stream id = rank * %d + s (N = 8: BASELINE configs[4])" % (S, S),
                       "samples_per_stream_per_step": n, "log2decim": LOG2DECIM, "fcpos": "cen",
                       "nb_fec": NB_FEC, "hb_variant": "EO1", "frames_per_stream_per_step": frames // max(args.steps, 1), "output": "zero-copy view of the frame area",
                       "parallelism": "stream-sharded x%d, no data-path collective" % world,
                       "stream_ids_by_rank": ids_by_rank if world * S <= 64 else "rank r: %s" % ("r, r+G, ..." if args.streams else "8r .. 8r+7")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(per_launch_samples, kname),
                         "traffic_source": "profiles/traffic.json (PMC passes of tools/prof.sh on this command, committed; not measured by this run)",
                         "kernel": kname, "launches": dec_n, "timer_stride": args.timer_stride, "avg_launch_ms": round(avg_ms, 4), "plan": plan,
                         "algorithmic_bytes_per_launch": BYTES_DECIM * per_launch_samples,
                         "pipe_gbps_config3": round(BYTES_CONFIG3 * value * 1e6 / 1e9 / world, 1),
                         # the PATH the metric names (decimate + framing + FEC encode, config-3 algorithmic bytes) against the same peak
                         "path_frac": round(BYTES_CONFIG3 * value * 1e6 / 1e9 / world / HBM_PEAK_GBS, 4),
                         "arrangement": "rx_direct: matrix-core decimator stores in the frame layout; CM256 encoder = additive FFT (gf_encode128_fft_kernel); 2 launches per step",
                         "fec_encode_avg_launch_ms": round(fec_ms / max(fec_n, 1), 4), "fec_encode_launches": fec_n},
        }
        res["verified"] = verified
        # the state of the box the line was measured on: both dominant kernels are limited by the board's power cap,
        # so ms_per_step on another box scales with what that box's silicon and cooling make of the cap
        res["box"] = box_state
        if isinstance(box_state, dict) and box_state.get("power_w"):
            # energy of one step at the sampled socket power: the figure kernel changes are judged by on a power-limited box
            res["box"]["energy_mj_per_step"] = round(float(box_state["power_w"]) * res["ms_per_step"], 1)
        lane_ops = pmc_valu_lane_ops(per_launch_samples, kname)
        if lane_ops and dec_n:
            # secondary figure of SURVEY.md 8(d): integer VALU issue, every op counted at the 16-lane / clk rate
            tl = lane_ops / (avg_ms * 1e-3) / 1e12
            res["roofline"]["valu"] = {"achieved": round(tl, 2), "peak": round(VALU_PEAK_TLANEOPS, 2), "unit": "T lane-ops/s",
                                       "frac": round(tl / VALU_PEAK_TLANEOPS, 4), "lane_ops_per_sample": round(lane_ops / per_launch_samples, 2)}
        if world == 1 and not args.no_configs:
            res["configs"] = extra_configs(ctx, dev, x, args.input, ids)
            if not args.streams:
                del x
                res["configs"].append(strong_layout_at_one_gpu(ctx, dev, n, args.input))
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
