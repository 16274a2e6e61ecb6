"""The C++ host side of BASELINE config 5 (tools/sdrhip_multi.hip: one host thread + one sdrhip context per device, stream s on
worker s mod G, no exchange -- VERDICT r5 missing #5): built on the GPU box with hipcc, run with one worker and with three workers
(three contexts on the one GPU of the box: the dealing and the concurrency of the multi-device shape, on the hardware that is here);
every stream's frames must not depend on the dealing, and must be the frames the Python mirror of the same pipe produces (which the
parity suite pins to the oracle and the reference digests)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv1a(b):
    h = 1469598103934665603
    for chunk in np.frombuffer(b, np.uint8).reshape(-1, 1 << 16) if len(b) % (1 << 16) == 0 else [np.frombuffer(b, np.uint8)]:
        for v in chunk.tolist():
            h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("multi") / "sdrhip_multi")
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++14", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "sdrhip_multi.hip"), "-L" + os.path.join(ROOT, "sdrdaemon_amd"), "-lsdrhip",
           "-Wl,-rpath," + os.path.join(ROOT, "sdrdaemon_amd"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def _run(binary, devices, streams=6, log2n=20, steps=2, warmup=1):
    r = subprocess.run([binary, "--devices", devices, "--streams", str(streams), "--log2-samples", str(log2n), "--steps", str(steps), "--warmup", str(warmup)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_streams_do_not_depend_on_the_dealing_and_match_the_python_pipe(binary):
    import sdrdaemon_amd as sd

    one = _run(binary, "0")
    three = _run(binary, "0,0,0")
    assert one["workers"] == 1 and three["workers"] == 3 and three["devices"] == [0, 0, 0]
    assert one["frames_per_stream_per_step"] == three["frames_per_stream_per_step"] == 4  # (3 x 65 536 decimated samples: 12 frames, 8 of them before the last call)
    assert len(one["stream_fnv"]) == 6 and one["stream_fnv"] == three["stream_fnv"]
    assert len(set(one["stream_fnv"])) == 6  # (six different carriers)
    assert three["value"] > 0 and three["streams_total"] == 6
    # the same three calls through the Python mirror: stream 2's frames of the last call
    ctx = sd.Context(0)
    ts = sd.TestSource(ctx, 1)
    assert ts.configure("srate=10000000,dfp=%d,power=20" % (100000 + 1000 * 2))
    x = ts.read(1 << 20)
    rx = sd.RxPipe(ctx, 1, log2decim=4, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=32, center_frequency_khz=435000, sample_rate=625000)
    for _ in range(3):
        fr = rx.process_view(x, tv_sec=1, tv_usec=0).torch().clone()
    ctx.synchronize()
    b = fr[0].contiguous().cpu().numpy().tobytes()
    assert fr.shape[1] == one["frames_per_stream_per_step"]
    assert "%016x" % _fnv1a(b) == one["stream_fnv"][2]
