"""sdrdaemon_amd -- MI355X-native engine for sdrdaemon's DSP/FEC hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/sdrhip.h, built into
libsdrhip.so) and engine.py (host-side mirror of the reference's interface).
"""
from ._lib import (BLOCK_BYTES, FC_CEN, FC_INF, FC_SUP, HB_DB, HB_EO1, MEM_DEVICE, MEM_HOST, NB_ORIGINAL,  # noqa: F401
                   SAMPLES_PER_BLOCK, SAMPLES_PER_FRAME, UDPSIZE, SdrHipError)
from .engine import (CM256, Context, Decimators, Downsampler, Interpolators, RxPipe, TestSource, TxPipe, Upsampler,  # noqa: F401
                     device_count, fec_decode_frames, fec_encode_frames)
