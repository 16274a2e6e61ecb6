// decim_kernels.hip -- cascaded integer half-band decimators for gfx950 (MI355X).
//
// Replaces the per-sample ring-buffer loops of Decimators::decimate{8..64}_{inf,sup} and
// decimate{2..64}_cen (Decimators.cpp:94-1305) over IntHalfbandFilterEO1/DB<64>::myDecimate
// (IntHalfbandFilterEO1.h:34-42,100-147 / IntHalfbandFilterDB.h:32-49,79-107).
//
// Design (DESIGN.md "K1"):
//  * grid = (segments, streams); a 256-thread workgroup walks one segment of one stream in
//    passes of 2048 first-stage inputs.  The inputs of every half-band stage live in LDS as
//    four planes ({even, odd} input parity x {I, Q}: the polyphase split of the filter)
//    behind 32 entries of history that are carried from invocation to invocation, so nothing
//    is recomputed inside a segment.
//  * output k of a stage = 32-tap FIR over the odd plane + centre tap from the even plane.
//    A thread produces R consecutive outputs of ONE component from a register window of
//    R + 32 plane entries (ds_read_b128); lanes 2j / 2j+1 own I / Q of the same outputs, the
//    plane stride is 8 mod 16 LDS slots, so every 16-lane read group is conflict-free.
//  * multi-rate schedule: stage 0 runs every pass (R = 8), stage s >= 1 runs every 2^(s-1)
//    passes when its planes hold 512 fresh entries (R = 4): all 256 threads are busy in
//    every stage, no wave idles while a narrow late stage runs, and the workgroup needs only
//    ~35 KB of LDS (4 workgroups = 16 waves per CU).
//  * the first stage of the centred modes reads raw int16 samples packed two per dword and
//    uses v_dot2c_i32_i16 (2 taps per lane-op, exact: |acc| < 2^30); later stages use
//    v_add_u32 + v_mad_i32_i24 (stage inputs are |x| <= 2^18, so the 24-bit multiply is exact
//    in the low 32 bits = the reference's wrap-around int32 arithmetic).
//  * the next pass's global loads are issued before the current pass is computed.
//  * segment 0 takes the filter histories from the bank state; every other segment rebuilds
//    them by running 64 * 2^L raw samples ahead of its first sample (>= the 62 * (2^L - 1)
//    samples that reach the last stage's history) with stores suppressed.  The workgroup of
//    the last segment writes the new state (double buffered).
#include "sdrhip_internal.h"

#include "decim_body.h"

namespace sdrhip {
namespace {

// FC: 0 inf, 1 sup (fs/4 rotate + sum of four raw samples first), 2 cen
template <int L, int FC, bool PACK16> __global__ __launch_bounds__(NT) void decim_kernel(DecimArgs a)
{
    constexpr int PRAW = P0 << (FC == 2 ? 0 : 2);
    __shared__ __attribute__((aligned(16))) int lds[DecimLds<L, FC, PACK16>::dwords];
    const int seg = blockIdx.x;
    const size_t seg_raw = (size_t)a.nsub_per_seg * PRAW;
    const size_t seg_start = (size_t)seg * seg_raw;
    size_t seg_end = seg_start + seg_raw;
    if (seg_end > a.n_used) seg_end = a.n_used;
    decim_piece<L, FC, PACK16>(a, lds, blockIdx.y, seg_start, seg_end, seg == 0, seg == a.nseg - 1, seg, a.nseg);
}

template <int L, int FC, bool PACK16> hipError_t launch_variant(const DecimArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL((decim_kernel<L, FC, PACK16>), dim3(a.nseg, a.nstreams), dim3(NT), 0, stream, a);
    return hipGetLastError();
}

} // namespace

void plan_decimate(int log2decim, int fcpos, size_t n_used, int nstreams, int *nsub_per_seg, int *nseg)
{
    const bool cen = (fcpos == 2);
    const size_t praw = cen ? (size_t)P0 : (size_t)P0 * 4;
    size_t npass = (n_used + praw - 1) / praw;
    if (npass == 0) npass = 1;
    // a segment is a whole number of schedule periods (2^(NS-2) passes) so that full-rate
    // stages never flush early; >= 16 passes keep the warm-up (64 * 2^L samples) below ~3 %.
    const int ns = cen ? log2decim : log2decim - 2;
    size_t period = (size_t)1 << (ns > 2 ? ns - 2 : 0);
    size_t per = period < 32 ? 32 : period; // warm-up = 64 * 2^L of 32 * 2048 samples: 1.6 % at L = 4
    // fewer passes per segment when the call is too short to fill 256 CUs x 4 workgroups
    while (per > period && ((npass + per - 1) / per) * (size_t)nstreams < 2048) per >>= 1;
    while (per > 1 && ((npass + per - 1) / per) * (size_t)nstreams < 256) per >>= 1;
    // the warm-up region must not reach before the stream start
    const size_t wraw = (size_t)64 << log2decim;
    while (per * praw < wraw) per <<= 1;
    *nsub_per_seg = (int)per;
    *nseg = (int)((npass + per - 1) / per);
}

hipError_t launch_decimate(int log2decim, int fcpos, bool pack16, const DecimArgs &a, hipStream_t stream)
{
#define SDRHIP_CEN(L_)                                                                                          \
    case L_:                                                                                                    \
        return pack16 ? launch_variant<L_, 2, true>(a, stream) : launch_variant<L_, 2, false>(a, stream);
#define SDRHIP_ROT(L_, FC_)                                                                                     \
    case L_:                                                                                                    \
        return launch_variant<L_, FC_, false>(a, stream);
    if (fcpos == 2) {
        switch (log2decim) {
            SDRHIP_CEN(1) SDRHIP_CEN(2) SDRHIP_CEN(3) SDRHIP_CEN(4) SDRHIP_CEN(5) SDRHIP_CEN(6)
        }
    } else if (fcpos == 0) {
        switch (log2decim) { SDRHIP_ROT(3, 0) SDRHIP_ROT(4, 0) SDRHIP_ROT(5, 0) SDRHIP_ROT(6, 0) }
    } else {
        switch (log2decim) { SDRHIP_ROT(3, 1) SDRHIP_ROT(4, 1) SDRHIP_ROT(5, 1) SDRHIP_ROT(6, 1) }
    }
#undef SDRHIP_CEN
#undef SDRHIP_ROT
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// filter-less variants: decimate1 (Decimators.cpp:22-35), decimate2_inf/sup (:38-91),
// decimate4_inf/sup (:127-170).  One thread per group of four input samples.
__global__ void decim_simple_kernel(int log2decim, int fcpos, const int16_t *in, size_t in_stride, int16_t *out,
                                    size_t out_stride, size_t n_in, int norm, int trunk)
{
    const int stream = blockIdx.y;
    const unsigned *src = reinterpret_cast<const unsigned *>(in) + (size_t)stream * in_stride;
    unsigned *dst = reinterpret_cast<unsigned *>(out) + (size_t)stream * out_stride;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (log2decim == 0) {
        for (size_t i = g; i < n_in; i += (size_t)gridDim.x * blockDim.x) {
            unsigned v = SDRHIP_STREAM_LOAD(src + i);
            int a = (short)(v & 0xffff), b = (int)v >> 16;
            dst[i] = (((unsigned)a << norm) & 0xffffu) | (((unsigned)b << norm) << 16);
        }
        return;
    }
    const size_t n_resize = n_in >> log2decim;
    const size_t ngroups = n_in / 4;
    for (size_t q = g; q < (n_in + 3) / 4; q += (size_t)gridDim.x * blockDim.x) {
        if (q >= ngroups) { // out.resize() elements the loop never writes stay zero (fresh vector)
            if (log2decim == 1 && 2 * q < n_resize) dst[2 * q] = 0;
            continue;
        }
        const unsigned v0 = SDRHIP_STREAM_LOAD(src + 4 * q), v1 = SDRHIP_STREAM_LOAD(src + 4 * q + 1), v2 = SDRHIP_STREAM_LOAD(src + 4 * q + 2), v3 = SDRHIP_STREAM_LOAD(src + 4 * q + 3);
        const int I0 = (short)(v0 & 0xffff), Q0 = (int)v0 >> 16, I1 = (short)(v1 & 0xffff), Q1 = (int)v1 >> 16;
        const int I2 = (short)(v2 & 0xffff), Q2 = (int)v2 >> 16, I3 = (short)(v3 & 0xffff), Q3 = (int)v3 >> 16;
        if (log2decim == 1) {
            int xa, ya, xb, yb;
            if (fcpos == 0) { xa = I0 - Q1; ya = Q0 + I1; xb = Q3 - I2; yb = -Q2 - I3; }
            else { xa = Q0 - I1; ya = -I0 - Q1; xb = I3 - Q2; yb = I2 + Q3; }
            dst[2 * q] = final_pack(xa, ya, norm, trunk);
            dst[2 * q + 1] = final_pack(xb, yb, norm, trunk);
        } else {
            int x, y;
            if (fcpos == 0) { x = I0 - Q1 + Q3 - I2; y = Q0 - Q2 + I1 - I3; }
            else { x = Q0 - I1 - Q2 + I3; y = -I0 - Q1 + I2 + Q3; }
            dst[q] = final_pack(x, y, norm, trunk);
        }
    }
}

hipError_t launch_decimate_simple(int log2decim, int fcpos, const int16_t *in, size_t in_stride, int16_t *out,
                                  size_t out_stride, size_t n_in, int nstreams, int norm, int trunk, hipStream_t stream)
{
    size_t work = log2decim == 0 ? n_in : (n_in + 3) / 4;
    size_t blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(decim_simple_kernel, dim3((unsigned)blocks, nstreams), dim3(256), 0, stream, log2decim, fcpos, in,
                       in_stride, out, out_stride, n_in, norm, trunk);
    return hipGetLastError();
}

} // namespace sdrhip
