// interp_kernels.hip -- cascaded integer half-band interpolators for gfx950 (MI355X).
//
// Replaces Interpolators::interpolate{2..64}_cen (Interpolators.cpp:23-606) over
// IntHalfbandFilterEO1/DB<64|32|16>::myInterpolate (IntHalfbandFilterEO1.h:44-65,149-168;
// DB twin IntHalfbandFilterDB.h:51-72,109-128 -- identical arithmetic):
//     v[2m]   = u[m - O/4]
//     v[2m+1] = (sum_{i < O/4} c[i] * (u[m - (O/2 - 1) + i] + u[m - i])) >> 13
// stage orders 64, 32, 16, 16, 16 (Interpolators.h:31-33); int32 between stages, int16
// truncation at the end.  interpolate64_cen is reproduced as the reference has it: five
// stages and 32 zero samples after every 32 outputs (Interpolators.cpp:363-606).
//
// Design (DESIGN.md "K5"), the mirror image of the decimator kernel:
//  * grid = (segments, streams); a 256-thread workgroup walks its segment in macro-cycles of
//    512 inputs.  The inputs of every stage sit in LDS (I and Q planes, int32, plane stride
//    8 mod 16 slots) behind 32 entries of history.
//  * the cascade is an expanding tree, walked depth first: one invocation of stage s
//    (512 inputs) feeds two invocations of stage s+1, so every invocation keeps all 256
//    threads busy: lanes 2j / 2j+1 compute the 8 outputs of 4 consecutive inputs of I / Q;
//    the last stage takes 1024 inputs per invocation and handles both components per thread
//    so that it can pack int16 I/Q and store 2 x 16 bytes per thread without a lane exchange.
//  * segment 0 takes the histories from the bank state, other segments rebuild them from the
//    64 preceding inputs (the cascade's memory is 43 inputs) with stores suppressed.
#include "interp_body.h"
#include "interp_wave.h"

namespace sdrhip {
namespace {

// L = log2 interpolation (6 = the reference's 5-stage + zero stuffing variant)
template <int L> __global__ __launch_bounds__(NT) void interp_kernel(InterpArgs a)
{
    __shared__ __attribute__((aligned(16))) int lds[IGeo<(L == 6) ? 5 : L>::ldsDw];
    interp_segment<L>(a, blockIdx.x, blockIdx.y, lds);
}

// K5w: one time slice of one stream per wave, no barrier (interp_wave.h).
// WPW waves per workgroup, each on a segment of its own (no barrier, no shared data: a workgroup is only a way of starting WPW
// waves at the same moment, which puts their clusters of input loads at the same moment -- see interp_wave_segment; measured
// 6 % on interpolate32, 0..1 % on the others, tools/experiments_r04 batch 20).  Launches too small to fill the chip with
// workgroups of four keep one wave per workgroup.
template <int L, int WPW> __global__ __launch_bounds__(WNT * WPW) void interp_wave_kernel(InterpArgs a)
{
    __shared__ __attribute__((aligned(16))) int lds[WPW][WGeo<(L == 6) ? 5 : L>::ldsDw];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int seg = (int)blockIdx.x * WPW + w;
    if (seg >= a.nseg) return;
    interp_wave_segment<L>(a, seg, blockIdx.y, lds[w]);
}
// ... with its input gathered through the Tx decoder's position map (InterpArgs::gmap)
template <int L, int WPW> __global__ __launch_bounds__(WNT * WPW) void interp_wave_gather_kernel(InterpArgs a)
{
    __shared__ __attribute__((aligned(16))) int lds[WPW][WGeo<(L == 6) ? 5 : L>::ldsDw];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int seg = (int)blockIdx.x * WPW + w;
    if (seg >= a.nseg) return;
    interp_wave_segment<L, true>(a, seg, blockIdx.y, lds[w]);
}

template <int L> hipError_t launch_w(const InterpArgs &a, hipStream_t stream)
{
    constexpr int WPW = 4;
    const bool four = (size_t)a.nseg * (size_t)a.nstreams >= 4096;
    if (a.gmap) {
        if (four) hipLaunchKernelGGL((interp_wave_gather_kernel<L, WPW>), dim3((a.nseg + WPW - 1) / WPW, a.nstreams), dim3(WNT * WPW), 0, stream, a);
        else hipLaunchKernelGGL((interp_wave_gather_kernel<L, 1>), dim3(a.nseg, a.nstreams), dim3(WNT), 0, stream, a);
    } else if (four)
        hipLaunchKernelGGL((interp_wave_kernel<L, WPW>), dim3((a.nseg + WPW - 1) / WPW, a.nstreams), dim3(WNT * WPW), 0, stream, a);
    else
        hipLaunchKernelGGL((interp_wave_kernel<L, 1>), dim3(a.nseg, a.nstreams), dim3(WNT), 0, stream, a);
    return hipGetLastError();
}

template <int L> hipError_t launch_l(const InterpArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL((interp_kernel<L>), dim3(a.nseg, a.nstreams), dim3(NT), 0, stream, a);
    return hipGetLastError();
}

} // namespace

void plan_interpolate(int log2interp, size_t n_in, int nstreams, int *nsub_per_seg, int *nseg)
{
    const size_t ci = log2interp == 1 ? 1024 : MC;
    size_t nsub = (n_in + ci - 1) / ci;
    if (nsub == 0) nsub = 1;
    size_t per = 16; // warm-up is 64 inputs: 16 macro-cycles per segment keep its cost below 1 %
    while (per > 1 && ((nsub + per - 1) / per) * (size_t)nstreams < 2048) per >>= 1;
    *nsub_per_seg = (int)per;
    *nseg = (int)((nsub + per - 1) / per);
}

// K5w: blocks of 128 inputs, segments of at most 2 x WPAIRS = 16 blocks (the wave parks its whole segment's input in registers at
// its start), an even number of them (the input comes back a PAIR of blocks at a time; an odd rest runs the guarded path)
void plan_interpolate_wave(int log2interp, size_t n_in, int nstreams, int n_cu, size_t seg_override, int *nsub_per_seg, int *nseg)
{
    (void)log2interp; (void)nstreams; (void)n_cu;
    const size_t maxper = 2 * (size_t)WPAIRS;
    size_t nsub = (n_in + WB - 1) / WB;
    if (nsub == 0) nsub = 1;
    size_t per = seg_override ? (seg_override + WB - 1) / WB : maxper;
    if (per > maxper) per = maxper;
    if (per > 1) per &= ~(size_t)1;
    if (per < 1) per = 1;
    *nsub_per_seg = (int)per;
    *nseg = (int)((nsub + per - 1) / per);
}

hipError_t launch_interpolate_wave(int log2interp, const InterpArgs &a, hipStream_t stream)
{
    switch (log2interp) {
    case 2: return launch_w<2>(a, stream);
    case 3: return launch_w<3>(a, stream);
    case 4: return launch_w<4>(a, stream);
    case 5: return launch_w<5>(a, stream);
    case 6: return launch_w<6>(a, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_interpolate(int log2interp, const InterpArgs &a, hipStream_t stream)
{
    switch (log2interp) {
    case 1: return launch_l<1>(a, stream);
    case 2: return launch_l<2>(a, stream);
    case 3: return launch_l<3>(a, stream);
    case 4: return launch_l<4>(a, stream);
    case 5: return launch_l<5>(a, stream);
    case 6: return launch_l<6>(a, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace sdrhip
