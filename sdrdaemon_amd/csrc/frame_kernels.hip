// frame_kernels.hip -- K2: UDPSinkFEC::write framing (UDPSinkFEC.cpp:79-191) as a kernel of its own.
//
// Takes the decimated samples of a call in stream order ([stream][n] IQ dwords) and lays them out as super blocks:
// sample g of a stream (counted from the start of the frame that was being filled when the call began) goes to
// dword 1 + g % 127 of super block 1 + (g % 16129) / 127 of frame g / 16129 (UDPSinkFEC.cpp:134-155); the frames
// that START in this call get block 0 = {header, 24-byte MetaDataFEC + CRC, zero fill} (:87-132) and the
// {frameIndex, blockIndex, 0} headers of blocks 1..127 (:150-152).  One thread per sample: loads and stores are
// coalesced (consecutive lanes write consecutive dwords, with a one-dword skip at the block boundaries).
//
// Used by the Rx pipe behind the matrix-core decimator (which writes plain stream-order output at full speed; the
// scattered 4-byte stores of a fused epilogue cost it more than this pass over 1/2^L of the data) and for the
// filter-less settings (decim 0, inf / sup 2 and 4), which have no cascade kernel to fuse into.
#include "sdrhip_internal.h"

namespace sdrhip {
namespace {

__global__ __launch_bounds__(256) void frame_pack_kernel(FrameArgs a)
{
    const int stream = blockIdx.y;
    const unsigned *src = a.in + (size_t)stream * a.in_stride;
    unsigned *dst = a.out + (size_t)stream * a.out_stride;
    const unsigned fdw = (unsigned)a.frame_blocks * 128u;
    // payload
    const size_t nskip = a.skip_to - a.skip_from, ncopy = a.n - nskip;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < ncopy; j += (size_t)gridDim.x * 256) {
        const size_t k = j < a.skip_from ? j : j + nskip;
        const uint64_t g = a.frame_sample_base + k;
        const uint64_t f = g / 16129u;
        const unsigned w = (unsigned)(g - f * 16129u);
        const unsigned b = w / 127u, i = w - b * 127u;
        dst[(size_t)f * fdw + (size_t)(1u + b) * 128u + 1u + i] = SDRHIP_STREAM_LOAD(src + k);
    }
    // meta block + super block headers of the frames this call starts: frame fi by workgroup fi mod gridDim.x
    if (threadIdx.x < 128) {
        const unsigned t = threadIdx.x;
        for (int fi = blockIdx.x; fi < a.meta_count; fi += gridDim.x) {
            unsigned w[6];
            frame_meta_words(a.meta_w, a.meta_idx0, a.meta_rate, fi, w); // per-frame time stamp + CRC (wave-uniform)
            unsigned mw = 0u; // dword t of block 0 behind the header: the 24-byte MetaDataFEC, then zeros
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (t == (unsigned)k + 1u) mw = w[k];
            unsigned *fr = dst + (size_t)(a.meta_first + fi) * fdw;
            const unsigned fidx = (a.meta_frame_count0 + (unsigned)fi) & 0xffffu;
            fr[t] = t == 0 ? fidx : mw; // block 0: 512 bytes = 128 dwords
            if (t >= 1) fr[(size_t)t * 128] = fidx | (t << 16);
        }
    }
}

} // namespace

hipError_t launch_frame_pack(const FrameArgs &a, int nstreams, hipStream_t stream)
{
    size_t blocks = (a.n - (a.skip_to - a.skip_from) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(frame_pack_kernel, dim3((unsigned)blocks, nstreams), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace sdrhip
