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

#include "frame_pack_body.h"

__global__ __launch_bounds__(256) void frame_pack_kernel(FrameArgs a)
{
    frame_pack_wg(a, (int)blockIdx.y, blockIdx.x, gridDim.x);
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
