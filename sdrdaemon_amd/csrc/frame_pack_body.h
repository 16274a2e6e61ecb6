// frame_pack_body.h -- K2 (UDPSinkFEC::write framing) as device code shared by frame_kernels.hip and gf_kernels.hip.
// Include inside namespace sdrhip { namespace { ... } }.
#pragma once
// one workgroup of K2: block bx of nbx of stream `stream` (a device function: gf_kernels.hip runs it in the encoder's launch too)
__device__ __forceinline__ void frame_pack_wg(const FrameArgs &a, int stream, unsigned bx, unsigned nbx)
{
    const unsigned *src = a.in + (size_t)stream * a.in_stride;
    unsigned *dst = a.out + (size_t)stream * a.out_stride;
    const unsigned fdw = (unsigned)a.frame_blocks * 128u;
    // payload
    const size_t nskip = a.skip_to - a.skip_from, ncopy = a.n - nskip;
    for (size_t j = (size_t)bx * 256 + threadIdx.x; j < ncopy; j += (size_t)nbx * 256) {
        const size_t k = j < a.skip_from ? j : j + nskip;
        const uint64_t g = a.frame_sample_base + k;
        const uint64_t f = g / 16129u;
        const unsigned w = (unsigned)(g - f * 16129u);
        const unsigned b = w / 127u, i = w - b * 127u;
        dst[(size_t)f * fdw + (size_t)(1u + b) * 128u + 1u + i] = SDRHIP_STREAM_LOAD(src + k);
    }
    // meta block + super block headers of the frames this call starts: frame fi by workgroup fi mod nbx
    if (threadIdx.x < 128) {
        const unsigned t = threadIdx.x;
        for (int fi = (int)bx; fi < a.meta_count; fi += (int)nbx) {
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

