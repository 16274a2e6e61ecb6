// Probe (GPU box): the load burst of the FFT encoder / decoder (K3f / K4f) on its own.  1024 workgroups of four waves, one frame of
// 128 x 512 bytes per workgroup; wave (ch, hf) loads 64 blocks, a dword per lane, all 64 loads in flight -- exactly the kernels' access
// pattern -- and optionally copies them out (the decoder's copy stores).  Variants: the 4-byte offset of the payload behind the block
// header (the kernels' 256-byte pieces then straddle three 128-byte lines) against aligned pieces; nt loads; 16 bytes per lane (a quarter
// wave per block); fewer workgroups per CU.  Between the timed launches a 1 GiB store stream flushes the caches like the interpolator does.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/fec_burst_probe.hip -o /tmp/fec_burst_probe && /tmp/fec_burst_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void flush(uint4_t *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint4_t){1u, 2u, 3u, (unsigned)i};
}

// MODE bit 0: aligned (no +4), bit 1: nt loads, bit 2: copy stores, bit 3: nt stores
template <int MODE> __global__ __launch_bounds__(256, 4) void burst4(const unsigned char *rx, unsigned char *pay, unsigned *out, int pitch_blocks)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, ch = wv & 1, hf = wv >> 1;
    const unsigned char *fb = rx + (size_t)blockIdx.x * pitch_blocks * 512 + ((MODE & 1) ? 0 : 4);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(fb), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc(pay + (size_t)blockIdx.x * 127 * 508, 0, 0x7fffffff, 0x00020000);
    int col = ch * 64 + lane;
    if (!(MODE & 1) && col > 126) col = 126;
    const unsigned lc4 = 4u * (unsigned)col;
    unsigned d[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) d[i] = __builtin_amdgcn_raw_buffer_load_b32(r, lc4, (64 * hf + i) * 512, (MODE & 2) ? 2 : 0);
    if (MODE & 4) {
        const unsigned st4 = 4u * (unsigned)(col > 126 ? 126 : col);
#pragma unroll
        for (int i = 0; i < 64; ++i)
            if (64 * hf + i >= 1) __builtin_amdgcn_raw_buffer_store_b32(d[i], w, st4, (64 * hf + i - 1) * 508, (MODE & 8) ? 2 : 0);
    }
    unsigned acc = 0u;
#pragma unroll
    for (int i = 0; i < 64; ++i) acc ^= d[i];
    if (acc == 0x12345u) out[blockIdx.x] = acc;
}

// 16 bytes per lane: a quarter wave per block, 16 loads per wave (64 blocks x 256 bytes), aligned
template <int MODE> __global__ __launch_bounds__(256, 4) void burst16(const unsigned char *rx, unsigned char *pay, unsigned *out, int pitch_blocks)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, ch = wv & 1, hf = wv >> 1;
    const unsigned char *fb = rx + (size_t)blockIdx.x * pitch_blocks * 512;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(fb), 0, 0x7fffffff, 0x00020000);
    const unsigned lo = (unsigned)(lane >> 4) * 512u + (unsigned)ch * 256u + 16u * (unsigned)(lane & 15);
    uint4_t d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = __builtin_amdgcn_raw_buffer_load_b128(r, lo, (64 * hf + 4 * i) * 512, (MODE & 2) ? 2 : 0);
    uint4_t acc = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= d[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = acc.x;
}

template <class K> float timeit(K &&launch, uint4_t *fl, size_t fln)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> t;
    for (int it = 0; it < 12; ++it) {
        flush<<<2048, 256>>>(fl, fln);
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const int F = 1024;
    unsigned char *rx, *pay; unsigned *out; uint4_t *fl;
    const size_t fln = ((size_t)1 << 30) / 16;
    (void)hipMalloc(&rx, (size_t)F * 160 * 512 + 4096);
    (void)hipMalloc(&pay, (size_t)F * 127 * 508 + 4096);
    (void)hipMalloc(&out, 1 << 20);
    (void)hipMalloc(&fl, fln * 16);
    (void)hipMemset(rx, 3, (size_t)F * 160 * 512 + 4096);
    const double mb = F * 65536.0 / 1e6;
    for (int pitch : {128, 160}) {
        printf("frame pitch %d blocks (%d bytes)\n", pitch, pitch * 512);
#define RUN(name, kern) { float ms = timeit([&] { kern<<<F, 256>>>(rx, pay, out, pitch); }, fl, fln); printf("  %-58s %7.2f us  %6.0f GB/s (reads)\n", name, ms * 1e3, mb / ms / 1e3); }
        RUN("dword loads, +4 (the kernels' pattern)", burst4<0>);
        RUN("dword loads, aligned", burst4<1>);
        RUN("dword loads, +4, nt", burst4<2>);
        RUN("dword loads, aligned, nt", burst4<3>);
        RUN("dword loads, +4, copy stores", burst4<4>);
        RUN("dword loads, aligned, copy stores", burst4<5>);
        RUN("dword loads, +4, nt, copy stores", burst4<6>);
        RUN("dword loads, +4, nt, nt copy stores", burst4<14>);
        RUN("dword loads, aligned, nt, nt copy stores", burst4<15>);
        RUN("16-byte loads, aligned", burst16<0>);
        RUN("16-byte loads, aligned, nt", burst16<2>);
    }
    // the same without the flush in front (input resident in the Infinity Cache: the encoder's case)
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int al = 0; al < 2; ++al) {
            for (int i = 0; i < 5; ++i) { if (al) burst4<1><<<F, 256>>>(rx, pay, out, 160); else burst4<0><<<F, 256>>>(rx, pay, out, 160); }
            (void)hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) { if (al) burst4<1><<<F, 256>>>(rx, pay, out, 160); else burst4<0><<<F, 256>>>(rx, pay, out, 160); }
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("back to back (cache-resident), pitch 160, dword loads, %s: %7.2f us\n", al ? "aligned" : "+4", ms / 20 * 1e3);
        }
    }
    return 0;
}
