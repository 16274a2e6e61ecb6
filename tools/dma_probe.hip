// Probe (GPU box): HBM -> LDS streaming with global_load_lds_dwordx4 (1 KiB per wave instruction) against register loads, in
// the geometry of the matrix-core decimator: 992 waves (248 workgroups x 4, one wave per SIMD), every wave walks 8 spans of
// SPAN bytes, one 1-KiB piece of one span per instruction (span = piece % 8), a bounded number of pieces in flight.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int INFL, int NT, int ORDER> __global__ __launch_bounds__(256) void dma_kernel(const char *in, unsigned *out, size_t span, int pieces)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * 32 * 1024];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wv));
    const unsigned ring = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(__attribute__((address_space(3))) char *)lds + wv * 32768));
    const char *wbase = in + (size_t)gw * 8 * span;
    unsigned voff = 16u * lane;
    unsigned acc = 0;
    // piece t: span t % 8, offset (t / 8) KiB; ORDER 1: the wave starts at span gw % 8 (de-phased)
    for (int t = 0; t < pieces; ++t) {
        const int sp = ORDER ? ((t + gw) & 7) : (t & 7);
        const unsigned long long b = (unsigned long long)(wbase + (size_t)sp * span + (size_t)(t >> 3) * 1024);
        const unsigned m = ring + (unsigned)(t & 31) * 1024u;
        if (NT)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 nt" ::"v"(voff), "s"(m), "s"(b) : "memory");
        else
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(m), "s"(b) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL) : "memory");
        if ((t & 7) == 7) acc += *reinterpret_cast<unsigned *>(lds + wv * 32768 + 4 * lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345u) out[gw] = acc;
}

template <int INFL> __global__ __launch_bounds__(256) void reg_kernel(const char *in, unsigned *out, size_t span, int pieces)
{
    const int lane = threadIdx.x & 63;
    const int gw = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    const char *wbase = in + (size_t)gw * 8 * span + 16 * lane;
    uint4_t acc = {0u, 0u, 0u, 0u};
    for (int t = 0; t < pieces; t += INFL) {
        uint4_t v[INFL];
#pragma unroll
        for (int d = 0; d < INFL; ++d) v[d] = *reinterpret_cast<const uint4_t *>(wbase + (size_t)((t + d) & 7) * span + (size_t)((t + d) >> 3) * 1024);
#pragma unroll
        for (int d = 0; d < INFL; ++d) acc ^= v[d];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[gw] = acc.x;
}

template <int INFL, int NT> __global__ __launch_bounds__(256) void reg2_kernel(const char *in, unsigned *out, size_t span, int pieces)
{
    const int lane = threadIdx.x & 63;
    const int gw = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    const char *wbase = in + (size_t)gw * 8 * span + 16 * lane;
    uint4_t acc = {0u, 0u, 0u, 0u};
    for (int t = 0; t < pieces; t += INFL) {
        uint4_t v[INFL];
#pragma unroll
        for (int d = 0; d < INFL; ++d) {
            const uint4_t *q = reinterpret_cast<const uint4_t *>(wbase + (size_t)((t + d) & 7) * span + (size_t)((t + d) >> 3) * 1024);
            v[d] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int d = 0; d < INFL; ++d) acc ^= v[d];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[gw] = acc.x;
}

// store-only: every wave fills its 8 spans, 1 KiB per instruction (MODE 0 plain, 1 nontemporal), or 32 B per thread
template <int NT> __global__ __launch_bounds__(256) void store_kernel(char *outp, size_t span, int pieces)
{
    const int lane = threadIdx.x & 63;
    const int gw = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    char *wbase = outp + (size_t)gw * 8 * span + 16 * lane;
    const uint4_t v = {(unsigned)gw, (unsigned)lane, 3u, 4u};
    for (int t = 0; t < pieces; ++t) {
        uint4_t *q = reinterpret_cast<uint4_t *>(wbase + (size_t)(t & 7) * span + (size_t)(t >> 3) * 1024);
        if (NT) __builtin_nontemporal_store(v, q); else *q = v;
    }
}

// the decimator's OUTPUT pattern: a wave owns 8 spans of `span` bytes; per instruction it writes CH bytes to each of 8 / (1024 / (8 * CH))...
// precisely: one instruction = 64 lanes x 16 B = 1 KiB = (1024 / CH) spans x CH bytes; the spans are visited round robin
template <int CH> __global__ __launch_bounds__(256) void scatter_store_kernel(char *outp, size_t span, int steps)
{
    const int lane = threadIdx.x & 63;
    const int gw = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    constexpr int LPS = CH / 16;      // lanes per span in one instruction
    constexpr int SPI = 64 / LPS;     // spans per instruction
    char *wbase = outp + (size_t)gw * 8 * span;
    const uint4_t v = {(unsigned)gw, (unsigned)lane, 3u, 4u};
    // the wave writes 8 * span bytes in total: steps instructions
    for (int t = 0; t < steps; ++t) {
        const int visit = t / (8 / (SPI < 8 ? SPI : 8));          // how often the round over the 8 spans has completed
        const int sp = SPI >= 8 ? lane / LPS : ((t % (8 / SPI)) * SPI + lane / LPS);
        char *q = wbase + (size_t)sp * span + (size_t)visit * CH + 16 * (lane % LPS);
        *reinterpret_cast<uint4_t *>(q) = v;
    }
}

template <class F> void timeit(const char *name, size_t bytes, F f)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 60; ++i) f();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("%-58s %.4f ms  %6.0f GB/s   (%s)\n", name, ms, bytes / ms / 1e6, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int waves = 992, wgs = waves / 4;
    const size_t span = 33792 * 4; // bytes, as the decimator's
    const int pieces = (int)(span * 8 / 1024);
    const size_t bytes = (size_t)waves * 8 * span;
    char *d; unsigned *o;
    (void)hipMalloc(&d, bytes + (1 << 20));
    (void)hipMalloc(&o, 1 << 16);
    (void)hipMemset(d, 1, bytes + (1 << 20));
#define DMA(I, N, O) timeit("LDS-DMA 1 KiB pieces, " #I " in flight, nt=" #N " dephase=" #O, bytes, [&] { hipLaunchKernelGGL((dma_kernel<I, N, O>), dim3(wgs), dim3(256), 0, 0, d, o, span, pieces); })
    DMA(7, 0, 0); DMA(15, 0, 0); DMA(23, 0, 0); DMA(31, 0, 0);
    DMA(15, 1, 0); DMA(23, 1, 0); DMA(31, 1, 0);
    DMA(15, 0, 1); DMA(23, 0, 1); DMA(23, 1, 1);
#define REG(I) timeit("register loads 1 KiB pieces, " #I " in flight", bytes, [&] { hipLaunchKernelGGL((reg_kernel<I>), dim3(wgs), dim3(256), 0, 0, d, o, span, pieces); })
    REG(8); REG(16); REG(32);
#define REG2(I, N) timeit("register loads 1 KiB pieces, " #I " in flight, nt=" #N, bytes, [&] { hipLaunchKernelGGL((reg2_kernel<I, N>), dim3(wgs), dim3(256), 0, 0, d, o, span, pieces); })
    REG2(8, 0); REG2(8, 1); REG2(16, 1);
    {
        const size_t ospan = 8448; // bytes of decimated output per span (33792 samples / 16 x 4 B)
        const size_t obytes = (size_t)waves * 8 * ospan;
        const int steps = (int)(8 * ospan / 1024);
#define SCAT(C) timeit("scattered stores, " #C " B per span per visit, 67 MB", obytes, [&] { hipLaunchKernelGGL((scatter_store_kernel<C>), dim3(wgs), dim3(256), 0, 0, d, ospan, steps); })
        SCAT(64); SCAT(128); SCAT(256); SCAT(512); SCAT(1024);
    }
    timeit("stores 1 KiB pieces, 992 waves, plain", bytes, [&] { hipLaunchKernelGGL((store_kernel<0>), dim3(wgs), dim3(256), 0, 0, d, span, pieces); });
    timeit("stores 1 KiB pieces, 992 waves, nt", bytes, [&] { hipLaunchKernelGGL((store_kernel<1>), dim3(wgs), dim3(256), 0, 0, d, span, pieces); });
    timeit("stores 1 KiB pieces, 3968 waves, plain", bytes, [&] { hipLaunchKernelGGL((store_kernel<0>), dim3(wgs * 4), dim3(256), 0, 0, d, span / 4, pieces / 4); });
    timeit("stores 1 KiB pieces, 3968 waves, nt", bytes, [&] { hipLaunchKernelGGL((store_kernel<1>), dim3(wgs * 4), dim3(256), 0, 0, d, span / 4, pieces / 4); });
    return 0;
}
