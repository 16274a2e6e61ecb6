// Probe (GPU box): HBM read bandwidth of coalesced 16-byte loads, 2 GiB per launch, with 1, 2, 4 or 8
// loads in flight per thread (the decimator keeps 2 (centred) or 8 (inf / sup) per thread in flight).
// build + run: hipcc --offload-arch=gfx950 -O3 tools/read_pattern.hip -o /tmp/read_pattern && /tmp/read_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int INFLIGHT> __global__ __launch_bounds__(256) void k(const uint4_t *in, unsigned *out, size_t iters)
{
    // a workgroup streams a contiguous region: iteration i covers INFLIGHT x 4 KB
    const uint4_t *p = in + (size_t)blockIdx.x * iters * INFLIGHT * 256 + threadIdx.x;
    uint4_t acc = (uint4_t){0u, 0u, 0u, 0u};
    for (size_t i = 0; i < iters; ++i) {
        uint4_t v[INFLIGHT];
#pragma unroll
        for (int n = 0; n < INFLIGHT; ++n) v[n] = p[(i * INFLIGHT + n) * 256];
#pragma unroll
        for (int n = 0; n < INFLIGHT; ++n) acc ^= v[n];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = acc.x;
}

template <int INFLIGHT> void run(const uint4_t *d, unsigned *o, int wgs)
{
    const size_t bytes = (size_t)2 << 30;
    const size_t iters = bytes / 16 / 256 / INFLIGHT / wgs;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 100; ++i) k<INFLIGHT><<<wgs, 256>>>(d, o, iters);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k<INFLIGHT><<<wgs, 256>>>(d, o, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("wgs %6d  %d loads in flight / thread  %.3f ms  %.0f GB/s\n", wgs, INFLIGHT, ms, bytes / ms / 1e6);
}

int main()
{
    uint4_t *d; unsigned *o;
    (void)hipMalloc(&d, (size_t)2 << 30);
    (void)hipMalloc(&o, 1 << 20);
    (void)hipMemset(d, 1, (size_t)2 << 30);
    for (int wgs : {1024, 2048, 4096, 16384}) {
        run<1>(d, o, wgs); run<2>(d, o, wgs); run<4>(d, o, wgs); run<8>(d, o, wgs);
    }
    return 0;
}
