// Probe (GPU box): HBM write bandwidth of two per-wave store patterns, 2 GiB per launch.
//   A: lane t stores 2 x 16 B at byte offsets 32 t and 32 t + 16 of its wave's 2 KB   (thread-contiguous)
//   B: lane t stores 2 x 16 B at byte offsets 16 t and 1024 + 16 t                      (instruction-contiguous)
//   C: lane t stores 8 x 16 B at byte offsets 128 t + 16 g of its wave's 8 KB            (128 B per thread)
//   D: lane t stores 4 x 16 B at byte offsets 64 t + 16 g of its wave's 4 KB             (64 B per thread)
// build + run: hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int PATTERN> __global__ __launch_bounds__(256) void k(unsigned *out, size_t chunks_per_wg)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned *base = out + ((size_t)blockIdx.x * chunks_per_wg * 4 + wave) * 512; // a wave owns 2 KB chunks, 4 waves interleaved
    uint4_t v = (uint4_t){(unsigned)lane, 1u, 2u, 3u};
    for (size_t c = 0; c < chunks_per_wg; ++c) {
        unsigned *p = base + c * 4 * 512;
        if (PATTERN == 0) {
            *reinterpret_cast<uint4_t *>(p + lane * 8) = v;
            *reinterpret_cast<uint4_t *>(p + lane * 8 + 4) = v;
        } else if (PATTERN == 1) {
            *reinterpret_cast<uint4_t *>(p + lane * 4) = v;
            *reinterpret_cast<uint4_t *>(p + 256 + lane * 4) = v;
        } else if (PATTERN == 2) { // 8 KB per wave and iteration: chunk index scaled by 4
            unsigned *q = out + (((size_t)blockIdx.x * chunks_per_wg + c) * 4 + wave) * 2048 / 4 * 4;
            if (c * 4 < chunks_per_wg)
                for (int g = 0; g < 8; ++g) *reinterpret_cast<uint4_t *>(out + ((size_t)blockIdx.x * chunks_per_wg * 4 * 512) + (c * 4 + wave) * 2048 + lane * 32 + g * 4) = v;
            (void)q;
        } else {
            if (c * 2 < chunks_per_wg)
                for (int g = 0; g < 4; ++g) *reinterpret_cast<uint4_t *>(out + ((size_t)blockIdx.x * chunks_per_wg * 4 * 512) + (c * 4 + wave) * 1024 + lane * 16 + g * 4) = v;
        }
        v.x += 64;
    }
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    unsigned *d;
    hipMalloc(&d, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {2048, 8192, 32768}) {
        const size_t chunks = bytes / 8192 / wgs; // 8 KB per workgroup iteration
        for (int pat = 0; pat < 4; ++pat) {
            auto go = [&]() {
                if (pat == 0) k<0><<<wgs, 256>>>(d, chunks);
                else if (pat == 1) k<1><<<wgs, 256>>>(d, chunks);
                else if (pat == 2) k<2><<<wgs, 256>>>(d, chunks);
                else k<3><<<wgs, 256>>>(d, chunks);
            };
            for (int i = 0; i < 200; ++i) go();
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) go();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("wgs %6d pattern %c  %.3f ms  %.0f GB/s\n", wgs, "ABCD"[pat], ms, bytes / ms / 1e6);
        }
    }
    return 0;
}
