// Probe (GPU box): HBM write bandwidth against the NUMBER of concurrent write streams and the burst per visit.
// One-wave workgroups; a wave owns a contiguous region and walks it in bursts of `burst` x 2 KiB (lane t stores 2 x 16 B at byte
// offsets 32 t, 32 t + 16 of each 2 KiB: K5 / K5w's store pattern).  Residency is capped with dynamic LDS (160 KiB per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/store_streams.hip -o /tmp/store_streams && /tmp/store_streams
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int BURST> __global__ __launch_bounds__(64) void k(unsigned *out, size_t bytes_per_wave, int spin)
{
    extern __shared__ int pad[];
    const int lane = threadIdx.x;
    unsigned *base = out + (size_t)blockIdx.x * (bytes_per_wave / 4);
    uint4_t v = (uint4_t){(unsigned)lane, 1u, 2u, 3u};
    for (size_t c = 0; c < bytes_per_wave / 2048; c += BURST) {
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            unsigned *p = base + (c + b) * 512;
            *reinterpret_cast<uint4_t *>(p + lane * 8) = v;
            *reinterpret_cast<uint4_t *>(p + lane * 8 + 4) = v;
        }
        // some arithmetic between bursts (the kernels compute ~700 VALU instructions per 4 bursts of 2 KiB)
        for (int i = 0; i < spin; ++i) v.x = v.x * 1664525u + 1013904223u;
    }
    if (v.x == 0x12345u) pad[lane] = 1;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    unsigned *d;
    hipMalloc(&d, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 150}) {
        for (int burst : {1, 4}) {
            for (int per_cu : {4, 8, 12, 16, 20, 32}) {
                for (size_t seg_kb : {32, 128, 512}) {
                    const size_t bpw = seg_kb << 10;
                    const int nwg = (int)(bytes / bpw);
                    const size_t lds = 160 * 1024 / per_cu / 256 * 256 - 256; // at most per_cu waves per CU
                    auto go = [&]() {
                        if (burst == 1) k<1><<<nwg, 64, lds>>>(d, bpw, spin);
                        else k<4><<<nwg, 64, lds>>>(d, bpw, spin);
                    };
                    for (int i = 0; i < 30; ++i) go();
                    hipEventRecord(e0);
                    for (int i = 0; i < 20; ++i) go();
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
                    printf("spin %3d burst %d x 2 KiB  <= %2d waves/CU  region %4zu KiB (%6d waves)  %.4f ms  %5.0f GB/s\n", spin, burst, per_cu, seg_kb, nwg, ms, bytes / ms / 1e6);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
