// Probe (GPU box): does LDS traffic of the same waves slow a streaming store pattern down (and vice versa)?
// One-wave workgroups, 20 per CU; per 2 KiB burst of global stores (lane t: 2 x 16 B at 32 t, 32 t + 16) a wave issues NW
// ds_write_b128 and NR ds_read_b128 on its private 7.5 KiB of LDS (K5w per last-stage invocation: ~7 writes, ~13 reads) and NV
// independent VALU multiply-adds.  STORE = 0 drops the global stores (LDS / VALU time alone).
//   hipcc --offload-arch=gfx950 -O3 tools/store_lds_mix.hip -o /tmp/store_lds_mix && /tmp/store_lds_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int NW, int NR, int NV, int STORE> __global__ __launch_bounds__(64) void k(unsigned *out, size_t bytes_per_wave)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[1920];
    const int lane = threadIdx.x;
    unsigned *base = out + (size_t)blockIdx.x * (bytes_per_wave / 4);
    uint4_t v = (uint4_t){(unsigned)lane, 1u, 2u, 3u};
    uint4_t *l4 = reinterpret_cast<uint4_t *>(lds);
    unsigned a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3;
    for (size_t c = 0; c < bytes_per_wave / 2048; ++c) {
#pragma unroll
        for (int i = 0; i < NW; ++i) l4[(lane + 64 * i) % 480] = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NR; ++i) { const uint4_t r = l4[(lane * 3 + 61 * i + 7) % 480]; v.x ^= r.x; v.y += r.y; v.z ^= r.z; v.w += r.w; }
#pragma unroll
        for (int i = 0; i < NV / 4; ++i) {
            asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a0) : "v"(v.x), "v"(a1));
            asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a1) : "v"(v.y), "v"(a2));
            asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a2) : "v"(v.z), "v"(a3));
            asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a3) : "v"(v.w), "v"(a0));
        }
        v.x += a0; v.y += a1;
        unsigned *p = base + c * 512;
        if (STORE) {
            *reinterpret_cast<uint4_t *>(p + lane * 8) = v;
            *reinterpret_cast<uint4_t *>(p + lane * 8 + 4) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (v.x == 0x12345u && a2 == 77u && a3 == 5u) out[lane] = v.y;
}

template <int NW, int NR, int NV, int STORE> void run(unsigned *d, size_t bytes, hipEvent_t e0, hipEvent_t e1)
{
    const size_t bpw = 128 << 10;
    const int nwg = (int)(bytes / bpw);
    auto go = [&]() { k<NW, NR, NV, STORE><<<nwg, 64>>>(d, bpw); };
    for (int i = 0; i < 30; ++i) go();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) go();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("per 2 KiB burst: %2d ds_write_b128 %2d ds_read_b128 %3d v_mad  stores %d:  %.4f ms per GiB  (%5.0f GB/s)\n", NW, NR, NV, STORE, ms, bytes / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    unsigned *d;
    (void)hipMalloc(&d, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run<0, 0, 0, 1>(d, bytes, e0, e1);
    run<7, 13, 0, 0>(d, bytes, e0, e1);
    run<7, 13, 0, 1>(d, bytes, e0, e1);
    run<7, 0, 0, 0>(d, bytes, e0, e1);
    run<7, 0, 0, 1>(d, bytes, e0, e1);
    run<0, 13, 0, 0>(d, bytes, e0, e1);
    run<0, 13, 0, 1>(d, bytes, e0, e1);
    run<0, 0, 172, 0>(d, bytes, e0, e1);
    run<0, 0, 172, 1>(d, bytes, e0, e1);
    run<7, 13, 172, 0>(d, bytes, e0, e1);
    run<7, 13, 172, 1>(d, bytes, e0, e1);
    run<4, 8, 120, 0>(d, bytes, e0, e1);
    run<4, 8, 120, 1>(d, bytes, e0, e1);
    return 0;
}
