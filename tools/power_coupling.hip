// Probe (GPU box): does HBM read bandwidth drop while the matrix and vector pipes are busy?  Half of the workgroups
// stream 1 GiB with coalesced 16-byte loads (4 loads in flight per lane), the other half spin on MFMA + VALU work
// without touching memory.  Reported: time of the streaming kernel alone, of the compute kernel alone, of both in
// one launch (one block of each kind per CU slot), and the shader clock seen by a compute wave (s_memtime / time).
// build + run: hipcc --offload-arch=gfx950 -O3 tools/power_coupling.hip -o /tmp/pc && /tmp/pc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int VALU_PER> __device__ void compute(int iters, unsigned long long *cyc, int *out)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    int4_t a = {1, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
    int4_t acc[4];
    for (int n = 0; n < 4; ++n) acc[n] = (int4_t){n, 0, 0, 0};
    unsigned v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * (j + 1);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            acc[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[n], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < VALU_PER; ++j) v[j & 7] = __builtin_amdgcn_perm(v[j & 7], v[(j + 1) & 7], 0x05010400u) + (unsigned)i;
        }
    }
    int s = 0;
    for (int j = 0; j < 8; ++j) s ^= (int)v[j];
    for (int n = 0; n < 4; ++n) s ^= acc[n][0] ^ acc[n][1] ^ acc[n][2] ^ acc[n][3];
    if (s == 0x1234567) out[0] = s;
    if (blockIdx.x == 1 && threadIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;
}

__device__ void stream(const uint4_t *in, unsigned *out, size_t iters, int bid, int nb)
{
    const uint4_t *p = in + (size_t)bid * iters * 4 * 256 + threadIdx.x;
    uint4_t acc = {0u, 0u, 0u, 0u};
    for (size_t i = 0; i < iters; ++i) {
        uint4_t v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = p[(i * 4 + n) * 256];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc ^= v[n];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[bid] = acc.x;
}

// mode 0: all blocks stream; 1: all compute; 2: even blocks stream, odd blocks compute
template <int VALU_PER> __global__ __launch_bounds__(256) void k(int mode, const uint4_t *in, unsigned *out, size_t siters, int citers, unsigned long long *cyc)
{
    const int b = blockIdx.x;
    if (mode == 0) stream(in, out, siters, b, gridDim.x);
    else if (mode == 1) compute<VALU_PER>(citers, cyc, (int *)out);
    else if (b & 1) compute<VALU_PER>(citers, cyc, (int *)out);
    else stream(in, out, siters, b >> 1, gridDim.x >> 1);
}

template <int VALU_PER> float run(int mode, int blocks, const uint4_t *d, unsigned *o, size_t siters, int citers, unsigned long long *cyc, double *ghz)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 30; ++i) k<VALU_PER><<<blocks, 256>>>(mode, d, o, siters, citers, cyc);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k<VALU_PER><<<blocks, 256>>>(mode, d, o, siters, citers, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc = 0;
    (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    *ghz = mode ? (double)hc / (ms / 20 * 1e6) : 0.0;
    return ms / 20;
}

template <int VALU_PER> void experiment(const uint4_t *d, unsigned *o, unsigned long long *cyc)
{
    const size_t bytes = (size_t)1 << 30;
    const int sb = 1024;                       // streaming blocks: 4 per CU
    const size_t siters = bytes / 16 / 256 / 4 / sb;
    double ghz;
    const float ts = run<VALU_PER>(0, sb, d, o, siters, 0, cyc, &ghz);
    // size the compute part to last about as long as the stream alone
    int citers = 2000;
    float tc = run<VALU_PER>(1, sb, d, o, siters, citers, cyc, &ghz);
    citers = (int)(citers * ts / tc);
    tc = run<VALU_PER>(1, sb, d, o, siters, citers, cyc, &ghz);
    const double ghz_c = ghz;
    const float tb = run<VALU_PER>(2, 2 * sb, d, o, siters, citers, cyc, &ghz);
    printf("VALU/MFMA %d: stream alone %.3f ms (%.0f GB/s) | compute alone %.3f ms (clock %.2f GHz) | both %.3f ms (stream >= %.0f GB/s, clock %.2f GHz)\n",
           VALU_PER, ts, bytes / ts / 1e6, tc, ghz_c, tb, bytes / tb / 1e6, ghz);
}

int main()
{
    uint4_t *d; unsigned *o; unsigned long long *cyc;
    (void)hipMalloc(&d, ((size_t)1 << 30) + 4096);
    (void)hipMalloc(&o, 1 << 20);
    (void)hipMalloc(&cyc, 8);
    (void)hipMemset(d, 1, (size_t)1 << 30);
    experiment<0>(d, o, cyc);
    experiment<2>(d, o, cyc);
    experiment<4>(d, o, cyc);
    experiment<8>(d, o, cyc);
    return 0;
}
