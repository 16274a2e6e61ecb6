// Probe (GPU box): what the decimator's matrix-core path relies on.
//  1. v_mfma_i32_16x16x64_i8 fragment layout: A lane l = row l&15, B lane l = column l&15, both hold K slots
//     16*(l>>4) .. +15 (byte t of the 4 dwords); D lane l holds rows 4*(l>>4) + reg of column l&15.
//  2. int32 accumulation wraps (no saturation).
//  3. issue rate of the instruction at 1 / 2 / 4 waves per SIMD, alone and with VALU work (v_perm_b32 /
//     v_lshl_add_u32) interleaved in the same wave.
//  4. bandwidth of the span-strided read pattern of the kernel (8 spans x 128 B per wave and step, I / Q lane pairs
//     reading the same 16 bytes) against the coalesced pattern.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const int8_t *A /*16x64*/, const int8_t *B /*64x16*/, const int *C /*16x16*/, int *D)
{
    const int l = threadIdx.x, rc = l & 15, kq = l >> 4;
    int4_t a, b, c;
    int8_t ab[16], bb[16];
    for (int t = 0; t < 16; ++t) {
        ab[t] = A[rc * 64 + 16 * kq + t];
        bb[t] = B[(16 * kq + t) * 16 + rc];
    }
    __builtin_memcpy(&a, ab, 16);
    __builtin_memcpy(&b, bb, 16);
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * kq + r) * 16 + rc];
    int4_t d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + rc] = d[r];
}

// MFMA issue rate: NACC independent accumulators, VALU_PER extra VALU instructions per MFMA
template <int NACC, int VALU_PER, int KIND> __global__ __launch_bounds__(256) void rate_kernel(int *out, int iters, unsigned seed, unsigned long long *cyc)
{
    const unsigned long long t0 = __builtin_readcyclecounter(); // s_memtime: shader clock
    int4_t a = {(int)seed, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
    int4_t acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = (int4_t){n, 0, 0, 0};
    unsigned v0 = threadIdx.x, v1 = seed, v2 = 0x05010400u, v3 = 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            acc[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[n], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU_PER; ++v) {
                if (KIND == 0) { // v_perm_b32 chain x2 (independent chains)
                    if (v & 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                    else asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2));
                } else { // v_lshl_add_u32
                    if (v & 1) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(v0) : "v"(v1));
                    else asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(v3) : "v"(v1));
                }
            }
        }
    }
    int s = (int)(v0 ^ v3);
    for (int n = 0; n < NACC; ++n) s ^= acc[n][0] ^ acc[n][1] ^ acc[n][2] ^ acc[n][3];
    if (s == 0x1234567) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;
}

template <int NACC, int VALU_PER, int KIND> void rate(int *out, int blocks_per_cu, const char *name)
{
    const int iters = 65536 / NACC;
    static unsigned long long *cyc = nullptr;
    if (!cyc) (void)hipMalloc(&cyc, 8);
    const int blocks = 256 * blocks_per_cu; // 256 threads = one wave per SIMD per block
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) rate_kernel<NACC, VALU_PER, KIND><<<blocks, 256>>>(out, iters, 1, cyc);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) rate_kernel<NACC, VALU_PER, KIND><<<blocks, 256>>>(out, iters, 1, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    unsigned long long hc = 0;
    (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * NACC * blocks_per_cu;
    const double ns_per = ms * 1e6 / mfma_per_simd;
    printf("%-10s nacc %d valu/mfma %d waves/SIMD %d: %.3f ms, %.2f ns per MFMA per SIMD, %.0f TOPS; wave 0: %.1f shader cycles per own MFMA (clock ~%.2f GHz)\n", name,
           NACC, VALU_PER, blocks_per_cu, ms, ns_per, 2.0 * 16384 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12, (double)hc / ((double)iters * NACC),
           (double)hc / (ms * 1e6));
}

// read patterns over 1 GiB: PAT 0 = coalesced 16 B per lane (1 KB per wave instruction), PAT 1 = the kernel's:
// a wave owns 8 spans; per step every span contributes 128 B: lane (p = span, c = I/Q duplicate, kq) reads 16 B at
// span + 128 step + 64 j + 16 kq for j = 0, 1 (lanes 2p and 2p+1 read the same bytes)
template <int PAT, int DEPTH> __global__ __launch_bounds__(256) void read_kernel(const uint4_t *in, unsigned *out, size_t span_bytes, int steps)
{
    const int lane = threadIdx.x & 63, wave = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    uint4_t acc = {0u, 0u, 0u, 0u};
    if (PAT == 0) {
        const uint4_t *p = in + (size_t)wave * (span_bytes * 8 / 16) + lane;
        for (int s = 0; s < steps; s += DEPTH) {
            uint4_t v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = p[(size_t)(s + d) * 64];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    } else {
        const int n = lane & 15, kq = lane >> 4, span = n >> 1;
        const char *base = reinterpret_cast<const char *>(in) + ((size_t)wave * 8 + span) * span_bytes + 16 * kq;
        for (int s = 0; s < steps; s += DEPTH) {
            uint4_t v[2 * DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                v[2 * d] = *reinterpret_cast<const uint4_t *>(base + (size_t)(s + d) * 128);
                v[2 * d + 1] = *reinterpret_cast<const uint4_t *>(base + (size_t)(s + d) * 128 + 64);
            }
#pragma unroll
            for (int d = 0; d < 2 * DEPTH; ++d) acc ^= v[d];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x] = acc.x;
}

template <int PAT, int DEPTH> void readbw(const uint4_t *d, unsigned *o, int waves_per_simd)
{
    const size_t bytes = (size_t)1 << 30;
    const int waves = 1024 * waves_per_simd, blocks = waves / 4;
    const size_t span_bytes = bytes / waves / 8;
    const int steps = (int)(PAT == 0 ? span_bytes * 8 / 1024 : span_bytes / 128);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) read_kernel<PAT, DEPTH><<<blocks, 256>>>(d, o, span_bytes, steps);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) read_kernel<PAT, DEPTH><<<blocks, 256>>>(d, o, span_bytes, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("read pattern %s, %d steps in flight, %d waves/SIMD (span %zu KB): %.3f ms, %.0f GB/s\n", PAT ? "span-strided" : "coalesced   ", DEPTH,
           waves_per_simd, span_bytes >> 10, ms, bytes / ms / 1e6);
}

int main()
{
    // ---- 1, 2: layout and wrap
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    std::vector<int> C(256), D(256), R(256);
    srand(7);
    for (auto &x : A) x = (int8_t)(rand() & 0xff);
    for (auto &x : B) x = (int8_t)(rand() & 0xff);
    for (int i = 0; i < 256; ++i) C[i] = (i & 1) ? 0x7fffff00 : -0x7fffff00; // forces wrap-around
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            unsigned s = (unsigned)C[i * 16 + j];
            for (int k = 0; k < 64; ++k) s += (unsigned)((int)A[i * 64 + k] * (int)B[k * 16 + j]);
            R[i * 16 + j] = (int)s;
        }
    int8_t *dA, *dB; int *dC, *dD;
    (void)hipMalloc(&dA, A.size()); (void)hipMalloc(&dB, B.size()); (void)hipMalloc(&dC, 1024); (void)hipMalloc(&dD, 1024);
    (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(dA, dB, dC, dD);
    (void)hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("mfma_i32_16x16x64_i8 layout + wrap-around accumulate: %s (%d mismatches of 256)\n", bad ? "MISMATCH" : "as assumed", bad);

    // ---- 3: issue rate
    int *out;
    (void)hipMalloc(&out, 64);
    for (int w : {1, 2, 4}) {
        rate<1, 0, 0>(out, w, "mfma");
        rate<4, 0, 0>(out, w, "mfma");
        rate<4, 2, 0>(out, w, "+perm");
        rate<4, 4, 0>(out, w, "+perm");
        rate<4, 8, 0>(out, w, "+perm");
        rate<4, 4, 1>(out, w, "+lshl_add");
        rate<4, 8, 1>(out, w, "+lshl_add");
    }

    // ---- 4: read patterns
    uint4_t *d; unsigned *o;
    (void)hipMalloc(&d, ((size_t)1 << 30) + 4096);
    (void)hipMalloc(&o, 1 << 20);
    (void)hipMemset(d, 1, (size_t)1 << 30);
    for (int w : {1, 2, 4}) {
        readbw<0, 4>(d, o, w);
        readbw<1, 2>(d, o, w);
        readbw<1, 4>(d, o, w);
        readbw<1, 8>(d, o, w);
    }
    return bad != 0;
}
