// Probe (GPU box): what do a FEW loads cost inside a streaming store pattern?  One-wave workgroups; a wave streams 2 KiB bursts
// (lane t: 2 x 16 B at 32 t, 32 t + 16) into its own 128 KiB region of a 1 GiB output; every 4th burst it also loads 512 B
// (2 dwords per lane) of a 64 MiB input -- 6 % of the bytes, K5 / K5w's ratio.
//   mode 0: stores only        mode 1: loads in the storing waves, result used 4 bursts later (vmcnt wait)
//   mode 2: loads in the storing waves, result never awaited inside the loop (accumulated at the end)
//   mode 3: the loads of 16 waves are issued by ONE extra loader wave per 16 storing waves (no loads in the storing waves)
//   mode 4: as 1, the loads always hit the same 64 KiB (cache hits)
//   mode 5: one 1 KiB load (dwordx4 per lane) per 8 bursts      mode 6: four 1 KiB loads back to back per 32 bursts
//   mode 7: the wave's whole input (8 KiB = 8 x dwordx4) at its start, awaited before the first store
//   mode 8: as 7, but the loads are NOT awaited before the stores start (consumed at the end)
//   hipcc --offload-arch=gfx950 -O3 tools/store_load_mix.hip -o /tmp/store_load_mix && /tmp/store_load_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(64) void k(unsigned *out, const unsigned *in, size_t bytes_per_wave, unsigned *sink)
{
    const int lane = threadIdx.x;
    const size_t nbursts = bytes_per_wave / 2048;
    if (MODE == 3 && (blockIdx.x % 17) == 16) { // loader wave: the 512-byte pieces of its 16 neighbours, nothing else
        unsigned acc = 0;
        const size_t w0 = (size_t)(blockIdx.x / 17) * 16;
        for (size_t c = 0; c < nbursts; c += 4)
            for (int w = 0; w < 16; ++w) {
                const uint2_t r = __builtin_nontemporal_load(reinterpret_cast<const uint2_t *>(in + ((w0 + w) * nbursts / 4 + c / 4) * 128 + 2 * lane));
                acc += r.x ^ r.y;
            }
        if (acc == 0x12345u) sink[lane] = acc;
        return;
    }
    const size_t wid = MODE == 3 ? (size_t)(blockIdx.x / 17) * 16 + blockIdx.x % 17 : blockIdx.x;
    unsigned *base = out + wid * (bytes_per_wave / 4);
    const unsigned *ib = in + wid * (nbursts / 4) * 128;
    uint4_t v = (uint4_t){(unsigned)lane, 1u, 2u, 3u};
    uint2_t pend = (uint2_t){0u, 0u};
    unsigned acc = 0;
    uint4_t big[8];
    if (MODE == 7 || MODE == 8) {
        for (int i = 0; i < 8; ++i) big[i] = __builtin_nontemporal_load(reinterpret_cast<const uint4_t *>(ib + (size_t)i * 256 + 4 * lane));
        if (MODE == 7) for (int i = 0; i < 8; ++i) v.y += big[i].x ^ big[i].w;
    }
    uint4_t p4 = (uint4_t){0u, 0u, 0u, 0u};
    for (size_t c = 0; c < nbursts; ++c) {
        if (MODE == 5 && (c & 7) == 0) {
            v.y += p4.x ^ p4.w;
            p4 = __builtin_nontemporal_load(reinterpret_cast<const uint4_t *>(ib + (c / 8) * 256 + 4 * lane));
        }
        if (MODE == 6 && (c & 31) == 0) {
            v.y += p4.x ^ p4.w;
            for (int i = 0; i < 4; ++i) { const uint4_t r = __builtin_nontemporal_load(reinterpret_cast<const uint4_t *>(ib + (c / 8 + i) * 256 + 4 * lane)); p4.x ^= r.x; p4.w += r.w; }
        }
        if ((MODE == 1 || MODE == 2 || MODE == 4) && (c & 3) == 0) {
            if (MODE == 1 || MODE == 4) { v.y += pend.x; v.z ^= pend.y; } // consumes the load issued 4 bursts ago
            else acc += pend.x ^ pend.y;
            const size_t off = MODE == 4 ? ((c / 4) & 127) * 128 : (c / 4) * 128;
            pend = __builtin_nontemporal_load(reinterpret_cast<const uint2_t *>(ib + off + 2 * lane));
            if (MODE == 2) asm volatile("" : "+v"(pend.x), "+v"(pend.y)); // (keeps the load, no use before the next one)
        }
        unsigned *p = base + c * 512;
        *reinterpret_cast<uint4_t *>(p + lane * 8) = v;
        *reinterpret_cast<uint4_t *>(p + lane * 8 + 4) = v;
        v.x += 64;
    }
    if (MODE == 8) for (int i = 0; i < 8; ++i) acc += big[i].x ^ big[i].w;
    if (acc == 0x12345u || pend.x == 0x77u || p4.x == 0x99u) sink[lane] = acc;
}

template <int MODE> void run(unsigned *d, const unsigned *in, unsigned *sink, size_t bytes, hipEvent_t e0, hipEvent_t e1, const char *what)
{
    const size_t bpw = 128 << 10;
    int nwg = (int)(bytes / bpw);
    if (MODE == 3) nwg = nwg / 16 * 17;
    auto go = [&]() { k<MODE><<<nwg, 64>>>(d, in, bpw, sink); };
    for (int i = 0; i < 30; ++i) go();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) go();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-86s %.4f ms per GiB written (%5.0f GB/s)\n", what, ms, bytes / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    unsigned *d, *in, *sink;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&in, bytes / 16 + 4096); (void)hipMalloc(&sink, 4096);
    (void)hipMemset(in, 1, bytes / 16 + 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(d, in, sink, bytes, e0, e1, "stores only");
        run<1>(d, in, sink, bytes, e0, e1, "+ 512 B load per 4 bursts in the storing wave, awaited 4 bursts later");
        run<2>(d, in, sink, bytes, e0, e1, "+ the same loads, never awaited inside the loop");
        run<4>(d, in, sink, bytes, e0, e1, "+ the same loads, awaited, always cache hits (64 KiB window)");
        run<3>(d, in, sink, bytes, e0, e1, "+ the same loads issued by one loader wave per 16 storing waves");
        run<5>(d, in, sink, bytes, e0, e1, "+ one 1 KiB load (dwordx4) per 8 bursts");
        run<6>(d, in, sink, bytes, e0, e1, "+ four 1 KiB loads back to back per 32 bursts");
        run<7>(d, in, sink, bytes, e0, e1, "+ the wave's whole input (8 x 1 KiB) at its start, awaited before the first store");
        run<8>(d, in, sink, bytes, e0, e1, "+ the wave's whole input at its start, not awaited");
    }
    return 0;
}
