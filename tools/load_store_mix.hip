// Probe (GPU box), the mirror image of store_load_mix.hip: what do a FEW stores cost inside a streaming READ pattern, and does
// clustering them in time help?  992 waves (one per SIMD, 248 CUs), K1m's geometry: a wave reads 8 spans in lockstep, 128 B per
// span and step (one global_load_dwordx4 per lane: 1 KiB per instruction), and writes 1/16 of that: 16 B per lane every 8 steps
// = 64 B per span, twice (lane pairs hold the same data), at the spans' output positions.
//   mode 0: loads only      mode 1: one store per 8 steps (K1m today)
//   mode 2: the stores of 64 steps held back and issued together (8 back to back)      mode 3: of 256 steps (32 back to back)
//   mode 4: of 1024 steps (128 back to back)
//   hipcc --offload-arch=gfx950 -O3 tools/load_store_mix.hip -o /tmp/load_store_mix && /tmp/load_store_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int MODE> __global__ __launch_bounds__(256) void k(const unsigned *in, unsigned *out, size_t span_bytes, unsigned *sink)
{
    constexpr int HOLD = MODE == 2 ? 8 : MODE == 3 ? 32 : MODE == 4 ? 128 : 1;
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int p = lane >> 3, sub = lane & 7; // span of the lane, 16-byte piece of the span's 128 B
    const char *src = reinterpret_cast<const char *>(in) + ((size_t)gw * 8 + p) * span_bytes + 16 * sub;
    char *dst = reinterpret_cast<char *>(out) + (((size_t)gw * 8 + p) * span_bytes >> 4) + 16 * (sub >> 1);
    const size_t steps = span_bytes / 128;
    uint4_t acc = (uint4_t){0u, 0u, 0u, 0u};
    for (size_t s0 = 0; s0 < steps; s0 += 8 * HOLD) {
#pragma unroll 1
        for (int h = 0; h < HOLD; ++h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4_t r = __builtin_nontemporal_load(reinterpret_cast<const uint4_t *>(src + (s0 + 8 * h + i) * 128));
                acc.x ^= r.x; acc.y += r.y; acc.z ^= r.z; acc.w += r.w;
            }
            if (MODE == 1) *reinterpret_cast<uint4_t *>(dst + (s0 / 8 + h) * 64) = acc;
        }
        if (MODE >= 2) {
#pragma unroll 1
            for (int h = 0; h < HOLD; ++h) { acc.x += h; *reinterpret_cast<uint4_t *>(dst + (s0 / 8 + h) * 64) = acc; }
        }
    }
    if (acc.x == 0x12345u && acc.y == 7u) sink[lane] = acc.z;
}

template <int MODE> void run(const unsigned *in, unsigned *out, unsigned *sink, size_t span_bytes, hipEvent_t e0, hipEvent_t e1, const char *what)
{
    auto go = [&]() { k<MODE><<<248, 256>>>(in, out, span_bytes, sink); };
    for (int i = 0; i < 30; ++i) go();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) go();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double bytes = 992.0 * 8 * span_bytes;
    printf("%-70s %.4f ms per %.0f MiB read (%5.0f GB/s)\n", what, ms, bytes / 1048576.0, bytes / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const size_t span_bytes = 135168; // 33 792 samples of 4 B: K1m's spans on 8 x 2^25
    const size_t bytes = 992 * 8 * span_bytes;
    unsigned *in, *out, *sink;
    (void)hipMalloc(&in, bytes + 4096); (void)hipMalloc(&out, bytes / 16 + 4096); (void)hipMalloc(&sink, 4096);
    (void)hipMemset(in, 1, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(in, out, sink, span_bytes, e0, e1, "loads only");
        run<1>(in, out, sink, span_bytes, e0, e1, "+ one 1 KiB store instruction per 8 steps (64 B per span, scattered)");
        run<2>(in, out, sink, span_bytes, e0, e1, "+ the same stores, 8 at a time every 64 steps");
        run<3>(in, out, sink, span_bytes, e0, e1, "+ the same stores, 32 at a time every 256 steps");
        run<4>(in, out, sink, span_bytes, e0, e1, "+ the same stores, 128 at a time every 1024 steps");
    }
    return 0;
}
