// testsource_kernels.hip -- a bank of TestSource CW generators on the GPU.
//
// The reference's TestSource::read_samples (TestSource.cpp:395-422) is a float phasor, `amplitude * cos(phasor) * 32768`
// truncated to int16, built with -ffast-math and with a wrap bug (:411-415): not bit-reproducible, so there is nothing
// to be exact against.  This generator keeps the reference's configuration semantics (sdrhip_testsource.cpp) and
// replaces the arithmetic by an integer-exact NCO of our own definition, restated independently in the test oracle:
//   phase(n) = phase0 + n * inc  (mod 2^32),  inc = round(2^32 * carrier offset / sample rate)
//   I = trunc(A * C[phase >> 20] / 2^30),  Q = trunc(A * C[(phase >> 20) - 1024 mod 4096] / 2^30),  clamped to int16
// with C[i] = 2^30 cos(2 pi i / 4096) from a 31-step integer CORDIC (host, sdrhip_testsource.cpp) and A the peak
// amplitude in Q15.  One thread writes four consecutive samples (16 bytes).
#include "sdrhip_internal.h"

namespace sdrhip {
namespace {

typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned nco_sample(const int *T, unsigned phase, int amp)
{
    const unsigned idx = phase >> 20;
    const long long vi = (long long)amp * T[idx], vq = (long long)amp * T[(idx - 1024u) & 4095u];
    int i = (int)(vi >= 0 ? vi >> 30 : -((-vi) >> 30)); // truncation toward zero, like the reference's float -> int16
    int q = (int)(vq >= 0 ? vq >> 30 : -((-vq) >> 30));
    i = i > 32767 ? 32767 : (i < -32768 ? -32768 : i);
    q = q > 32767 ? 32767 : (q < -32768 ? -32768 : q);
    return ((unsigned)i & 0xffffu) | ((unsigned)q << 16);
}

__global__ __launch_bounds__(256) void testsource_kernel(const int *table, const TestSourceParams *par, unsigned *out, size_t out_stride, size_t n)
{
    __shared__ int T[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) T[i] = table[i];
    __syncthreads();
    const TestSourceParams p = par[blockIdx.y];
    unsigned *dst = out + (size_t)blockIdx.y * out_stride;
    const size_t nq = n / 4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < (n + 3) / 4; g += (size_t)gridDim.x * 256) {
        const unsigned ph = p.phase0 + (unsigned)(4 * g) * p.inc; // (mod 2^32: only the low 32 bits of n * inc matter)
        if (g < nq) {
            uint4_t v;
            v.x = nco_sample(T, ph, p.amp);
            v.y = nco_sample(T, ph + p.inc, p.amp);
            v.z = nco_sample(T, ph + 2u * p.inc, p.amp);
            v.w = nco_sample(T, ph + 3u * p.inc, p.amp);
            *reinterpret_cast<uint4_t *>(dst + 4 * g) = v;
        } else {
            for (size_t k = 4 * g; k < n; ++k) dst[k] = nco_sample(T, p.phase0 + (unsigned)k * p.inc, p.amp);
        }
    }
}

} // namespace

hipError_t launch_testsource(const int *table, const TestSourceParams *par, int16_t *out, size_t out_stride, size_t n, int nstreams,
                             hipStream_t stream)
{
    if (n == 0 || nstreams <= 0) return hipSuccess;
    size_t blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(testsource_kernel, dim3((unsigned)blocks, nstreams), dim3(256), 0, stream, table, par,
                       reinterpret_cast<unsigned *>(out), out_stride, n);
    return hipGetLastError();
}

} // namespace sdrhip
