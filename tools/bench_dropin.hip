// tools/bench_dropin.hip -- the drop-in path as a C++ caller sees it: Decimators::decimate16_cen of the adapter
// header on one 65 536-sample IQSampleVector per call (TestSource.h:33, sdrdaemonrx.cpp:590,640), host to host, microseconds per
// call; beside it what the call is made of: an empty launch + synchronise (the floor of ANY synchronous GPU call), the staging
// memcpy, the same call on device pointers.
// build: hipcc -O2 -std=c++14 --offload-arch=gfx950 -Iinclude -Isdrdaemon_amd/adapters tools/bench_dropin.hip -Lsdrdaemon_amd -lsdrhip -Wl,-rpath,$PWD/sdrdaemon_amd -o tools/experiments_r06/bin/bench_dropin   (rpath $ORIGIN/../../../sdrdaemon_amd)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define USE_SSE4_1 1
#include "Decimators.h"

__global__ void empty_kernel(int *p) { if (p) *p = 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 65536;
    const int K = 2000;
    IQSampleVector in(n), out;
    std::mt19937 rng(1234);
    for (size_t i = 0; i < n; ++i) { const unsigned w = rng(); in[i] = IQSample((short)(w & 0xffff), (short)(w >> 16)); }
    Decimators d;
    unsigned ss = 16;
    for (int i = 0; i < 200; ++i) { ss = 16; d.decimate16_cen(ss, in, out); }
    double t0 = now_us();
    for (int i = 0; i < K; ++i) { ss = 16; d.decimate16_cen(ss, in, out); }
    const double t_call = (now_us() - t0) / K;
    // floor: empty launch + synchronise on a stream of our own
    hipStream_t st;
    (void)hipStreamCreate(&st);
    for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, (int *)nullptr); (void)hipStreamSynchronize(st); }
    t0 = now_us();
    for (int i = 0; i < K; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, (int *)nullptr); (void)hipStreamSynchronize(st); }
    const double t_floor = (now_us() - t0) / K;
    // the staging copy: n samples into pinned memory
    void *pin = nullptr;
    (void)hipHostMalloc(&pin, n * 4, hipHostMallocDefault);
    t0 = now_us();
    for (int i = 0; i < K; ++i) { memcpy(pin, in.data(), n * 4); asm volatile("" ::: "memory"); }
    const double t_copy = (now_us() - t0) / K;
    // the same call on device pointers (+ synchronise)
    sdrhip_ctx *ctx = sdrhip_adapter::context();
    sdrhip_decimators *h = nullptr;
    (void)sdrhip_decimators_create(ctx, 1, SDRHIP_HB_EO1, &h);
    int16_t *din = nullptr, *dout = nullptr;
    (void)hipMalloc((void **)&din, n * 4 + 64);
    (void)hipMalloc((void **)&dout, n / 4 + 64);
    (void)hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice);
    size_t no = 0;
    for (int i = 0; i < 200; ++i) { ss = 16; (void)sdrhip_decimate(h, 4, 2, &ss, din, n, n, dout, n >> 4, &no, SDRHIP_MEM_DEVICE); (void)sdrhip_ctx_synchronize(ctx); }
    t0 = now_us();
    for (int i = 0; i < K; ++i) { ss = 16; (void)sdrhip_decimate(h, 4, 2, &ss, din, n, n, dout, n >> 4, &no, SDRHIP_MEM_DEVICE); (void)sdrhip_ctx_synchronize(ctx); }
    const double t_dev = (now_us() - t0) / K;
    printf("%zu samples per call: Decimators::decimate16_cen (adapter, host vectors) %.2f us = %.1f M samples/s | empty launch + synchronise %.2f us | "
           "staging memcpy of the input %.2f us | the call on device pointers + synchronise %.2f us\n", n, t_call, n / t_call, t_floor, t_copy, t_dev);
    return 0;
}
