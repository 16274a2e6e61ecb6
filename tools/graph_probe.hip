// Probe (GPU box): the drop-in call's device work -- upload of one TestSource block (65 536 samples = 256 KiB) from pinned memory, one
// short kernel, download of the result (16 KiB), synchronise -- as three stream operations against ONE hipGraphLaunch of the same
// three nodes.  Question: would a captured graph take anything off the 11-us "launch + synchronise" floor of a host-pointer call?
//   hipcc --offload-arch=gfx950 -O3 tools/graph_probe.hip -o tools/experiments_r06/bin/graph_probe && tools/experiments_r06/bin/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(const int *in, int *out, int n_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += in[16 * i + k];
        out[i] = s;
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t nin = 65536, nout = nin / 16;
    int *hin, *hout, *din, *dout;
    CK(hipHostMalloc(&hin, nin * 4)); CK(hipHostMalloc(&hout, nout * 4));
    CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&dout, nout * 4));
    memset(hin, 1, nin * 4);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto ops = [&]() {
        (void)hipMemcpyAsync(din, hin, nin * 4, hipMemcpyHostToDevice, st);
        hipLaunchKernelGGL(work, dim3((nout + 255) / 256), dim3(256), 0, st, din, dout, (int)nout);
        (void)hipMemcpyAsync(hout, dout, nout * 4, hipMemcpyDeviceToHost, st);
    };
    // the same three operations captured once
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    ops();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const int K = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 200; ++i) { ops(); CK(hipStreamSynchronize(st)); }
        double t0 = now_us();
        for (int i = 0; i < K; ++i) { ops(); CK(hipStreamSynchronize(st)); }
        const double a = (now_us() - t0) / K;
        for (int i = 0; i < 200; ++i) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
        t0 = now_us();
        for (int i = 0; i < K; ++i) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
        const double b = (now_us() - t0) / K;
        t0 = now_us();
        for (int i = 0; i < K; ++i) { hipLaunchKernelGGL(work, dim3((nout + 255) / 256), dim3(256), 0, st, din, dout, (int)nout); CK(hipStreamSynchronize(st)); }
        const double c = (now_us() - t0) / K;
        t0 = now_us();
        for (int i = 0; i < K; ++i) { (void)hipMemcpyAsync(din, hin, nin * 4, hipMemcpyHostToDevice, st); CK(hipStreamSynchronize(st)); }
        const double d = (now_us() - t0) / K;
        printf("round %d  upload + kernel + download + synchronise: stream operations %.2f us | one hipGraphLaunch %.2f us | kernel alone + synchronise %.2f us | upload alone + synchronise %.2f us\n", rep, a, b, c, d);
    }
    return hout[0] == 12345;
}
