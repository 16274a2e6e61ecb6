// Probe (GPU box): how many 256-thread workgroups of a given footprint does a CU of the MI355X hold at once?  (The FFT encoder -- 96 VGPRs, 23.5 KB
// of LDS -- should fit five per CU by the arithmetic, the stamps show four: profiles/r06_enc_occupancy.txt.)  Every workgroup notes s_memrealtime at its
// start and spins 30 us; workgroups that start in the first 5 us are the resident round.
//   hipcc --offload-arch=gfx950 -O3 tools/residency_probe.hip -o tools/experiments_r06/bin/residency_probe && tools/experiments_r06/bin/residency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int VG> __global__ __launch_bounds__(256) void k(unsigned long long *t)
{
    extern __shared__ unsigned char lds[];
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (threadIdx.x == 0) t[blockIdx.x] = t0;
    // force the register footprint: the highest register named is allocated
    if (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (VG == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
    if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    unsigned long long t1;
    do {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    } while (t1 - t0 < 3000); // 30 us at 100 MHz
}

template <int VG> void run(int lds_bytes, unsigned long long *dt)
{
    const int grid = 256 * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<VG>, dim3(grid), dim3(256), lds_bytes, 0, dt);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), dt, grid * 8, hipMemcpyDeviceToHost);
    const unsigned long long m = *std::min_element(h.begin(), h.end());
    int first = 0;
    for (auto v : h) first += (v - m) < 500;
    printf("%3d VGPRs, %6d B of LDS per workgroup of 256 threads: %4d of %d workgroups start in the first 5 us = %.2f per CU\n", VG, lds_bytes, first, grid, first / 256.0);
}

int main()
{
    unsigned long long *dt;
    (void)hipMalloc(&dt, 8 * 4096);
    for (int lds : {64, 23556, 32768, 36356, 65536}) {
        run<64>(lds, dt);
        run<96>(lds, dt);
        run<104>(lds, dt);
        run<128>(lds, dt);
    }
    return 0;
}
