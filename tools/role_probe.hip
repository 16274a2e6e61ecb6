// Probe (GPU box): where does the dispatcher put the first workgroups of a big grid, and does the first-on-its-CU role claim of
// rx_fused_kernel (decim_mfma.hip) give one long-running unit per CU?
// build + run: hipcc --offload-arch=gfx950 -O3 tools/role_probe.hip -o /tmp/role_probe && /tmp/role_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
template <int WPE> __global__ __launch_bounds__(256, WPE) void k(unsigned *tab, unsigned tag, int nmf, int nenc, unsigned *log, int spin_long, int spin_short)
{
    __shared__ int lds[8705];
    __shared__ int s_unit;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned key = ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);
    if (threadIdx.x == 0) {
        unsigned *cnt = tab + 4096 + 4 * (tag & 1u);
        if (blockIdx.x == 0) { unsigned *nxt = tab + 4096 + 4 * ((tag + 1u) & 1u); nxt[0] = 0u; nxt[1] = 0u; nxt[2] = 0u; }
        const bool first = atomicMax(tab + key, tag) < tag;
        int unit = -1;
        if (first) { const unsigned u = atomicAdd(cnt + 0, 1u); if (u < (unsigned)nmf) unit = (int)u; }
        if (unit < 0) { const unsigned u = atomicAdd(cnt + 2, 1u); if (u < (unsigned)nenc) unit = nmf + (int)u; }
        if (unit < 0) { const unsigned u = atomicAdd(cnt + 0, 1u); if (u < (unsigned)nmf) unit = (int)u; }
        s_unit = unit;
        log[3 * blockIdx.x] = key; log[3 * blockIdx.x + 1] = (unsigned)unit; log[3 * blockIdx.x + 2] = first;
    }
    __syncthreads();
    const int unit = s_unit;
    lds[threadIdx.x] = unit;
    const long long t0 = clock64();
    const long long dur = unit < nmf ? spin_long : spin_short;
    while (clock64() - t0 < dur) { asm volatile("s_sleep 1"); }
    if (lds[(threadIdx.x + 1) & 255] == 0x7fffffff) log[0] = 1;
}
template <int WPE> void run(const char *name, int nmf, int nenc, int static_roles)
{
    unsigned *tab, *log;
    const int grid = nmf + nenc;
    hipMalloc(&tab, (4096 + 8) * 4); hipMemset(tab, 0, (4096 + 8) * 4);
    hipMalloc(&log, grid * 12);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (unsigned tag = 1; tag <= 5; ++tag) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<WPE>), dim3(grid), dim3(256), 0, 0, tab, tag, nmf, nenc, log, 240000, 6000); // 100 MHz clock64: 2.4 ms / 60 us?
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<unsigned> h(grid * 3);
    hipMemcpy(h.data(), log, grid * 12, hipMemcpyDeviceToHost);
    std::map<unsigned, int> percu; int firsts = 0, mfirst = 0;
    for (int i = 0; i < grid; ++i) { firsts += h[3 * i + 2]; if ((int)h[3 * i + 1] < nmf && (int)h[3 * i + 1] >= 0) { percu[h[3 * i]]++; mfirst += h[3 * i + 2]; } }
    int mx = 0; for (auto &p : percu) if (p.second > mx) mx = p.second;
    // where did blocks 0..nmf-1 land (static roles)?
    std::map<unsigned, int> st; for (int i = 0; i < nmf && i < grid; ++i) st[h[3 * i]]++;
    int smx = 0; for (auto &p : st) if (p.second > smx) smx = p.second;
    printf("%-34s grid %5d: %.3f ms; firsts %d; long units on %zu CUs (max %d per CU, %d claimed by firsts); blocks 0..%d lie on %zu CUs (max %d per CU)\n", name, grid, best, firsts,
           percu.size(), mx, mfirst, nmf - 1, st.size(), smx);
    hipFree(tab); hipFree(log);
}
int main()
{
    run<2>("bounds 2, no short units", 248, 24, 0);
    run<2>("bounds 2, 2080 short units", 248, 2080, 0);
    run<3>("bounds 3, no short units", 248, 24, 0);
    run<3>("bounds 3, 2080 short units", 248, 2080, 0);
    run<4>("bounds 4, 2080 short units", 248, 2080, 0);
    return 0;
}
