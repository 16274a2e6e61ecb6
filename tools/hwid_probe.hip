#include <hip/hip_runtime.h>
__global__ void k(unsigned *out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main()
{
    unsigned *d; hipMalloc(&d, 8 * 4096);
    k<<<2048, 64>>>(d);
    unsigned h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // distinct keys
    int seen[65536] = {0}, nd = 0;
    for (int i = 0; i < 2048; ++i) { unsigned key = ((h[2*i+1] & 0xf) << 8) | ((h[2*i] >> 8) & 0xff); if (!seen[key]++) ++nd; }
    printf("distinct (xcc, se/sh/cu) keys: %d; sample hw %08x xcc %08x, hw %08x xcc %08x\n", nd, h[0], h[1], h[2], h[3]);
    unsigned orv = 0; for (int i = 0; i < 2048; ++i) orv |= h[2*i]; printf("OR of HW_ID: %08x\n", orv);
    unsigned orx = 0; for (int i = 0; i < 2048; ++i) orx |= h[2*i+1]; printf("OR of XCC_ID: %08x\n", orx);
    return 0;
}
