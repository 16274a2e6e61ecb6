// Probe (GPU box): peak issue rate of the integer VALU instructions the kernels are made of.
// Each thread runs ITER x 32 instructions on 8 independent accumulators; 256 CUs x 8 waves/SIMD resident.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const unsigned b = blockIdx.x * 2654435761u + 12345u;
    unsigned c = seed | 1u;
    asm volatile("s_mov_b32 %0, %0" : "+s"(c));
    for (int i = 0; i < iters; ++i) {
#define MAD(n) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a##n) : "v"(b), "s"(c));
#define DOT(n) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a##n) : "v"(b), "s"(c));
#define ADD(n) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define XOR(n) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define PRM(n) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a##n) : "v"(b), "s"(c));
#define MUL(n) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a##n) : "v"(b));
        if (OP == 0) { REP8(MAD) REP8(MAD) REP8(MAD) REP8(MAD) }
        if (OP == 1) { REP8(DOT) REP8(DOT) REP8(DOT) REP8(DOT) }
        if (OP == 2) { REP8(ADD) REP8(ADD) REP8(ADD) REP8(ADD) }
        if (OP == 3) { REP8(XOR) REP8(XOR) REP8(XOR) REP8(XOR) }
        if (OP == 4) { REP8(PRM) REP8(PRM) REP8(PRM) REP8(PRM) }
        if (OP == 5) { REP8(MUL) REP8(MUL) REP8(MUL) REP8(MUL) }
#define MADADD(n) MAD(n) ADD(n)
#define PRMXOR(n) PRM(n) XOR(n)
#define DOTADD(n) DOT(n) ADD(n)
#define MAD3ADD(n) MAD(n) MAD(n) MAD(n) ADD(n)
        if (OP == 6) { REP8(MADADD) REP8(MADADD) }                 // 16 mad + 16 add, alternating (stage >= 1 of K1)
        if (OP == 7) { REP8(PRMXOR) REP8(PRMXOR) }                 // 16 perm + 16 xor (GF kernels)
        if (OP == 8) { REP8(DOTADD) REP8(DOTADD) }
        if (OP == 9) { REP8(MAD) REP8(MAD) REP8(ADD) REP8(ADD) }   // same mix in blocks of 16
        if (OP == 10) { REP8(MAD3ADD) }                            // 24 mad + 8 add
#define M24(n) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define AD3(n) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a##n) : "v"(b), "s"(c));
#define LSA(n) asm volatile("v_lshl_add_u32 %0, %1, 3, %0" : "+v"(a##n) : "v"(b));
#define BT3(n) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96" : "+v"(a##n) : "v"(b), "s"(c));
#define AND(n) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define SHR(n) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a##n));
#define ASR(n) asm volatile("v_ashrrev_i32 %0, 13, %0" : "+v"(a##n));
#define BFE(n) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a##n));
#define PKA(n) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a##n) : "v"(b));
#define DT2(n) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a##n) : "v"(b), "s"(c));
#define DT4(n) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a##n) : "v"(b), "s"(c));
#define MOV(n) asm volatile("v_mov_b32 %0, %1" : "+v"(a##n) : "v"(b));
        if (OP == 11) { REP8(M24) REP8(M24) REP8(M24) REP8(M24) }
        if (OP == 12) { REP8(AD3) REP8(AD3) REP8(AD3) REP8(AD3) }
        if (OP == 13) { REP8(LSA) REP8(LSA) REP8(LSA) REP8(LSA) }
        if (OP == 14) { REP8(BT3) REP8(BT3) REP8(BT3) REP8(BT3) }
        if (OP == 15) { REP8(AND) REP8(AND) REP8(AND) REP8(AND) }
        if (OP == 16) { REP8(SHR) REP8(SHR) REP8(SHR) REP8(SHR) }
        if (OP == 17) { REP8(ASR) REP8(ASR) REP8(ASR) REP8(ASR) }
        if (OP == 18) { REP8(BFE) REP8(BFE) REP8(BFE) REP8(BFE) }
        if (OP == 19) { REP8(PKA) REP8(PKA) REP8(PKA) REP8(PKA) }
        if (OP == 20) { REP8(DT2) REP8(DT2) REP8(DT2) REP8(DT2) }
        if (OP == 21) { REP8(DT4) REP8(DT4) REP8(DT4) REP8(DT4) }
        if (OP == 22) { REP8(MOV) REP8(MOV) REP8(MOV) REP8(MOV) }
#define DTL(n) asm volatile("v_dot2c_i32_i16 %0, 0xff9b0047, %1" : "+v"(a##n) : "v"(b));
#define ADL(n) asm volatile("v_add_u32 %0, 0x12345678, %0" : "+v"(a##n));
        if (OP == 23) { REP8(DTL) REP8(DTL) REP8(DTL) REP8(DTL) }   // 32-bit literal operand (8-byte encoding)
        if (OP == 24) { REP8(ADL) REP8(ADL) REP8(ADL) REP8(ADL) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

static int g_wg_per_cu = 8;
template <int OP> void run(const char *name, unsigned *d)
{
    const int wgs = 256 * g_wg_per_cu, iters = 4096; // g_wg_per_cu workgroups of 4 waves per CU = that many waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) k<OP><<<wgs, 256>>>(d, iters, i);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<OP><<<wgs, 256>>>(d, iters, i);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double ops = (double)wgs * 256 * iters * 32;
    printf("%-16s %.3f ms  %.2f T lane-ops/s  (%.2f lanes / clk / SIMD at 2.4 GHz)\n", name, ms, ops / ms / 1e9, ops / (ms * 1e-3) / (1024 * 2.4e9));
}

int main(int argc, char **argv)
{
    unsigned *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    if (argc > 1) g_wg_per_cu = atoi(argv[1]);
    printf("waves per SIMD: %d\n", g_wg_per_cu);
    run<0>("v_mad_i32_i24", d);
    run<1>("v_dot2c_i32_i16", d);
    run<2>("v_add_u32", d);
    run<3>("v_xor_b32", d);
    run<4>("v_perm_b32", d);
    run<5>("v_mul_lo_u32", d);
    run<6>("mad+add 1:1", d);
    run<7>("perm+xor 1:1", d);
    run<8>("dot2c+add 1:1", d);
    run<9>("16 mad, 16 add", d);
    run<10>("mad+add 3:1", d);
    run<11>("v_mul_i32_i24", d);
    run<12>("v_add3_u32", d);
    run<13>("v_lshl_add_u32", d);
    run<14>("v_bitop3_b32", d);
    run<15>("v_and_b32", d);
    run<16>("v_lshrrev_b32", d);
    run<17>("v_ashrrev_i32", d);
    run<18>("v_bfe_u32", d);
    run<19>("v_pk_add_u16", d);
    run<20>("v_dot2_i32_i16", d);
    run<21>("v_dot4_i32_i8", d);
    run<22>("v_mov_b32", d);
    run<23>("v_dot2c literal", d);
    run<24>("v_add_u32 literal", d);
    return 0;
}
