// gf_encode128_fft.h -- the CM256 encoder for 128 originals and up to 32 recovery blocks as an additive FFT (round 5).
// Include inside namespace sdrhip { namespace { ... } } behind gf_encode128_body.h (kmul's table format, lds_addr, x3, GF_NT).
//
// Same bytes as gf_encode128_wg (UDPSinkFEC::transmitUDP's cm256_encode, UDPSinkFEC.cpp:228-256), a different algorithm.
// gf_encode128_body.h writes the Cauchy rows as recovery_r = P ^ r * S(128 ^ r), S(x) = XOR_j d_j / (x ^ j), and evaluates S as
// 16-point XOR-convolutions by Karatsuba: 16 blocks x 81 = 1296 constant multiplications and ~3100 XORs per 4-byte column.
// S is a rational function with a very regular denominator: the poles j = 0..127 are the GF(2)-subspace V7 of GF(256), so
// S = N / s_7 with s_7 the subspace polynomial of V7 (linearised: constant, q, on the whole coset 128 + V7 where the recovery
// rows are evaluated) and N the polynomial of degree < 128 that takes the values c d_j at the points j (c = s_7'(0)).  In the
// polynomial basis of Lin, Chung and Han (X_i = product over the set bits k of i of s^_k, the normalised subspace polynomials of
// V_k) interpolation and evaluation on cosets are butterfly networks like a radix-2 FFT's -- a stage-k butterfly is
// a ^= const * b, b ^= a (inverse: b ^= a, a ^= const * b) with ONE constant per block, s^_k of the block's coset representative:
//   1. coefficients of N / c from the data: the inverse transform of size 128 on V7.  Done as two halves of 64 through one
//      routine (stages 0..5; the constants of the coset 64 + V6 differ from those of V6); the last stage's constant is 0, it
//      only XORs the halves, and by linearity that XOR moves behind the fold of step 2;
//   2. fold onto the coset 128 + V5: on it s^_5 and s^_6 are constants (t5, t6), so the 128 coefficients collapse to 32:
//      e_i = f_i ^ t5 f_{32+i} per half, then e = e_lo ^ t6 (e_hi ^ e_lo);
//   3. the transform of size 32 on 128 + V5: values N(128 ^ r) / c, r = 0..31.  Its first stage (one block) couples rows r and
//      16 + r, behind it the two halves of 16 are independent;
//   4. recovery_r = P ^ (r c / q) * value_r.
// 384 - 63 (the leading block of every stage of the first half has the constant 0: skipped) + 64 + 32 + 80 + 32 = 529 constant
// multiplications and ~1200 XORs per column: ~45 % of the instructions of the Karatsuba walk (measured: 12.6 M against 25.6 M
// VALU wave-instructions per 1040 frames).
// (tools/experiments_r05/lch_encode_proto.py is the same algorithm in numpy, checked against the oracle's cm256_encode.)
//
// Mapping: a lane owns one 4-byte column of one frame; a WORKGROUP is one frame, its four waves = (column half ch, block half hf):
// a wave loads 64 blocks of its 64 columns (and writes them into the frame area where the framing copy is fused), runs the
// size-64 inverse transform and the t5 fold in registers (64 values per lane); the hf = 0 wave hands its 32 folded coefficients
// to the hf = 1 wave through LDS (barrier), that one applies t6 and the first stage of the size-32 transform and hands rows
// 0..15 back (barrier); then each wave finishes a size-16 transform and 16 recovery rows.  A frame = 4 waves instead of the
// first version's 2 (a wave = a frame half with all 128 blocks: 96 live values, spilled at 128 registers, and 2080 waves of
// 14 us each on 1024 SIMDs is a two-and-a-bit-round launch); 95 registers, 23.5 KB of LDS: four to five workgroups per CU.
#pragma once

#ifndef ENC_LOAD_AUX
#define ENC_LOAD_AUX 0 // (buffer cache policy of the 64 data loads; 2 = nt: see gf_decode128_fft.h, DEC_LOAD_AUX.  The encoder's input was written by the launch in front of it)
#endif
constexpr int FFT_NTAB = 192;                       // gf256.h: CM256_FFT_TABLES
constexpr int FFT_TAB_BYTES = FFT_NTAB * 32;        // the 32-byte records of gf_build_tables as they are: {Ta, Tb} 16 B, {Tc} 4 B, 12 B unused (ONE address register)
// + per column half 32 x 64 dwords of exchange (e_lo down, rows 0..15 back up) and 2 x 64 of parity
constexpr int FFT_XCH_DWORDS = 32 * 64 + 2 * 64;
constexpr int ENC128_FFT_LDS_BYTES = FFT_TAB_BYTES + 2 * FFT_XCH_DWORDS * 4;
constexpr int FFT_MAX_ROWS = 32;
#ifndef FFT_WAVES_PER_EU
#define FFT_WAVES_PER_EU 4 // (the compiler's register budget: 128; the kernel takes 95, so five waves per SIMD are resident where the LDS allows)
#endif

#ifdef FFT_STAMPS
// timeline experiment (tools/experiments_r05/fft_stamps.py, dec_stamps.py, tools/experiments_r06/stagger_rank.py): lane 0 of every wave of
// the first 2048 workgroups leaves s_memrealtime (100 MHz) at up to eight points (+ HW_ID).  FFT_STAMP_SET picks which points: 0 = the
// phases of the kernel, 1 = inside the decoder's plan, 2 = inside the size-64 inverse transform; points 0 (start) and 7 (end) in every set
#ifndef FFT_STAMP_SET
#define FFT_STAMP_SET 0
#endif
__device__ unsigned long long g_fft_stamps[8192 * 8];
#define FFT_STAMP_RAW(k) do { if (blockIdx.x < 2048 && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_fft_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)
#define FFT_STAMP(k) do { if (FFT_STAMP_SET == 0 || (k) == 0 || (k) == 7) FFT_STAMP_RAW(k); } while (0)
#define PLAN_STAMP(k) do { if (FFT_STAMP_SET == 1) FFT_STAMP_RAW(k); } while (0)
#define INV_STAMP(k) do { if (FFT_STAMP_SET == 2) FFT_STAMP_RAW(k); } while (0)
#else
#define FFT_STAMP(k) do { } while (0)
#define PLAN_STAMP(k) do { } while (0)
#define INV_STAMP(k) do { } while (0)
#endif

// The multiplier tables travel through two or three register sets: while block n is multiplied, the table of block n + 1 (inverse
// transform: and n + 2) is on its way from
// LDS (the experience of the Karatsuba walk, gf_encode128_body.h: left to the compiler every table load of the unrolled network is
// hoisted to the top -- immediate addresses, no dependencies -- and the registers spill; with load + wait inside every block the
// wave waits ~100 cycles 190 times).  asm statements keep the loads in program order; the wait statement re-defines the registers
// ("+v"), so nothing reads them between issue and wait.
struct FftTabs {
    uint4_t t[3]; // (the third set: the size-64 inverse transform runs TWO blocks ahead -- its first stage has one butterfly per block)
    unsigned c[3];
};
template <int P, int IDX> __device__ __forceinline__ void fft_issue(FftTabs &R, unsigned la)
{
    asm volatile("ds_read_b128 %0, %2 offset:%c3\n\tds_read_b32 %1, %2 offset:%c4"
                 : "=&v"(R.t[P]), "=&v"(R.c[P]) : "v"(la), "i"(IDX * 32), "i"(IDX * 32 + 16) : "memory");
}
template <int P> __device__ __forceinline__ void fft_wait(FftTabs &R)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R.t[P]), "+v"(R.c[P])::"memory");
}
// ... while the NEXT block's table (two LDS reads, issued behind this one's: LDS operations of a wave return in order) stays in flight
template <int P> __device__ __forceinline__ void fft_wait_ahead(FftTabs &R)
{
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(R.t[P]), "+v"(R.c[P])::"memory");
}
// a ^= T * b in four-input XOR form (v_bitop3 + v_xor)
template <int P> __device__ __forceinline__ void fft_muladd(unsigned &a, unsigned &b, const FftTabs &R)
{
    // (fences in front and behind: asm statements keep their order, so the butterflies run one after the other -- dependent VALU
    // instructions issue back to back on this machine, nothing is lost -- and neither the selector words of a whole block of
    // butterflies -- three registers each, up to 32 of them -- are formed first nor those of the next stage hoisted over a table wait)
    asm volatile("" : "+v"(b));
    const unsigned sa = b & 0x07070707u, sb = (b >> 3) & 0x07070707u, sc = (b >> 6) & 0x03030303u;
    a = x3(a, __builtin_amdgcn_perm(R.t[P].y, R.t[P].x, sa), __builtin_amdgcn_perm(R.t[P].w, R.t[P].z, sb)) ^ __builtin_amdgcn_perm(0u, R.c[P], sc);
    asm volatile("" : "+v"(a));
}

// the lane's index in its wave, formed anew wherever it is called (the opaque zero keeps the calls apart): what follows from it --
// column, lane offset, exchange address -- need not stay in registers across the transform (they were the kernel's last spills)
__device__ __forceinline__ unsigned fft_lane()
{
    unsigned z = 0u;
    asm volatile("" : "+v"(z));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

template <class F, int... Is> __device__ __forceinline__ void fft_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void fft_for(F &&f) { fft_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The 63 blocks of the size-64 inverse transform in DEPTH-FIRST order: position n = 0..30 = stages 0..4 of elements 0..31, n = 31..61 the
// same for elements 32..63, n = 62 the one block of stage 5.  (Breadth first -- all of stage 0 first -- the network needs all 64 values
// before its first butterfly is through; this way the first half runs while the second half's loads are still on their way, and the
// caller's `mid` hook -- stores and parity of the second half -- sits between the halves.)  Stage k of the whole transform has 32 >> k
// blocks, block j works on elements [2 j 2^k, 2 (j + 1) 2^k) with table 64 - (64 >> k) + j (+ 63 for the second block half).
struct FftBlk { int k, j; };
__host__ __device__ constexpr FftBlk fft_inv_block(int n)
{
    if (n >= 62) return FftBlk{5, 0};
    const int g = n / 31, p = n % 31;
    int k = 0;
    while (k < 4 && p >= 32 - (32 >> (k + 1))) ++k; // (a group of 32 elements: 16 >> k blocks in stage k, 32 - (32 >> k) in front of it)
    return FftBlk{k, g * (16 >> k) + (p - (32 - (32 >> k)))};
}
__host__ __device__ constexpr int fft_inv_table(int n) { return n >= 63 ? 126 : 64 - (64 >> fft_inv_block(n).k) + fft_inv_block(n).j; }

// inverse transform of size 64 (values on the coset 64 hf + V6 -> novel-basis coefficients) and the t5 fold: d[0..31] = the half's
// 32 coefficients on 128 + V5.  lh: LDS address of table 0 of this half; la: of table 0.  mid(): called between the two halves of 32.
template <int HF, class MID> __device__ __forceinline__ void fft_inverse64_fold(unsigned (&d)[64], unsigned lh, unsigned la, MID &&mid)
{
    FftTabs R;
    fft_issue<0, fft_inv_table(0)>(R, lh);
    fft_issue<1, fft_inv_table(1)>(R, lh);
    fft_for<63>([&](auto nc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value, k = fft_inv_block(n).k, j = fft_inv_block(n).j, h = 1 << k, blk = j * 2 * h, P = n % 3;
        if constexpr (n == 31) { INV_STAMP(3); mid(); INV_STAMP(4); }
        fft_wait_ahead<P>(R); // (position n + 1's table -- t5 behind the last block -- is on its way)
        if constexpr (n + 2 < 63) fft_issue<(n + 2) % 3, fft_inv_table(n + 2)>(R, lh);
        else if constexpr (n + 2 == 63) fft_issue<(n + 2) % 3, 126>(R, la); // t5
#pragma unroll
        for (int i = 0; i < h; ++i) d[blk + h + i] ^= d[blk + i];
        // (the leading block of a stage sits on the coset representative 64 hf: s^_k(0) = 0, nothing to multiply in the first half)
        if constexpr (j != 0 || HF != 0) {
#pragma unroll
            for (int i = 0; i < h; ++i) fft_muladd<P>(d[blk + i], d[blk + h + i], R);
        }
    });
    INV_STAMP(5);
    fft_wait<63 % 3>(R);
#pragma unroll
    for (int i = 0; i < 32; ++i) fft_muladd<63 % 3>(d[i], d[32 + i], R);
    INV_STAMP(6);
}

// rows 16 hh .. 16 hh + 15 of the size-32 transform on 128 + V5 behind its first stage: stages 3..0 on 16 values; the tables of
// stage k are 128 + (32 - (32 >> k)) + (8 >> k) hh + j.
__device__ __forceinline__ void fft_forward16(unsigned (&e)[16], int hh, unsigned la)
{
    FftTabs R;
    // blocks in order: n = 0 (k = 3), 1..2 (k = 2), 3..6 (k = 1), 7..14 (k = 0)
    auto base = [&](int k) { return la + (unsigned)hh * (unsigned)((8 >> k) * 32); };
    fft_issue<0, 128 + 32 - 4>(R, base(3));
    fft_for<15>([&](auto nc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
        constexpr int k = n < 1 ? 3 : n < 3 ? 2 : n < 7 ? 1 : 0;
        constexpr int j = n - ((8 >> k) - 1); // blocks before stage k: 0, 1, 3, 7
        constexpr int h = 1 << k, blk = j * 2 * h, P = n & 1;
        fft_wait<P>(R);
        if constexpr (n + 1 < 15) {
            constexpr int n1 = n + 1, k1 = n1 < 1 ? 3 : n1 < 3 ? 2 : n1 < 7 ? 1 : 0, j1 = n1 - ((8 >> k1) - 1);
            fft_issue<P ^ 1, 128 + 32 - (32 >> k1) + j1>(R, base(k1));
        }
#pragma unroll
        for (int i = 0; i < h; ++i) {
            fft_muladd<P>(e[blk + i], e[blk + h + i], R);
            e[blk + h + i] ^= e[blk + i];
        }
    });
}

// the workgroup's tables: global (32-byte records) -> LDS {16 B} + {4 B} arrays.  All threads; a barrier follows.
__device__ __forceinline__ void fft_fill_tables(const Enc128Args &a, unsigned char *ldsraw)
{
    // (both loads of a thread are issued before the first is written: as a copy loop the two iterations were load, wait, write twice --
    // two global round trips in front of the workgroup's data loads.  In asm: written in C the same statements moved the register
    // allocation of the whole kernel from 96 to 127 registers, i.e. from five to four workgroups per CU)
    static_assert(GF_NT == 256 && 2 * FFT_NTAB == 384, "two loads per thread");
    const unsigned tid = threadIdx.x, i1 = tid < 128u ? 256u + tid : tid; // (threads 128..255: their own entry again)
    const uint4_t *src = reinterpret_cast<const uint4_t *>(a.fft_tables);
    uint4_t f0, f1;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(f0), "=&v"(f1) : "v"(src + tid), "v"(src + i1) : "memory");
    uint4_t *lt = reinterpret_cast<uint4_t *>(ldsraw);
    lt[tid] = f0;
    lt[i1] = f1;
    __syncthreads();
}

// from a wave's 64 values (blocks 64 hf .. 64 hf + 63 of its 64 columns) to rows 16 hf .. 16 hf + 15 of the size-32 transform on
// 128 + V5: inverse transform + t5 fold in registers, the hf = 0 wave's 32 coefficients go to the hf = 1 wave through the column
// half's exchange area xch0 ([32][64] dwords; barrier), that one applies t6 and the first stage of the size-32 transform and hands
// rows 0..15 back (barrier), then each wave finishes its size-16 transform.  Two __syncthreads inside: all four waves call it.
// HF is a template parameter: the two block halves run different code (the first half's stages begin with a block whose constant
// is zero -- 63 of its 224 multiplications -- and the exchange is not symmetric); as a run-time value the skipped blocks became
// branches inside the network and the register allocator spilled at every join.
// post(): called behind the fold, in front of the exchange -- the decoder issues its recovery-row loads there (the second half of d[]
// is free from here on): they used to go out behind the exchange, right in front of their use -- a global round trip on the path.
template <int HF, class MID, class POST> __device__ __forceinline__ void fft_rows16(unsigned (&d)[64], unsigned (&e)[16], unsigned la, unsigned *xch0, MID &&mid, POST &&post)
{
    constexpr int hf = HF;
    FFT_STAMP(2);
    INV_STAMP(2);
    fft_inverse64_fold<HF>(d, la + (unsigned)(hf * 63 * 32), la, mid);
    post();
    FFT_STAMP(3);
    unsigned *const xch = xch0 + fft_lane();
    if constexpr (hf == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) xch[i * 64] = d[i];
    }
    __syncthreads();
    if constexpr (hf != 0) {
        // t6 and the one block of stage 4, rows i and 16 + i together: the pair is finished (row i parked for the other wave, row
        // 16 + i kept) before the next one is read -- 64 + 16 live values instead of 96
        FftTabs R;
        fft_issue<0, 127>(R, la);           // t6
        fft_issue<1, 128 + 32 - 2>(R, la);  // stage 4
        fft_wait<0>(R);
        fft_wait<1>(R);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned va = xch[i * 64], vb = xch[(16 + i) * 64];
            d[i] ^= va; fft_muladd<0>(va, d[i], R);
            d[16 + i] ^= vb; fft_muladd<0>(vb, d[16 + i], R);
            fft_muladd<1>(va, vb, R);
            e[i] = vb ^ va;
            xch[i * 64] = va;
        }
    }
    __syncthreads();
    if constexpr (hf == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = xch[i * 64];
    }
    FFT_STAMP(4);
    fft_forward16(e, hf, la);
}

// one WORKGROUP of the encoder: frame list entry `fi`, a.rows <= FFT_MAX_ROWS.  Input handling (frame list, meta block derived in
// place, fused framing copy) is gf_encode128_wg's, block for block.
// (HF = the wave's block half as a template parameter: the two halves run different code, see fft_rows16; the dispatch sits at the
// very top -- gf_encode128_fft_wg -- so that no register value has to survive a join of the two variants)
// (xslot: which of the workgroup's exchange areas the column half uses -- ch in a whole-frame workgroup, 0 in a half-frame one)
template <int HF> __device__ __forceinline__ void gf_encode128_fft_wave(const Enc128Args &a, int fi, unsigned char *ldsraw, int ch, int xslot)
{
    constexpr int hf = HF;
    const unsigned la = lds_addr(ldsraw);
    const int lane = (int)fft_lane();
    const int fr = a.gen_done > 0 ? (fi / a.gen_done) * a.gen_cap + fi % a.gen_done : (a.frame_list ? __builtin_amdgcn_readfirstlane(a.frame_list[fi]) : fi);
    if (fr < 0 || fr >= a.nframes) return; // (workgroup-uniform: all four waves leave in front of the barriers below)
    const int col = ch * 64 + lane;
    const bool live = col < 127;            // (lane 63 of the second half has no column: it walks column 126 again and stores nothing)
    const unsigned lc = live ? (unsigned)col : 126u;
    // Addressing through buffer descriptors: a uniform base (4 SGPRs per source), a uniform block offset (one scalar add per
    // access) and ONE lane offset register between a wave's 64 loads and 64 stores.  (Written with pointers the compiler formed a
    // 64-bit vector address per access: 128 registers of addresses in flight, 500 dwords of spills.)
    const unsigned *fbase = reinterpret_cast<const unsigned *>(a.in + (size_t)fr * a.in_frame_bytes) + 1;  // dword 0 of block 0's payload
    unsigned *obase = reinterpret_cast<unsigned *>(a.out + (size_t)fr * a.out_frame_bytes) + 1;
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(fbase), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0x7fffffff, 0x00020000);
    const unsigned lc4 = 4u * lc;
    // block 0 of the frame (header + meta block): from memory, or (K2 runs in this very launch) derived like K2 derives it
    unsigned hdr0 = 0u, blk0 = 0u;
    bool own0 = false;
    if (a.meta_count > 0 && a.gen_done > 0) {
        const int f = fr % a.gen_cap, mi = f - a.meta_first;
        if (mi >= 0 && mi < a.meta_count) { // (workgroup-uniform: the shuffle inside frame_meta_words sees whole waves)
            unsigned w[6];
            frame_meta_words(a.meta_w, a.meta_idx0, a.meta_rate, mi, w);
            own0 = true;
            hdr0 = (a.meta_frame_count0 + (unsigned)mi) & 0xffffu;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (col == k) blk0 = w[k];
        }
    }
    if (!own0) hdr0 = fbase[-1]; // (uniform load: {frameIndex, blockIndex 0, 0})
    // fused framing: blocks 1..127 of this frame come straight from the decimated stream (127 samples each): block b at
    // lbase[127 b + col].  The frame that was open when the call began (lin_straddle): its first lin_pending samples are in the
    // frame area, the rest comes now, sample w of the call at sbase[w].
    bool fused = false, strad = false;
    const unsigned *lbase = fbase, *sbase = fbase;
    if (a.lin) {
        const int s = fr / a.lin_cap, f = fr - s * a.lin_cap;
        if (f >= a.lin_first) {
            fused = true;
            lbase = a.lin + (size_t)s * a.lin_stride + ((size_t)f * 16129u - (size_t)a.lin_pending) - 127;
        } else if (a.lin_straddle && f == 0) {
            strad = true;
            sbase = a.lin + (size_t)s * a.lin_stride;
        }
    }
    unsigned *const xch0 = reinterpret_cast<unsigned *>(ldsraw + FFT_TAB_BYTES) + xslot * FFT_XCH_DWORDS; // [i][lane], i < 32; parity at [32], [33]

    // ONE load sequence for both sources (descriptor and block pitch are picked once, uniformly): two sequences that define the
    // same 64 registers met in a join the register allocator answered with a few hundred moves and spills
    unsigned d[64];
    const int b0 = 64 * hf;
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(lbase), 0, 0x7fffffff, 0x00020000);
    const int pitch = fused ? 508 : 512;
    {
        // (block 0 = the meta block is never in the stream: the first half's wave takes it from the frame area or derives it)
        const unsigned v0 = __builtin_amdgcn_raw_buffer_load_b32(rf, lc4, 0, 0);
        const unsigned vb = __builtin_amdgcn_raw_buffer_load_b32(rl, lc4, (hf ? b0 : 1) * pitch, 0);
        d[0] = hf ? vb : (own0 ? blk0 : v0);
    }
#pragma unroll
    for (int i = 1; i < 64; ++i) d[i] = __builtin_amdgcn_raw_buffer_load_b32(rl, lc4, (b0 + i) * pitch, ENC_LOAD_AUX);
    if (strad) {
        // the straddling frame went through the sequence above as a frame in memory; now every lane replaces what the stream holds
        // (in place, sixteen blocks at a time: a second sequence DEFINING the 64 registers is what the allocator could not join)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(sbase), 0, 0x7fffffff, 0x00020000);
        fft_for<4>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            unsigned vl[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int w = (b0 + 16 * g + i - 1) * 127 - a.lin_pending + (int)lc; // sample of the call this lane's dword of the block is
                vl[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, w < 0 ? 0u : 4u * (unsigned)w, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = b0 + 16 * g + i, w = (b - 1) * 127 - a.lin_pending + (int)lc;
                d[16 * g + i] = (b != 0 && w >= 0) ? vl[i] : d[16 * g + i];
            }
            asm volatile("" ::: "memory");
        });
    }
    if ((fused || strad) && live) { // (the straddling frame: dwords that came from the area are rewritten as they are)
#pragma unroll
        for (int i = 0; i < 64; ++i) __builtin_amdgcn_raw_buffer_store_b32(d[i], rf, lc4, (b0 + i) * 512, 0);
    }
    // parity: of the first 32 values here, of the other 32 between the two halves of the transform (they may still be on their way
    // while the first half runs); both block halves' parities wait in LDS: one long-lived register less
    unsigned par = 0u;
#pragma unroll
    for (int i = 0; i < 32; i += 2) par = x3(par, d[i], d[i + 1]);
    unsigned e[16];
    fft_rows16<HF>(d, e, la, xch0, [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 32; i < 64; i += 2) par = x3(par, d[i], d[i + 1]);
        (xch0 + fft_lane())[(32 + hf) * 64] = par;
    }, []() __attribute__((always_inline)) {});
    // rows 16 hf + i
    {
        FftTabs R;
        const unsigned ln = fft_lane(), col = (unsigned)ch * 64u + ln;
        const bool live = col < 127u;
        const unsigned lc = live ? col : 126u, lc4 = 4u * lc;
        const unsigned par = (xch0 + ln)[32 * 64] ^ (xch0 + ln)[33 * 64];
        const unsigned lk = la + (unsigned)hf * 512u;
        fft_issue<0, 160>(R, lk);
        fft_for<16>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, P = i & 1;
            fft_wait<P>(R);
            if constexpr (i + 1 < 16) fft_issue<P ^ 1, 160 + i + 1>(R, lk);
            const int r = 16 * hf + i;
            if (r < a.rows && live) {
                unsigned v = par;
                fft_muladd<P>(v, e[i], R);
                __builtin_amdgcn_raw_buffer_store_b32(v, ro, lc4, r * 512, 0);
                // header {frameIndex (of the frame's block 0), 128 + r, filler 0}, UDPSinkFEC.cpp:239-243
                if (col == 0) (obase + (size_t)r * 128)[-1] = (hdr0 & 0xffffu) | ((unsigned)(128 + r) << 16);
            }
        });
    }
    FFT_STAMP(5);
#ifdef FFT_STAMPS
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_fft_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 6] = hw; }
#endif
}

#ifndef FFT_LDS_PAD
#define FFT_LDS_PAD 0 // (experiment: extra bytes of LDS per workgroup = fewer workgroups per CU)
#endif
constexpr int ENC128_FFT_KERNEL_LDS = ENC128_FFT_LDS_BYTES + FFT_LDS_PAD;
__device__ __forceinline__ void gf_encode128_fft_wg(const Enc128Args &a, int fi, unsigned char *ldsraw)
{
    FFT_STAMP(0);
    fft_fill_tables(a, ldsraw);
    fec_stagger_sleep(fi, a.stagger, a.stagger_div);
    FFT_STAMP(1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wv >> 1) gf_encode128_fft_wave<1>(a, fi, ldsraw, wv & 1, wv & 1);
    else gf_encode128_fft_wave<0>(a, fi, ldsraw, wv & 1, wv & 1);
}

// Half-frame workgroups (round 6, last session; context option enc_units = half): the two column halves of a frame share nothing but the
// tables, so a workgroup can be ONE column half -- two waves (block halves), 128 threads, 6 KB of tables + one exchange area.  The launch
// lasts as long as the CU with the most work (profiles/r06_enc_count_scan.txt: ~10 us + 5.1 us x ceil(frames / 256)); in half frames the
// Rx step's 1040 frames are 8.1 -> 9 units on the fullest CU = 4.5 frames instead of 5.
constexpr int ENC128_FFT_HALF_LDS = FFT_TAB_BYTES + FFT_XCH_DWORDS * 4;
__device__ __forceinline__ void fft_fill_tables_half(const Enc128Args &a, unsigned char *ldsraw)
{
    static_assert(FFT_NTAB * 2 == 3 * 128, "three loads per thread");
    const unsigned tid = threadIdx.x;
    const uint4_t *src = reinterpret_cast<const uint4_t *>(a.fft_tables);
    uint4_t f0, f1, f2;
    asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_dwordx4 %2, %5, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(f0), "=&v"(f1), "=&v"(f2) : "v"(src + tid), "v"(src + 128u + tid), "v"(src + 256u + tid) : "memory");
    uint4_t *lt = reinterpret_cast<uint4_t *>(ldsraw);
    lt[tid] = f0;
    lt[128u + tid] = f1;
    lt[256u + tid] = f2;
    __syncthreads();
}
__device__ __forceinline__ void gf_encode128_fft_half_wg(const Enc128Args &a, int ui, unsigned char *ldsraw)
{
    const int fi = ui >> 1, ch = ui & 1;
    fft_fill_tables_half(a, ldsraw);
    fec_stagger_sleep(fi, a.stagger, a.stagger_div);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wv) gf_encode128_fft_wave<1>(a, fi, ldsraw, ch, 0);
    else gf_encode128_fft_wave<0>(a, fi, ldsraw, ch, 0);
}
__device__ __forceinline__ void gf_encode128_fft_unit(const Enc128Args &a, int fi, unsigned char *ldsraw) { gf_encode128_fft_wg(a, fi, ldsraw); }
