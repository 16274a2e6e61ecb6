// gf_encode128_body.h -- the structured CM256 encoder for 128 originals as device code shared by gf_kernels.hip (stand-alone
// kernel) and decim_mfma.hip (fused Rx launch).  Include inside namespace sdrhip { namespace { ... } } after uint4_t is defined.
#pragma once
constexpr int GF_NT = 256;
// ------------------------------------------------------------------------------------------
// Structured encoder for OriginalCount = 128 (the only geometry sdrdaemon uses, UDPSinkFEC.h:57).
//
// The Cauchy element (y_j ^ x_0) / (x_i ^ y_j) with x_0 = 128, x_i = 128 + r, y_j = j is
//     M[r][j] = (128 ^ j) / (128 ^ (r ^ j)) = 1 ^ r * G[r ^ j],      G[t] = 1 / (128 ^ t)
// (because (128 ^ j) ^ (128 ^ (r ^ j)) = r), hence
//     recovery_r = P ^ r * (G (*) x)[r],   P = XOR of the 128 originals,
// where (*) is an XOR-convolution over the block index: (G (*) x)[r] = XOR_j G[r ^ j] x_j.  Splitting
// j = 16 cb + jl and r = 16 rt + rl turns it into 16-point dyadic convolutions with the kernel blocks
// G_b[u] = G[16 b + u], b = rt ^ cb, and over a field of characteristic 2 those obey a Karatsuba rule
//     y_lo = g_lo(*)z_lo ^ g_hi(*)z_hi,   y_hi = y_lo ^ (g_lo ^ g_hi)(*)(z_lo ^ z_hi)
// (3 half-size products instead of 4).  Four levels: 81 constant multiplications + 195 XORs per
// 16 x 16 block instead of 256 multiplications.  The 81 leaf constants of each of the 8 blocks G_b come
// from the host (gf256.cpp, cm256_karatsuba_leaf_tables) in depth-first order as ready-made 32-byte
// multiplier tables and sit in LDS as {Ta, Tb} (16 B) + {Tc} (4 B) arrays: immediate offsets, no
// dependent loads.
// A lane owns one 4-byte column of a frame; a workgroup = one (frame, half block) = 64 columns; its four
// waves take two column blocks each, apply them to BOTH 16-row tiles of a row pair in one walk of the
// tree (acc_conv2), XOR their partial sums into LDS (ds_xor), then each wave finishes 8 of the 32 rows:
// recovery_r = P ^ r * c_r, and writes their {frameIndex, 128 + r, 0} headers.
__device__ __forceinline__ unsigned kmul(unsigned zval, const uint4_t &t, unsigned tc)
{
    const unsigned sa = zval & 0x07070707u, sb = (zval >> 3) & 0x07070707u, sc = (zval >> 6) & 0x03030303u;
    return __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_perm(t.y, t.x, sa), __builtin_amdgcn_perm(t.w, t.z, sb), __builtin_amdgcn_perm(0u, tc, sc), 0x96);
}

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; // generic -> LDS byte address
}

constexpr int KN = 16;                                                                    // points per block
constexpr int KLEAVES = 81;                                                               // 3^4
__host__ __device__ constexpr int pow3(int n) { return n <= 1 ? 1 : 3 * pow3(n / 2); }    // leaves of an n-point node
__host__ __device__ constexpr int sbase(int n) { return KN + (KN - n); }                  // scratch of the level of size n

// the multiplier tables of the leaves travel through two register sets: while leaf L is multiplied, the tables of leaf L + 1 are
// already on their way from LDS (round 4: with load + wait inside every leaf a wave spent most of its time in s_waitcnt --
// SQ_WAIT_ANY 57 % of the decoder's wave cycles, `tools/experiments_r04/exp14.sh` -- four waves per SIMD do not cover 162
// dependent LDS round trips per walk)
// (The loads are started in one asm statement and awaited in a later one; the hardware does not interlock registers that wait for LDS
// data, so nothing may touch them in between: tools/check_asm_tables.py -- tests/test_asm_guard.py -- proves that on the compiled
// assembly of every kernel, for this walk and for the additive-FFT one.)
struct LeafRegs {
    uint4_t ta[2], tb[2];
    unsigned tca[2], tcb[2];
};
template <int LEAF, int P> __device__ __forceinline__ void leaf_issue(LeafRegs &R, unsigned la16, unsigned la4, unsigned lb16, unsigned lb4)
{
    asm volatile("ds_read_b128 %0, %4 offset:%c8\n\tds_read_b32 %1, %5 offset:%c9\n\t"
                 "ds_read_b128 %2, %6 offset:%c8\n\tds_read_b32 %3, %7 offset:%c9"
                 : "=&v"(R.ta[P]), "=&v"(R.tca[P]), "=&v"(R.tb[P]), "=&v"(R.tcb[P])
                 : "v"(la16), "v"(la4), "v"(lb16), "v"(lb4), "i"(LEAF * 16), "i"(LEAF * 4)
                 : "memory");
}

// the three selector words of a data dword (kmul): bitwise functions of z, so sel(z ^ w) = sel(z) ^ sel(w)
struct Sel3 { unsigned a, b, c; };
__device__ __forceinline__ Sel3 sel_of(unsigned z)
{
    return Sel3{z & 0x07070707u, (z >> 3) & 0x07070707u, (z >> 6) & 0x03030303u};
}
__device__ __forceinline__ Sel3 sel_xor(const Sel3 &x, const Sel3 &y) { return Sel3{x.a ^ y.a, x.b ^ y.b, x.c ^ y.c}; }

// the six v_perm products of one leaf for the two tiles (tables of leaf LEAF, fetched one leaf ahead; the next leaf's go out here)
struct LeafProd { unsigned a0, a1, a2, b0, b1, b2; };
template <int LEAF>
__device__ __forceinline__ LeafProd leaf_products(Sel3 &s, unsigned la16, unsigned la4, unsigned lb16, unsigned lb4, LeafRegs &R)
{
    constexpr int P = LEAF & 1;
    // (the selector words are tied to the statement: whatever forms them -- an XOR of two earlier leaves' words -- stays here)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R.ta[P]), "+v"(R.tca[P]), "+v"(R.tb[P]), "+v"(R.tcb[P]), "+v"(s.a), "+v"(s.b), "+v"(s.c)::"memory");
    if constexpr (LEAF + 1 < KLEAVES) leaf_issue<LEAF + 1, P ^ 1>(R, la16, la4, lb16, lb4);
    const uint4_t ta = R.ta[P], tb = R.tb[P];
    const unsigned tca = R.tca[P], tcb = R.tcb[P];
    return LeafProd{__builtin_amdgcn_perm(ta.y, ta.x, s.a), __builtin_amdgcn_perm(ta.w, ta.z, s.b), __builtin_amdgcn_perm(0u, tca, s.c),
                    __builtin_amdgcn_perm(tb.y, tb.x, s.a), __builtin_amdgcn_perm(tb.w, tb.z, s.b), __builtin_amdgcn_perm(0u, tcb, s.c)};
}
__device__ __forceinline__ unsigned x3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); } // a ^ b ^ c: one v_bitop3_b32

// ya[YO .. YO+N) ^= ga (*) v[ZO .. ZO+N) and yb[..] ^= gb (*) v[..] in one walk of the Karatsuba tree: the two
// kernel blocks (the two 16-row tiles of a row pair) see the same z sums, so the tree's XORs on the z side
// and the three selector words of every leaf are formed once and used twice.  la16 / la4, lb16 / lb4 = LDS
// byte addresses of the leaf tables of the two blocks.  The caller issues leaf 0 (leaf_issue<0, 0>) in front of the walk.
// (The 81 table loads of a block have immediate addresses; left to the compiler they are all hoisted to the top -- hundreds of
// VGPRs of tables -- and spilled: asm statements keep them in program order, one leaf ahead.)
template <int N, int ZO, int YO, int LEAF0>
__device__ __forceinline__ void acc_conv2(unsigned (&v)[2 * KN - 1], unsigned (&ya)[KN], unsigned (&yb)[KN], unsigned la16, unsigned la4,
                                          unsigned lb16, unsigned lb4, LeafRegs &R)
{
    if constexpr (N == 4) {
        // The lowest two levels of the tree, hand-scheduled (round 4):
        //  * SELECTOR domain: the nine leaves of a 4-point node need the selector words of z0..z3 and of five XOR combinations of
        //    them; formed from the four inputs' words (4 x 5 + 5 x 3 instructions) instead of from nine z values (5 XORs + 9 x 5);
        //  * the y side in 3-input XORs (v_bitop3_b32): written out, the node is y0 ^= P0 ^ P1 ^ P3 ^ P4, y1 ^= y0o ^ P2 ^ P5 ^ y0',
        //    y2 ^= y0o ^ P6 ^ P7 ^ y0', y3 ^= y1o ^ y2 ^ y0o ^ P8 ^ y2b ^ y1' (Pk = the three v_perm products of leaf k; o = on
        //    entry, ' = final; the recursion's post-XOR of the first child and pre-XOR of the second cancel): 20 instructions per
        //    tile instead of 37 two-input XORs.  Measured: the encoder 0.0611 -> 0.0583 (selectors) -> 0.0518 ms per 1040 frames.
        // Same leaf order as the recursion below (the table layout does not change).
        unsigned &a0 = ya[YO], &a1 = ya[YO + 1], &a2 = ya[YO + 2], &a3 = ya[YO + 3];
        unsigned &b0 = yb[YO], &b1 = yb[YO + 1], &b2 = yb[YO + 2], &b3 = yb[YO + 3];
        a2 ^= a0; b2 ^= b0;                        // y2a = y2 ^ y0o
        a3 = x3(a3, a1, a2); b3 = x3(b3, b1, b2);  // y3b = y3 ^ y1o ^ y2a
        a1 ^= a0; b1 ^= b0;                        // y1 ^ y0o
        unsigned z0 = v[ZO], z1 = v[ZO + 1];
        asm volatile("" : "+v"(z0), "+v"(z1)); // (selector words are formed here, not where v[] is produced)
        Sel3 s0 = sel_of(z0), s1 = sel_of(z1);
        unsigned pa, pb; // a product waiting for a partner
        { const LeafProd q = leaf_products<LEAF0>(s0, la16, la4, lb16, lb4, R); a0 = x3(a0, q.a0, q.a1); b0 = x3(b0, q.b0, q.b1); pa = q.a2; pb = q.b2; }
        asm volatile("" : "+v"(a0), "+v"(b0), "+v"(pa), "+v"(pb)); // accumulate now: the compiler otherwise parks the products of many leaves
        { const LeafProd q = leaf_products<LEAF0 + 1>(s1, la16, la4, lb16, lb4, R); a0 = x3(x3(a0, pa, q.a0), q.a1, q.a2); b0 = x3(x3(b0, pb, q.b0), q.b1, q.b2); }
        asm volatile("" : "+v"(a0), "+v"(b0));
        { Sel3 t = sel_xor(s0, s1); const LeafProd q = leaf_products<LEAF0 + 2>(t, la16, la4, lb16, lb4, R); a1 = x3(a1, q.a0, q.a1) ^ q.a2; b1 = x3(b1, q.b0, q.b1) ^ q.b2; }
        asm volatile("" : "+v"(a1), "+v"(b1));
        unsigned z2 = v[ZO + 2], z3 = v[ZO + 3];
        asm volatile("" : "+v"(z2), "+v"(z3));
        Sel3 s2 = sel_of(z2), s3 = sel_of(z3);
        s0 = sel_xor(s0, s2); s1 = sel_xor(s1, s3); // (the third child's words now: twelve selector registers live, not fifteen)
        asm volatile("" : "+v"(s0.a), "+v"(s0.b), "+v"(s0.c), "+v"(s1.a), "+v"(s1.b), "+v"(s1.c));
        { const LeafProd q = leaf_products<LEAF0 + 3>(s2, la16, la4, lb16, lb4, R); a0 = x3(a0, q.a0, q.a1); b0 = x3(b0, q.b0, q.b1); pa = q.a2; pb = q.b2; }
        asm volatile("" : "+v"(a0), "+v"(b0), "+v"(pa), "+v"(pb));
        { const LeafProd q = leaf_products<LEAF0 + 4>(s3, la16, la4, lb16, lb4, R); a0 = x3(x3(a0, pa, q.a0), q.a1, q.a2); b0 = x3(x3(b0, pb, q.b0), q.b1, q.b2); }
        asm volatile("" : "+v"(a0), "+v"(b0)); // y0'
        s2 = sel_xor(s2, s3);
        { const LeafProd q = leaf_products<LEAF0 + 5>(s2, la16, la4, lb16, lb4, R); a1 = x3(x3(a1, q.a0, q.a1), q.a2, a0); b1 = x3(x3(b1, q.b0, q.b1), q.b2, b0); }
        asm volatile("" : "+v"(a1), "+v"(b1)); // y1'
        { const LeafProd q = leaf_products<LEAF0 + 6>(s0, la16, la4, lb16, lb4, R); a2 = x3(a2, q.a0, q.a1); b2 = x3(b2, q.b0, q.b1); pa = q.a2; pb = q.b2; }
        asm volatile("" : "+v"(a2), "+v"(b2), "+v"(pa), "+v"(pb));
        { const LeafProd q = leaf_products<LEAF0 + 7>(s1, la16, la4, lb16, lb4, R); a2 = x3(x3(a2, pa, q.a0), q.a1, q.a2); b2 = x3(x3(b2, pb, q.b0), q.b1, q.b2); }
        asm volatile("" : "+v"(a2), "+v"(b2)); // y2b
        s0 = sel_xor(s0, s1);
        { const LeafProd q = leaf_products<LEAF0 + 8>(s0, la16, la4, lb16, lb4, R); a3 = x3(x3(a3, q.a0, q.a1), q.a2, a2) ^ a1; b3 = x3(x3(b3, q.b0, q.b1), q.b2, b2) ^ b1; }
        a2 ^= a0; b2 ^= b0; // y2' = y2b ^ y0'
        asm volatile("" : "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
    } else {
        constexpr int H = N / 2, L3 = pow3(H);
#pragma unroll
        for (int i = 0; i < H; ++i) { ya[YO + H + i] ^= ya[YO + i]; yb[YO + H + i] ^= yb[YO + i]; }
        acc_conv2<H, ZO, YO, LEAF0>(v, ya, yb, la16, la4, lb16, lb4, R);
        acc_conv2<H, ZO + H, YO, LEAF0 + L3>(v, ya, yb, la16, la4, lb16, lb4, R);
#pragma unroll
        for (int i = 0; i < H; ++i) v[sbase(N) + i] = v[ZO + i] ^ v[ZO + H + i];
        acc_conv2<H, sbase(N), YO + H, LEAF0 + 2 * L3>(v, ya, yb, la16, la4, lb16, lb4, R);
#pragma unroll
        for (int i = 0; i < H; ++i) { ya[YO + H + i] ^= ya[YO + i]; yb[YO + H + i] ^= yb[YO + i]; }
    }
}

// one 16 x 16 block pair: issues the first leaf, walks the tree
__device__ __forceinline__ void conv_block2(unsigned (&v)[2 * KN - 1], unsigned (&ya)[KN], unsigned (&yb)[KN], unsigned la16, unsigned la4,
                                            unsigned lb16, unsigned lb4)
{
    LeafRegs R;
    leaf_issue<0, 0>(R, la16, la4, lb16, lb4);
    acc_conv2<KN, 0, 0, 0>(v, ya, yb, la16, la4, lb16, lb4, R);
}

// LDS of one encoder workgroup (256 threads): carved out of `ldsraw` (16-byte aligned, ENC128_LDS_BYTES)
constexpr int ENC128_LDS_BYTES = 8 * KLEAVES * 16 + 8 * KLEAVES * 4 + 128 * 16 + 128 * 4 + 33 * 64 * 4;

// one workgroup of the encoder: unit `bx` = (frame list entry bx >> 1, half bx & 1).  A device function so that the fused Rx
// launch (decim_mfma.hip: rx_fused_kernel) can run encoder workgroups beside the decimator's in ONE launch.
__device__ __forceinline__ void gf_encode128_wg(const Enc128Args &a, int bx, unsigned char *ldsraw)
{
    uint4_t *lt16 = reinterpret_cast<uint4_t *>(ldsraw);                                   // {Ta, Tb} of the leaves of G_0..G_7
    unsigned *lt4 = reinterpret_cast<unsigned *>(ldsraw + 8 * KLEAVES * 16);                // {Tc}
    uint4_t *rt16 = reinterpret_cast<uint4_t *>(ldsraw + 8 * KLEAVES * 20);                 // tables of the row constants r < 128
    unsigned *rt4 = reinterpret_cast<unsigned *>(ldsraw + 8 * KLEAVES * 20 + 128 * 16);
    unsigned (*ysum)[64] = reinterpret_cast<unsigned (*)[64]>(ldsraw + 8 * KLEAVES * 20 + 128 * 20); // reduced convolution (32 rows) + parity (row 32)
    const int tid = threadIdx.x;
    {
        // (32-byte table records: whole 16-byte loads, ALL of a thread's loads before its first LDS write -- written as a copy loop
        // hipcc made every iteration load, wait, write: three global round trips one after the other)
        static_assert(GF_NT == 256 && 8 * KLEAVES <= 3 * GF_NT, "three records per thread");
        uint4_t l16[3], r16;
        unsigned l4[3], r4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + k * GF_NT;
            const uint4_t *src = reinterpret_cast<const uint4_t *>(a.leaf_tables) + (size_t)(i < 8 * KLEAVES ? i : 0) * 2;
            l16[k] = src[0];
            l4[k] = reinterpret_cast<const unsigned *>(src + 1)[0];
        }
        {
            const uint4_t *src = reinterpret_cast<const uint4_t *>(a.tab) + (size_t)(tid & 127) * 2;
            r16 = src[0];
            r4 = reinterpret_cast<const unsigned *>(src + 1)[0];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + k * GF_NT;
            if (i < 8 * KLEAVES) { lt16[i] = l16[k]; lt4[i] = l4[k]; }
        }
        if (tid < 128) { rt16[tid] = r16; rt4[tid] = r4; }
    }
    for (int i = tid; i < 33 * 64; i += GF_NT) (&ysum[0][0])[i] = 0;
    __syncthreads();

    const int w = tid >> 6, lane = tid & 63;  // wave w owns column blocks 2w, 2w + 1
    const int fi = bx >> 1;
    const int fr = a.gen_done > 0 ? (fi / a.gen_done) * a.gen_cap + fi % a.gen_done : (a.frame_list ? a.frame_list[fi] : fi);
    const int col = (bx & 1) * 64 + lane;
    const bool live = fr >= 0 && fr < a.nframes && col < 127;
    const size_t frc = live ? (size_t)fr : 0;
    const unsigned *src = reinterpret_cast<const unsigned *>(a.in + frc * a.in_frame_bytes) + 1 + (live ? col : 0);
    unsigned *dst = reinterpret_cast<unsigned *>(a.out + frc * a.out_frame_bytes) + 1 + (live ? col : 0);
    // block 0 of the frame (header + meta block): from memory, or (K2 runs in this very launch) derived like K2 derives it
    unsigned hdr0 = 0u, blk0 = 0u;
    bool own0 = false;
    if (a.meta_count > 0 && a.gen_done > 0) {
        const int f = fr % a.gen_cap, fi = f - a.meta_first;
        if (fr >= 0 && fi >= 0 && fi < a.meta_count) { // (workgroup-uniform: the shuffle inside frame_meta_words sees whole waves)
            unsigned w[6];
            frame_meta_words(a.meta_w, a.meta_idx0, a.meta_rate, fi, w);
            own0 = true;
            hdr0 = (a.meta_frame_count0 + (unsigned)fi) & 0xffffu;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (col == k) blk0 = w[k];
        }
    }
    if (!own0) hdr0 = (live && col == 0) ? src[-1] : 0u;
    // fused framing: blocks 1..127 of this frame come straight from the decimated stream (127 samples each)
    bool fused = false;
    const unsigned *bsrc = src; // block b at bsrc[b * bstride]
    int bstride = 128;
    if (a.lin && live) {
        const int s = fr / a.lin_cap, f = fr - s * a.lin_cap;
        if (f >= a.lin_first) {
            fused = true;
            bsrc = a.lin + (size_t)s * a.lin_stride + ((size_t)f * 16129u - (size_t)a.lin_pending) + col - 127;
            bstride = 127;
        }
    }
    unsigned *fdst = const_cast<unsigned *>(src);
    // the frame that was open when the call began: its first lin_pending samples are in the frame area, the rest comes now
    bool strad = false;
    const unsigned *linp = nullptr;
    if (a.lin && live && a.lin_straddle && !fused) {
        const int s = fr / a.lin_cap, f = fr - s * a.lin_cap;
        if (f == 0) { strad = true; linp = a.lin + (size_t)s * a.lin_stride; }
    }

    const int npairs = (a.rows + 31) / 32; // pairs of 16-row tiles
#pragma unroll 1
    for (int tp = 0; tp < npairs; ++tp) {
        unsigned y0[KN], y1[KN];
#pragma unroll
        for (int i = 0; i < KN; ++i) { y0[i] = 0; y1[i] = 0; }
        unsigned p = 0;
#pragma unroll 1
        for (int q = 0; q < 2; ++q) {
            const int cb = 2 * w + q;
            unsigned v[2 * KN - 1];
            if (live && strad) {
#pragma unroll
                for (int i = 0; i < KN; ++i) {
                    const int b = KN * cb + i;
                    if (b == 0) { v[i] = src[0]; continue; }
                    const int wi = (b - 1) * 127 + col;
                    if (wi < a.lin_pending) {
                        v[i] = src[(size_t)b * 128];
                    } else {
                        v[i] = linp[wi - a.lin_pending];
                        if (tp == 0) fdst[(size_t)b * 128] = v[i];
                    }
                }
            } else if (live) {
                // (block 0 = the meta block is never in the stream)
                v[0] = cb == 0 ? (own0 ? blk0 : src[0]) : bsrc[(size_t)(KN * cb) * bstride];
#pragma unroll
                for (int i = 1; i < KN; ++i) v[i] = bsrc[(size_t)(KN * cb + i) * bstride];
                if (fused && tp == 0) {
#pragma unroll
                    for (int i = 0; i < KN; ++i) fdst[(size_t)(KN * cb + i) * 128] = v[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < KN; ++i) v[i] = 0u;
            }
#pragma unroll
            for (int i = 0; i < KN; i += 2) p = x3(p, v[i], v[i + 1]);
            const int b0 = (2 * tp) ^ cb, b1 = (2 * tp + 1) ^ cb;
            conv_block2(v, y0, y1, lds_addr(lt16 + b0 * KLEAVES), lds_addr(lt4 + b0 * KLEAVES), lds_addr(lt16 + b1 * KLEAVES), lds_addr(lt4 + b1 * KLEAVES));
        }
        if (tp == 0) __hip_atomic_fetch_xor(&ysum[32][lane], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int i = 0; i < KN; ++i) {
            __hip_atomic_fetch_xor(&ysum[i][lane], y0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_xor(&ysum[KN + i][lane], y1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        // rows 8 w .. 8 w + 7 of the 32-row pair: recovery_r = P ^ r * c_r
        const unsigned P = ysum[32][lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int rl = 8 * w + q, r = 32 * tp + rl;
            if (r < a.rows && live) {
                dst[(size_t)r * 128] = P ^ kmul(ysum[rl][lane], rt16[r], rt4[r]);
                // header {frameIndex (of the frame's block 0), 128 + r, filler 0}, UDPSinkFEC.cpp:239-243
                if (col == 0) dst[(size_t)r * 128 - 1] = (hdr0 & 0xffffu) | ((unsigned)(128 + r) << 16);
            }
        }
        if (tp + 1 < npairs) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) ysum[8 * w + q][lane] = 0;
            __syncthreads();
        }
    }
}

