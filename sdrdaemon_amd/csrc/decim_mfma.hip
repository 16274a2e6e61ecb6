// decim_mfma.hip -- centred half-band decimator cascades on the gfx950 matrix cores (MI355X).
//
// Same arithmetic as decim_kernels.hip (Decimators::decimate{2..64}_cen, Decimators.cpp:94-1305, over
// IntHalfbandFilterEO1/DB<64>::myDecimate, IntHalfbandFilterEO1.h:100-147 / IntHalfbandFilterDB.h:79-107),
// bit for bit, but the 32-tap polyphase FIR + centre tap of every stage is an exact integer matrix product
// on v_mfma_i32_16x16x64_i8 instead of 34 VALU lane-ops per output (DESIGN.md "K1m"):
//
//  * a wave owns 16 columns = 8 consecutive spans of one stream x {I, Q}; all columns advance in lockstep,
//    32 raw samples (16 first-stage outputs) per step.  D = A x B: the rows of A are the 16 output times of a
//    tile, its 64 K slots hold the taps; B holds, per column, the last four 16-entry blocks of the stage's odd
//    (FIR) or even (centre tap) input plane, one signed byte ("limb") of every entry.  Lane (column n, kq) of B
//    holds entries 8 * (t >> 1) + 2 * kq + (t & 1) (byte t) of each block = exactly the values that lane
//    (n, q = kq) of D produced two tiles of the previous stage ago: the FIR data path runs in registers, no
//    cross-lane traffic except the I / Q pairing of the raw loads and of the final outputs.
//  * exactness: int16 input x = lo + 256 hi + 128 with signed bytes lo = (x & 255) ^ 128, hi = x >> 8 (the 128
//    becomes a constant in the accumulator); stage outputs |v| <= 2^18 are split as v = b0 + 256 b1 + 65536 b2
//    - 229248 with signed bytes b0, b1 = bytes 0, 1 of ((acc >>> 13) ^ 0x8080), b2 = bits 16-18 with bit 18 flipped (0..7):
//    the 19-bit two's complement field with its sign bit flipped is v + 2^18, excess-128 bytes are signed bytes + 128,
//    and the constant goes into the next stage's accumulator start; taps h = h0 + 256 h1.  The limb products of
//    equal weight share an accumulator (|sum| < 2^21: no overflow), the four accumulators are recombined with
//    shifts modulo 2^32 = the reference's wrap-around int32 sum.
//  * the centre tap (x[2k - 30] << 13) rides in the same MFMAs: a 16-row tile needs 47 of the 64 entries of its
//    window (three blocks), the fourth dword of the window -- the one the NEXT odd block will overwrite -- holds the
//    16 even-plane entries k - 15 of the tile's rows, and the h1 limb matrix has the entry 32 (32 * 256 = 2^13)
//    where row r meets even entry r - 15.  A lane's even entries come out of the same registers as its odd ones
//    (outputs 4q, 4q+2 of the previous stage's tiles / even raw samples of its own load), one block later; the only
//    irregularity is that the window of rows 16 I .. 16 I + 15 ends with even entry 16 I, the FIRST entry of the next
//    block: lanes kq = 0 replace byte 0 (the unneeded entry 16 (I - 1)) with it.  No LDS, no cross-lane traffic.
//  * the newest block of a window replaces the oldest in place (dword `phase` of the fragment), the tap
//    matrices exist in the four rotations; the multi-rate schedule (stage s runs every 2^s steps) is unrolled
//    over one period of 4 * 2^(NS-1) steps so that every phase is a compile-time constant.
//  * a span is preceded by one period (64 * 2^L raw samples) of warm-up with stores suppressed, exactly like
//    the VALU kernel's segments.  Head [0, head) and tail of every stream go through the VALU code
//    (decim_body.h) in the same launch: it owns the bank state (load at the start, store at the end).
#include "decim_body.h"

#include <type_traits>
#include <utility>

#ifndef MF_WAVES
#define MF_WAVES 2
#endif
#ifndef MF_DMA_BURST
#define MF_DMA_BURST 1 // LDS-DMA ring: DMAs issued per step (1, 2, 4 or 8), see mf_loop_dma
#endif

namespace sdrhip {
namespace {

// slot (lane kq, byte t) of a 16-entry block <-> entry index inside the block
__host__ __device__ constexpr int mf_entry(int kq, int t) { return 8 * (t >> 1) + 2 * kq + (t & 1); }

struct MfATab {
    unsigned w[2][4][64][4]; // tap limb (h0, h1) x phase x lane x dword
};

constexpr MfATab mf_make_atab()
{
    MfATab T{};
    for (int m = 0; m < 2; ++m)
        for (int ph = 0; ph < 4; ++ph)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, kq = lane >> 4;
                for (int j = 0; j < 4; ++j) {
                    const int beta = (ph - j) & 3; // dword j holds the block `beta` blocks before the newest
                    unsigned word = 0;
                    for (int t = 0; t < 4; ++t) {
                        const int d = r - mf_entry(kq, t) + 16 * beta; // delay of the entry w.r.t. output r, in plane entries
                        int v = 0;
                        if (beta == 3) {
                            // the dword the next odd block will replace: even-plane entries 16 (I - 1) + e, except that
                            // slot (kq 0, t 0) carries entry 16 I; row r takes entry 16 I + r - 15 times 2^13 = 32 * 256
                            const int e = mf_entry(kq, t);
                            const bool hit = (kq == 0 && t == 0) ? r == 15 : e == r + 1;
                            v = (m == 1 && hit) ? 32 : 0;
                        } else if (d >= 0 && d <= 31) {
                            const int h = H32(d);
                            const int h1 = (h + 128) >> 8, h0 = h - 256 * h1;
                            v = m == 0 ? h0 : h1;
                        }
                        word |= (unsigned)(v & 0xff) << (8 * t);
                    }
                    T.w[m][ph][lane][j] = word;
                }
            }
    return T;
}

__device__ const MfATab mf_atab = mf_make_atab();

constexpr int mf_tap_sum()
{
    int s = 0;
    for (int i = 0; i < 32; ++i) s += H32(i);
    return s;
}

template <class F, int... Is> __device__ __forceinline__ void mf_static_for_impl(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void mf_static_for(F &&f)
{
    mf_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int4_t mfma(int4_t a, int4_t b, int4_t c)
{
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

template <int NS> struct MfState {
    int4_t O[NS][3];        // window of every stage, one signed byte per entry and limb: three odd blocks + the even entries
    unsigned pend[NS][2];   // odd inputs of stage s >= 1: first half of the block being formed (limbs 0-1, limb 2)
    unsigned evp[NS][2];    // even inputs, likewise
};

// value of a stage output in terms of its limbs: v = b0 + 256 b1 + 65536 b2 + MF_LIMB_BIAS
constexpr int MF_LIMB_BIAS = 128 + 32768 - 262144;

struct MfConst {
    int4_t A[2][4]; // tap matrices (limbs h0, h1) in the four rotations
    int4_t c0;      // accumulator start of stage 0: 128 * (sum of the FIR taps + 2^13) + (bias << 13), in all four rows
    int cinN;       // other stages: MF_LIMB_BIAS * (sum of the FIR taps + 2^13) + (bias << 13)
    unsigned selA;  // v_perm selectors {new, old}: lanes kq = 0 take byte 0 (selA) / byte 2 (selB) of `new` into byte 0,
    unsigned selB;  // the other lanes keep `old`
};

// FR = false: stream-order output; FR = true: UDPSinkFEC::write's frame layout (UDPSinkFEC.cpp:134-155), DecimArgs::frame_mode
template <bool FR> struct MfOut {
    unsigned *p;          // stream order: where this lane's next two outputs go
    unsigned *dump;       // 16 bytes per lane that swallow the stores of the warm-up period (no branch in the loop body)
    int store;            // 0 during warm-up
    int norm, trunk;
    unsigned sel_pack;    // v_perm selector {own, received} -> {I, Q} halves of an output dword
    // frame layout: the lane's next two samples sit at byte `off` (+ 4) of the stream's frame area unless they lie beyond the end
    // of super block 1 + b (column c of the lane pair's first sample >= t0 / t1): then `extra` bytes further
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned off, c, b, t0, t1;
    unsigned gap;         // bytes from the last sample of a frame to the first one of the next, minus 4
};

typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sbfe16(unsigned v, int off) { return (int)__builtin_amdgcn_sbfe((int)v, (unsigned)off, 16u); }

// (a << sh) + b = one v_lshl_add_u32.  Plain C on purpose: the operands come straight out of MFMAs and go into MFMAs,
// and hipcc pads those hazards only for instructions it emits itself, not for inline asm.  The limb accumulators are
// recombined in Horner form ((g3 << 8) + g2) << 8 ... so that every step is exactly one shift-add (a flat sum of
// shifted terms compiles to v_lshlrev + v_add3 pairs: one op more per output).
__device__ __forceinline__ unsigned lshl_add(unsigned a, int sh, unsigned b) { return (a << sh) + b; }
// keeps hipcc from re-associating a Horner chain back into a flat sum: the partial result (a VALU result, never a raw
// MFMA output) becomes opaque; the statement holds no instruction
__device__ __forceinline__ unsigned opaque(unsigned v)
{
    asm("" : "+v"(v));
    return v;
}

// decimate32 / 64: the windows of the last one / two stages are shifted with moves (six per tile, of stages that run every
// 16 and 32 steps) instead of rotating in place: their phase then is a constant and the unrolled schedule repeats after
// 32 steps like decimate16's, a half / a quarter of the code (decimate64: ~20 KB instead of ~80 KB, which did not fit the
// instruction cache).
__host__ __device__ constexpr int mf_nfixed(int ns) { return ns > 4 ? ns - 4 : 0; }
__host__ __device__ constexpr bool mf_fixed(int ns, int s) { return s >= ns - mf_nfixed(ns); } // stage s of ns has a fixed window
template <int NS> constexpr int mf_period() { return 4 << (NS - 1 - mf_nfixed(NS)); } // first-stage steps per period

template <int NS, int S, int I, class OC> __device__ __forceinline__ void mf_stage(MfState<NS> &st, const MfConst &k, OC &oc, int comp)
{
    // (the last stages of the long cascades keep their windows in fixed dwords, see mf_nfixed)
    constexpr int PH = mf_fixed(NS, S) ? 0 : (I & 3);
    const int4_t Ah0 = k.A[0][PH], Ah1 = k.A[1][PH];
    const int4_t z = {0, 0, 0, 0};
    unsigned acc[4];
    if constexpr (S == 0) {
        int4_t g0 = mfma(Ah0, st.O[0][0], k.c0);
        int4_t g1 = mfma(Ah0, st.O[0][1], z);
        int4_t g2 = mfma(Ah1, st.O[0][1], z);
        g1 = mfma(Ah1, st.O[0][0], g1);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = lshl_add(opaque(lshl_add((unsigned)g2[r], 8, (unsigned)g1[r])), 8, (unsigned)g0[r]);
    } else {
        const int4_t cN = {k.cinN, k.cinN, k.cinN, k.cinN};
        int4_t g0 = mfma(Ah0, st.O[S][0], cN);
        int4_t g1 = mfma(Ah0, st.O[S][1], z);
        int4_t g2 = mfma(Ah0, st.O[S][2], z);
        int4_t g3 = mfma(Ah1, st.O[S][2], z);
        g1 = mfma(Ah1, st.O[S][0], g1);
        g2 = mfma(Ah1, st.O[S][1], g2);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[r] = lshl_add(opaque(lshl_add(opaque(lshl_add((unsigned)g3[r], 8, (unsigned)g2[r])), 8, (unsigned)g1[r])), 8, (unsigned)g0[r]);
    }

    if constexpr (S < NS - 1) {
        // outputs 4q .. 4q+3 of this tile as 19-bit fields u = acc >>> 13 (limbs: bytes 0, 1 and bits 16-18).  r = 1, 3
        // are odd inputs of stage S+1 (entries 8 I + 2q, + 1 of its odd plane), r = 0, 2 the same entries of its even plane
        constexpr int SIG = I & 1, J = I >> 1;
        const unsigned X = 0x80808080u, X2 = 0x04040404u;
        const unsigned u0 = acc[0] >> 13, u1 = acc[1] >> 13, u2 = acc[2] >> 13, u3 = acc[3] >> 13;
        const unsigned po = perm(u3, u1, 0x05010400u), po2 = perm(u3, u1, 0x0c0c0602u);
        const unsigned pe = perm(u2, u0, 0x05010400u), pe2 = perm(u2, u0, 0x0c0c0602u);
        if constexpr (SIG == 0) {
            st.pend[S + 1][0] = po; st.pend[S + 1][1] = po2;
            // this tile's first even output (lanes q = 0) is entry 16 J of the even plane: the last one tile J of the
            // next stage needs; the other 15 are in place since the previous block was completed (below)
            constexpr int DJ = mf_fixed(NS, S + 1) ? 1 : ((J + 1) & 3);
            st.O[S + 1][0][DJ] = (int)(perm(pe, (unsigned)st.O[S + 1][0][DJ], k.selA) ^ X);
            st.O[S + 1][1][DJ] = (int)(perm(pe, (unsigned)st.O[S + 1][1][DJ], k.selB) ^ X);
            st.O[S + 1][2][DJ] = (int)(perm(pe2, (unsigned)st.O[S + 1][2][DJ], k.selA) ^ X2);
            st.evp[S + 1][0] = pe; st.evp[S + 1][1] = pe2;
        } else {
            constexpr bool FIX = mf_fixed(NS, S + 1);
            constexpr int NJ = FIX ? 0 : (J & 3), DN = FIX ? 1 : ((J + 2) & 3);
            if constexpr (FIX) {
                // phase 0 for ever: dword 0 = newest block, 3 = the previous one, 2 = the one before, 1 = the even entries
#pragma unroll
                for (int l = 0; l < 3; ++l) { st.O[S + 1][l][2] = st.O[S + 1][l][3]; st.O[S + 1][l][3] = st.O[S + 1][l][0]; }
            }
            st.O[S + 1][0][NJ] = (int)(perm(po, st.pend[S + 1][0], 0x05040100u) ^ X);
            st.O[S + 1][1][NJ] = (int)(perm(po, st.pend[S + 1][0], 0x07060302u) ^ X);
            st.O[S + 1][2][NJ] = (int)(perm(po2, st.pend[S + 1][1], 0x05040100u) ^ X2);
            // even block J (its limbs not yet made signed: that happens when the entry of lanes kq = 0 is merged in)
            const unsigned e0 = perm(pe, st.evp[S + 1][0], 0x05040100u), e1 = perm(pe, st.evp[S + 1][0], 0x07060302u),
                           e2 = perm(pe2, st.evp[S + 1][1], 0x05040100u);
            mf_stage<NS, S + 1, J>(st, k, oc, comp);
            // tile J was the last reader of odd block J - 2: its dword takes the even entries of tile J + 1
            st.O[S + 1][0][DN] = (int)e0; st.O[S + 1][1][DN] = (int)e1; st.O[S + 1][2][DN] = (int)e2;
        }
    } else {
        // Lanes n = 2p (I) and 2p + 1 (Q) hold the two components of the same four outputs.  The I lane finishes outputs 0, 1, the Q
        // lane outputs 2, 3: each hands the other the two values it needs (one DPP move each), shifts its own two and the received
        // two, and one v_perm with a lane-constant selector makes the {I, Q} dword.  (Until round 5 both lanes packed and stored all
        // four dwords: twice the shifts and selects, 40 instructions per tile instead of 20.)  No divergence, no branch: the loop
        // body stays one basic block and hipcc's vmcnt waits stay exact.
        const int o0 = (int)acc[0] >> 13, o1 = (int)acc[1] >> 13, o2 = (int)acc[2] >> 13, o3 = (int)acc[3] >> 13;
        const int g0 = comp ? o0 : o2, g1 = comp ? o1 : o3; // for the neighbour
        const int m0 = comp ? o2 : o0, m1 = comp ? o3 : o1; // mine
        const int r0 = __builtin_amdgcn_update_dpp(0, g0, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
        const int r1 = __builtin_amdgcn_update_dpp(0, g1, 0xB1, 0xf, 0xf, true);
        // `x << norm_shift >> trunk_shift` then FixReal truncation, Decimators.cpp:112-113 (final_pack)
        const unsigned pk0 = perm((unsigned)((int)((unsigned)r0 << oc.norm) >> oc.trunk), (unsigned)((int)((unsigned)m0 << oc.norm) >> oc.trunk), oc.sel_pack);
        const unsigned pk1 = perm((unsigned)((int)((unsigned)r1 << oc.norm) >> oc.trunk), (unsigned)((int)((unsigned)m1 << oc.norm) >> oc.trunk), oc.sel_pack);
        if constexpr (std::is_same<OC, MfOut<true>>::value) {
            // frame layout: 127 samples per super block behind a 4-byte header, 127 blocks per frame behind the meta block.  A
            // sample beyond the end of the block lies `extra` bytes further (the next block's header; from the last block on: the
            // recovery blocks, the next frame's meta block and header).  Warm-up: `bias` puts the offsets beyond the descriptor's
            // range, the hardware drops the stores.
            const unsigned extra = oc.b == 126u ? oc.gap : 4u;
            const unsigned bias = oc.store ? 0u : 0x80000000u;
            // The lane's two samples go out as ONE 8-byte store (lane pairs then write 16 contiguous bytes, a tile 64 per span)
            // unless the block ends between them (column c == t1, once in 127 samples): then as two 4-byte stores.  All three stores
            // are always issued -- no branch -- with bit 30 of the offset set on the ones that do not apply: beyond the descriptor's
            // range (1 GiB) like the warm-up's, the hardware drops them.  (First version: two 4-byte stores per lane -- every store
            // instruction wrote every other dword of its 64-byte pieces: the launch took 243 us instead of 224, profiles/r05_rx_direct.txt.)
            const unsigned a0 = oc.off + (oc.c >= oc.t0 ? extra : 0u) + bias;
            const unsigned hi = a0 ^ 0x40000000u;
            const bool split = oc.c == oc.t1;
            const unsigned a8 = split ? hi : a0, a4 = split ? a0 : hi;
            __builtin_amdgcn_raw_buffer_store_b64((uint2_t){pk0, pk1}, oc.rsrc, a8, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(pk0, oc.rsrc, a4, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(pk1, oc.rsrc, a4 + extra, 4, 0);
            // the lane pair's next four outputs are 16 samples on
            const unsigned inc = oc.store ? 16u : 0u;
            oc.c += inc;
            const bool wrap = oc.c >= 127u;
            oc.off += 4u * inc + (wrap ? extra : 0u);
            oc.c -= wrap ? 127u : 0u;
            oc.b += wrap ? 1u : 0u;
            oc.b = oc.b >= 127u ? 0u : oc.b;
        } else {
            // (Stored as they come a tile is 64 bytes per span.  On the memory skeleton of this kernel those 67 MB of
            // scattered writes cost as much as 250 MB of reads although the same stores ALONE take 0.01 ms (tools/dma_probe.hip): it
            // is the mix of the write stream with 5 TB/s of reads, not the granularity -- pairing two tiles into one 128-byte store
            // changed nothing: tools/experiments_r03/decim_mfma_experiments.patch, MF_PAIR.)
            unsigned *dst = oc.store ? oc.p : oc.dump;
            *reinterpret_cast<uint2_t *>(dst) = (uint2_t){pk0, pk1};
            oc.p += oc.store ? 16 : 0;
        }
    }
}

// front end of a step: the lane's 16 raw bytes (four samples x, y, z, w: y, w are odd-plane entries, x, z even-plane entries
// 2q, 2q + 1 of their half block) -> limb bytes of the stage-0 window
struct MfFront {
    unsigned sel_own, sel_oth, sel_lo, sel_hi;
    unsigned ev_lo, ev_hi; // even entries of the previous step (limbs made signed)
};
template <int NS, int I> __device__ __forceinline__ void mf_front(MfState<NS> &st, const MfConst &k, MfFront &f, const uint4_t r)
{
    constexpr int i = I;
    const unsigned ao = perm(r.w, r.y, f.sel_own), ae = perm(r.z, r.x, f.sel_own);
    const unsigned xo = perm(r.w, r.y, f.sel_oth), xe = perm(r.z, r.x, f.sel_oth);
    const unsigned yo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xo, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
    const unsigned ye = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xe, 0xB1, 0xf, 0xf, true);
    st.O[0][0][i & 3] = (int)(perm(ao, yo, f.sel_lo) ^ 0x80808080u);
    st.O[0][1][i & 3] = (int)perm(ao, yo, f.sel_hi);
    const unsigned cur_lo = perm(ae, ye, f.sel_lo) ^ 0x80808080u, cur_hi = perm(ae, ye, f.sel_hi);
    // even entries of tile i: those of the previous step, and the first one of this step (lanes kq = 0)
    st.O[0][0][(i + 1) & 3] = (int)perm(cur_lo, f.ev_lo, k.selA);
    st.O[0][1][(i + 1) & 3] = (int)perm(cur_hi, f.ev_hi, k.selA);
    f.ev_lo = cur_lo; f.ev_hi = cur_hi;
}

// The same front end fed from LDS: both lanes of an I / Q pair read BOTH 16-byte pieces of their column's step (first half rA,
// second half rB; the same addresses: an LDS broadcast) and pick their own component -- no exchange with the neighbour (2 DPP moves
// and 2 v_perm less per step: VALU issue is what bounds the kernel; the second ds_read_b128 issues on the LDS port).
template <int NS, int I> __device__ __forceinline__ void mf_front2(MfState<NS> &st, const MfConst &k, MfFront &f, const uint4_t rA, const uint4_t rB)
{
    constexpr int i = I;
    const unsigned aoA = perm(rA.w, rA.y, f.sel_own), aoB = perm(rB.w, rB.y, f.sel_own);
    const unsigned aeA = perm(rA.z, rA.x, f.sel_own), aeB = perm(rB.z, rB.x, f.sel_own);
    st.O[0][0][i & 3] = (int)(perm(aoA, aoB, 0x01000504u) ^ 0x80808080u);
    st.O[0][1][i & 3] = (int)perm(aoA, aoB, 0x03020706u);
    const unsigned cur_lo = perm(aeA, aeB, 0x01000504u) ^ 0x80808080u, cur_hi = perm(aeA, aeB, 0x03020706u);
    st.O[0][0][(i + 1) & 3] = (int)perm(cur_lo, f.ev_lo, k.selA);
    st.O[0][1][(i + 1) & 3] = (int)perm(cur_hi, f.ev_hi, k.selA);
    f.ev_lo = cur_lo; f.ev_hi = cur_hi;
}

// ---- LDS-DMA input ring (round 3).  Measured on the register ring above (8 x 2^25, decimate16): arithmetic alone 0.201 ms, the
// memory skeleton alone 0.206-0.238 ms, both together 0.240 ms: with ONE wave per SIMD nothing covers a wave's s_waitcnt (33 %
// of its cycles), and a second wave per SIMD does not help because the SIMD's issue port is what the arithmetic saturates.  So
// the loads leave the wave's instruction stream's critical path altogether: global_load_lds_dwordx4 copies 1 KiB per
// instruction straight into LDS (no VGPRs, nothing for hipcc to drain at the loop top), 24 steps = 24 KiB per wave ahead.
//  * a DMA covers ONE span for 8 steps (1 KiB contiguous: eight full 128-byte lines touched once, against 32 line visits for
//    the span-strided register loads); a group = 8 steps = 8 DMAs (one per span), the ring holds 4 groups.
//  * step i issues DMA (group i / 8 + 3, span i % 8); the group needed next is waited for once per 8 steps with
//    s_waitcnt vmcnt(15): fifteen DMAs were issued after its last one (loads return in order; stores in between only make
//    the wait conservative).
//  * a lane reads its 16 bytes of a step back with one ds_read_b128 (issued one step ahead).  The 1-KiB block of span p
//    starts 72 p + 2 (p >> 2) sixteen-byte units into the group: the sixteen lanes of every ds_read_b128 service group
//    (MI355X_MICROARCH.md, LDS table) then hit sixteen distinct 4-bank columns.
constexpr int MF_GROUP_BYTES = 9216;             // 8 blocks of 1 KiB + their skew
// Ring depth NG (groups per wave): 4 = 36 KiB per wave, 147 KiB per workgroup -- the workgroup owns its CU; 3 = 27 KiB per wave,
// 108 KiB per workgroup, which leaves 52 KiB of the CU's 160 KiB for two workgroups of ANOTHER kernel (the CM256 encoder of the
// previous call on a second stream: sdrhip_pipes.cpp, "overlap" mode).  The DMAs run NG - 1 groups (8 (NG - 1) steps) ahead.
constexpr int mf_wave_ring(int ng) { return ng * MF_GROUP_BYTES; } // bytes of LDS per wave
// decimate16 only: measured with 3 interleaved rounds (tools/experiments_r03/exp24.sh), register ring against LDS-DMA ring: decimate16 0.2445 / 0.2363 ms,
// decimate32 0.2512 / 0.2537, decimate64 0.2693 / 0.2947 (their warm-up and the ring's run-ahead past the span grow with the ratio)
__host__ __device__ constexpr bool mf_dma_applies(int ns) { return ns == 4; } // (period of 32 steps; one workgroup per CU)
__host__ __device__ constexpr int mf_block_units(int p) { return 72 * p + 2 * (p >> 2); }

template <int D> __device__ __forceinline__ void mf_dma_issue(unsigned slot, unsigned voff, unsigned long long span_base)
{
    constexpr int off = 16 * mf_block_units(D);
    // M0 = LDS byte address of the block (wave-uniform); the lanes' 16 bytes land at M0 + 16 * lane
    asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3 nt" ::"v"(voff), "s"(slot), "n"(off), "s"(span_base) : "memory", "scc");
}
template <int N> __device__ __forceinline__ void mf_wait_vm()
{
    static_assert(N == 0 || N == 7 || N == 8 || N == 15 || N == 16, "vmcnt immediates of the ring");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

#ifdef MF_STAMPS
// timeline experiment (tools/experiments_r06/k1m_timeline.py): lane 0 of every matrix-core wave leaves s_memrealtime (100 MHz) at eight
// points -- 0 start, 1 constants + ring issued, 2 first group landed, 3 .. 6 = 1, 25, 50, 75 % of the periods done, 7 end -- and HW_ID
__device__ unsigned long long g_mf_stamps[4096 * 10];
__device__ __forceinline__ void mf_stamp(int gw, int k)
{
    if ((threadIdx.x & 63) == 0 && gw < 4096) {
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        g_mf_stamps[gw * 10 + k] = t;
        if (k == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_mf_stamps[gw * 10 + 8] = ((unsigned long long)(xcc & 0xfu) << 32) | hw;
        }
    }
}
#define MF_STAMP(gw, k) mf_stamp(gw, k)
#else
#define MF_STAMP(gw, k) do { } while (0)
#endif

// (A loader-wave variant -- a fifth wave per workgroup issues all DMAs, the compute waves meet it at one s_barrier per group -- was
// measured slower, 0.244-0.252 ms against 0.232-0.238 ms with the DMAs in the compute waves: tools/experiments_r03/
// decim_mfma_experiments.patch, MF_LOADER.)
//
// Ring bookkeeping, for any depth NG: the period has 4 groups (32 steps); absolute group k lives in slot k % NG.  rs[j] / lr[j]
// (j < NG) are the LDS byte address of the slot that holds group j OF THE CURRENT PERIOD (as M0 base, wave-uniform; as the lane's
// read address): group G of the period (G may run into the next one) is slot rs[G % NG], and at the end of a period the arrays
// rotate by 4 % NG (NG = 4: not at all, NG = 3: by one -- three s_mov + three v_mov per 32 steps).  Step i issues the DMA of
// group i / 8 + NG - 1, span i % 8 into the slot group i / 8 - 1 has just left; the group read next is awaited once per 8 steps:
// behind its last DMA the wave has issued the NG - 3 + 1 full groups in between and 7 DMAs of the current issue group.
template <int NS, int NG, class OC>
__device__ __forceinline__ void mf_loop_dma(const DecimArgs &a, MfState<NS> &st, const MfConst &k, OC &oc, MfFront &fr, unsigned lds_addr,
                                            const char *wbase, size_t S, int nper, int WP, int lane, int p, int comp, int q, int gw)
{
    constexpr int P = mf_period<NS>();
    static_assert(P == 32, "four groups of 8 steps per period");
    static_assert(NG == 2 || NG == 3 || NG == 4, "ring depth");
    constexpr int LA = NG - 1; // groups the DMAs run ahead
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned ring = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_addr + (unsigned)wv * (unsigned)mf_wave_ring(NG)));
    unsigned long long sb[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const unsigned long long v = (unsigned long long)(wbase + (size_t)d * S * 4);
        sb[d] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    }
    unsigned rs[NG];
    // what this lane reads back: bytes 128 j + 64 comp + 16 q of step j of span p
    const __attribute__((address_space(3))) char *lr[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        rs[j] = ring + (unsigned)(j * MF_GROUP_BYTES);
        lr[j] = (const __attribute__((address_space(3))) char *)(size_t)(rs[j] + 16u * (unsigned)(mf_block_units(0) + 72 * p + 2 * (p >> 2) + 4 * comp + q));
    }
    // voff[g]: byte offset (inside a span) of the group that issue position g = i / 8 of the period loads next: absolute group
    // 4 per + g + LA, 1 KiB per group; past the wave's last group the DMAs re-read that last group -- cache hits, no HBM traffic,
    // and the vmcnt arithmetic stays as it is
    unsigned voff[4];
    const unsigned vmax = 16u * (unsigned)lane + 1024u * (unsigned)(4 * nper - 1); // offset of the wave's last group
#pragma unroll
    for (int g = 0; g < 4; ++g) voff[g] = min(16u * (unsigned)lane + 1024u * (unsigned)(g + LA), vmax);
    {
        // groups 0 .. LA - 1 of the ring
        unsigned v0 = 16u * (unsigned)lane;
        mf_static_for<8 * LA>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            mf_dma_issue<i % 8>(rs[i / 8], v0 + 1024u * (unsigned)(i / 8), sb[i % 8]);
        });
    }
    MF_STAMP(gw, 1);
    mf_wait_vm<8 * (LA - 1)>(); // group 0 has landed
    uint4_t r = *reinterpret_cast<const __attribute__((address_space(3))) uint4_t *>(lr[0]);
    MF_STAMP(gw, 2);
    for (int per = 0; per < nper; ++per) {
#ifdef MF_STAMPS
        if (per == 1) MF_STAMP(gw, 3);
        if (per == nper / 4) MF_STAMP(gw, 4);
        if (per == nper / 2) MF_STAMP(gw, 5);
        if (per == 3 * nper / 4) MF_STAMP(gw, 6);
#endif
        oc.store = per >= WP;
        mf_static_for<P>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int i1 = i + 1; // the step read ahead; i1 = 32: step 0 of the next period = group 4 of this one
            // (MF_DMA_BURST = B: the 8 DMAs of a group go out B per step in its first 8 / B steps instead of one per step -- the
            // shallow ring's worst-case lead is then 8 LA steps instead of 8 LA - 7)
            // (NG = 2, the two-waves-per-SIMD experiment: one group ahead, so the whole next group goes out at the first step of this one)
            constexpr int B = NG == 2 ? 8 : MF_DMA_BURST;
            if constexpr (i1 % 8 == 0) mf_wait_vm<(B == 1 ? 8 * (LA - 2) + 7 : 8 * (LA - 1))>(); // the next group has landed
            const uint4_t rn = *reinterpret_cast<const __attribute__((address_space(3))) uint4_t *>(lr[(i1 / 8) % NG] + 128 * (i1 % 8));
            constexpr int g = i / 8, d = i % 8;
            if constexpr (d < 8 / B) {
                mf_static_for<B>([&](auto bc) {
                    constexpr int sp = B * d + decltype(bc)::value;
                    mf_dma_issue<sp>(rs[(g + LA) % NG], voff[g], sb[sp]);
                });
                if constexpr (d == 8 / B - 1) voff[g] = min(voff[g] + 4096u, vmax); // next period's group of this issue position
            }
            mf_front<NS, i>(st, k, fr, r);
            mf_stage<NS, 0, i>(st, k, oc, comp);
            r = rn;
        });
        if constexpr (4 % NG != 0) { // the next period's group 0 is this period's group 4
            const unsigned t = rs[0];
            const __attribute__((address_space(3))) char *u = lr[0];
#pragma unroll
            for (int j = 0; j + 1 < NG; ++j) { rs[j] = rs[j + 1]; lr[j] = lr[j + 1]; }
            rs[NG - 1] = t; lr[NG - 1] = u;
        }
    }
    mf_wait_vm<0>(); // no DMA may outlive the workgroup's LDS allocation
}

template <int NS, int NG = 0, bool FR = false> __device__ __forceinline__ void mf_wave(const DecimArgs &a, int gw, unsigned lds_addr) // NG = 0: register ring
{
    // beside another kernel's waves (overlap mode) the matrix-core wave is the one the launch waits for: it issues first
    if (a.mf_prio) __builtin_amdgcn_s_setprio(3);
    constexpr int L = NS;
    constexpr int P = mf_period<NS>();  // first-stage steps per period of the unrolled schedule
#ifndef MF_DEPTH
#define MF_DEPTH 8
#endif
#ifndef MF_BURST
#define MF_BURST 1 // loads are issued for MF_BURST consecutive steps at a time (contiguous addresses per span)
#endif
    constexpr int D = MF_DEPTH < P ? MF_DEPTH : P / 2; // steps of loads in flight
    constexpr size_t W = (size_t)64 << L; // warm-up, raw samples: one, two or four periods
    constexpr int WP = (int)(W / (32 * (size_t)P));
    static_assert(P % D == 0, "prefetch ring");
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, q = lane >> 4, comp = n & 1, p = n >> 1;
    const int stream = gw / a.mf_wps, ws = gw - stream * a.mf_wps;
    const size_t S = a.mf_span;
    const size_t wave_start = a.mf_head + (size_t)ws * 8 * S; // first stored raw sample of column pair 0
    const char *wbase = reinterpret_cast<const char *>(a.in) + ((size_t)stream * a.in_stride + wave_start - W) * 4;
    // the I lane of a column pair loads raw samples 4q .. 4q+3 of a step's 32, the Q lane 16 + 4q .. 16 + 4q+3; each
    // extracts both components and hands the other one to its neighbour (one DPP move per plane)
    const unsigned loff = (unsigned)((size_t)p * S * 4) + 16u * (unsigned)q + 64u * (unsigned)comp;
    const int T = (int)((W + S) / 32); // steps
    const int nper = T / P;

    MfConst k;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) k.A[m][ph] = *reinterpret_cast<const int4_t *>(&mf_atab.w[m][ph][lane][0]);
    const unsigned b13 = (unsigned)a.bias << 13;
    const unsigned gain = (unsigned)mf_tap_sum() + 8192u; // FIR taps + centre tap
    {
        const int c = (int)(128u * gain + b13); // x = lo + 256 hi + 128 on both planes
        k.c0 = (int4_t){c, c, c, c};
    }
    k.cinN = (int)((unsigned)MF_LIMB_BIAS * gain + b13);
    k.selA = q == 0 ? 0x03020104u : 0x03020100u;
    k.selB = q == 0 ? 0x03020106u : 0x03020100u;

    MfOut<FR> oc;
    oc.store = 0;
    oc.dump = a.mf_dump + 4 * lane;
    oc.norm = a.norm; oc.trunk = a.trunk;
    oc.sel_pack = comp ? 0x01000504u : 0x05040100u;
    {
        // this lane's first output: the I lane of a pair finishes outputs 4q, 4q + 1 of a tile, the Q lane 4q + 2, 4q + 3
        const size_t first = ((wave_start + (size_t)p * S) >> L) + 4u * (unsigned)q;
        if constexpr (FR) {
            // a.out = the stream's first frame slot, a.out_stride its pitch in dwords (DecimArgs::frame_mode)
            unsigned *obase = reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride;
            oc.rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0x3fffffff, 0x00020000);
            const unsigned long long g = a.frame_sample_base + first;
            const unsigned long long f = g / 16129u;
            const unsigned w = (unsigned)(g - f * 16129u);
            oc.b = w / 127u; oc.c = w - oc.b * 127u;
            oc.off = 4u * (unsigned)(((size_t)f * (size_t)a.frame_blocks + 1u + oc.b) * 128u + 1u + oc.c + 2u * (unsigned)comp);
            oc.t0 = 127u - 2u * (unsigned)comp; oc.t1 = 126u - 2u * (unsigned)comp;
            oc.gap = 4u * ((unsigned)a.frame_blocks * 128u - 16255u);
            oc.p = nullptr;
        } else {
            unsigned *obase = reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride;
            oc.p = obase + first + 2 * comp;
            oc.off = oc.c = oc.b = oc.t0 = oc.t1 = oc.gap = 0u;
        }
    }

    MfState<NS> st;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int b = 0; b < 3; ++b) st.O[s][b] = (int4_t){0, 0, 0, 0};
        st.pend[s][0] = st.pend[s][1] = 0u;
        st.evp[s][0] = st.evp[s][1] = 0u;
    }

    // byte selectors: a sample dword is {I lo, I hi, Q lo, Q hi}
    MfFront fr;
    fr.sel_own = comp ? 0x07030602u : 0x05010400u; // {lo(a), lo(b), hi(a), hi(b)} of this lane's component
    fr.sel_oth = comp ? 0x05010400u : 0x07030602u; // ... of the neighbour's
    // {own half, received half} -> {first, second} half of the block: the I lane loaded the first half
    fr.sel_lo = comp ? 0x05040100u : 0x01000504u; fr.sel_hi = comp ? 0x07060302u : 0x03020706u;
    fr.ev_lo = 0u; fr.ev_hi = 0u;
    if constexpr (NG != 0 && mf_dma_applies(NS)) {
        mf_loop_dma<NS, NG>(a, st, k, oc, fr, lds_addr, wbase, S, nper, WP, lane, p, comp, q, gw);
        return;
    }
    uint4_t ld[D];
    // step g of this lane's column pair: 128 bytes at src + 128 g.  No bounds handling: the loads run D steps past the
    // end of the span, i.e. into the next span or (last span of a stream) the first 32 D samples of the tail that
    // plan_decimate_mfma() guarantees
    const char *src = wbase + loff;
#define MF_LDSTRIDE 128
    // (Tried: loads and s_waitcnt vmcnt(D - 1) issued by hand in asm, because hipcc, which counts outstanding VMEM
    // operations exactly only inside a basic block, drains the ring with a vmcnt(0) at the top of every period: same
    // launch time, 0.255 ms both ways, and the register allocator may copy an asm load's destination before the wait.)
#pragma unroll
    for (int d = 0; d < D; ++d) ld[d] = SDRHIP_STREAM_LOAD(reinterpret_cast<const uint4_t *>(src + MF_LDSTRIDE * d));

    for (int per = 0; per < nper; ++per) {
        oc.store = per >= WP;
        mf_static_for<P>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int slot = i % D;
            const uint4_t r = ld[slot];
            ld[slot] = SDRHIP_STREAM_LOAD(reinterpret_cast<const uint4_t *>(src + MF_LDSTRIDE * (i + D)));
            mf_front<NS, i>(st, k, fr, r);
            mf_stage<NS, 0, i>(st, k, oc, comp);
        });
        src += MF_LDSTRIDE * P;
    }
}

// grid.x = the matrix-core workgroups (four waves = four groups of 8 spans each), then nstreams * mf_npieces VALU
// workgroups (head + tail pieces of every stream)
__host__ __device__ constexpr int mf_block_threads(int) { return NT; }

template <int L, bool PACK16, int NG, bool FR> __global__ __launch_bounds__(mf_block_threads(L), mf_dma_applies(L) && NG != 2 ? 1 : MF_WAVES) void decim_mfma_kernel(DecimArgs a)
{
    // the VALU pieces' stage buffers, or (matrix-core workgroups of the long cascades) the four waves' LDS-DMA rings
    constexpr int LDSDW = mf_dma_applies(L) && mf_wave_ring(NG) > DecimLds<L, 2, PACK16>::dwords ? mf_wave_ring(NG) : DecimLds<L, 2, PACK16>::dwords; // (4 waves x ring bytes / 4)
    __shared__ __attribute__((aligned(16))) int lds[LDSDW];
    // The matrix-core workgroups come FIRST in the grid: the dispatcher deals the first workgroups of a launch across
    // the empty CUs, and with one wave per SIMD (plan_decimate_mfma) the launch takes as long as its fullest CU: a CU
    // that got two of them while the short VALU pieces held slots elsewhere doubled the time of the whole launch.
    const int nmf = (a.nstreams * a.mf_wps + 3) / 4;
    const int bx = blockIdx.x;
    if (bx >= nmf) {
        // With the LDS-DMA ring every workgroup of this kernel owns a whole CU's LDS, so a piece workgroup never sits beside a
        // matrix-core one: only as many of them as the planner left CUs free start with the launch, the others run AFTER the
        // matrix-core waves (tools/experiments_r04/k1m_stamps.py: 16 of 24 started at 272-289 us of a 278-us matrix part).  So the
        // first mf_piece_early piece workgroups (one per free CU) take mf_piece_share pieces each -- as many as fit beside the matrix
        // part -- and only what is left after that goes to workgroups of a single piece behind it.
        const int nitems = a.nstreams * a.mf_npieces, j = bx - nmf, nearly = a.mf_piece_early * a.mf_piece_share;
        const int nmine = j < a.mf_piece_early ? a.mf_piece_share : 1;
        for (int k = 0; k < nmine; ++k) {
            const int lx = j < a.mf_piece_early ? j + k * a.mf_piece_early : nearly + (j - a.mf_piece_early);
            if (lx >= nitems || (j < a.mf_piece_early && lx >= nearly)) break; // (workgroup-uniform)
            const int stream = lx / a.mf_npieces, piece = lx - stream * a.mf_npieces;
            if (piece == 0) {
                decim_piece<L, 2, PACK16>(a, lds, stream, 0, a.mf_head, true, false, piece, a.mf_npieces);
            } else {
                const size_t s0 = a.mf_tail_start + (size_t)(piece - 1) * a.mf_tail_seg;
                size_t s1 = s0 + a.mf_tail_seg;
                if (s1 > a.n_used || piece == a.mf_npieces - 1) s1 = a.n_used;
                decim_piece<L, 2, PACK16>(a, lds, stream, s0, s1, false, piece == a.mf_npieces - 1, piece, a.mf_npieces);
            }
            __syncthreads(); // (the next piece reuses the stage buffers)
        }
        return;
    }
    const int gw = __builtin_amdgcn_readfirstlane(bx * 4 + (int)(threadIdx.x >> 6));
    if (gw >= a.nstreams * a.mf_wps) return;
    MF_STAMP(gw, 0);
    mf_wave<L, NG, FR>(a, gw, (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds);
    MF_STAMP(gw, 7);
}
#ifdef MF_STAMPS
} // namespace
extern "C" int sdrhip_debug_mf_stamps(unsigned long long *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_mf_stamps), sizeof(g_mf_stamps)); }
namespace {
#endif

// ---- fused Rx launch: the decimator of THIS call and the CM256 encoder of the frames the PREVIOUS call completed, in one grid.
// The matrix-core waves run one per SIMD and leave ~40 % of their SIMD's issue slots empty (memory waits, dependent issue);
// a kernel boundary cannot fill them, co-resident encoder workgroups can: first the decimator's workgroups (as in
// decim_mfma_kernel, register ring: its 34 KB of static LDS leave room for three encoder workgroups per CU), then one
// encoder workgroup per (frame, half block).  No dependency between the two roles inside the launch (sdrhip_pipes.cpp).
#include "gf_encode128_body.h"
// Roles are claimed at run time, not by blockIdx: the decimator's workgroups must sit ONE per CU (a launch lasts as long as
// its fullest CU), but every workgroup of a kernel has the same resource footprint, and with 2 000 encoder workgroups in the
// grid the dispatcher packs the first 248 workgroups four to a CU (measured: 1.18 ms instead of 0.26).  So a workgroup first
// asks whether it is the FIRST of this launch on its CU (atomicMax of the launch's tag on a per-CU word, key from HW_ID /
// XCC_ID): the first ones take the matrix-core units, all others the VALU pieces and the encoder units (leftovers of any
// kind go to whoever comes last: every unit is run exactly once, whatever the placement).
#ifndef MF_FUSED_WPE
#define MF_FUSED_WPE 3
#endif
struct FusedRoles {
    unsigned *tab;  // [4096] per-CU tags + [2][4] unit counters (set tag & 1 is this launch's, the other one is cleared for the next)
    unsigned tag;   // increases with every launch of the context
};

template <int L, bool PACK16> __global__ __launch_bounds__(NT, MF_FUSED_WPE) void rx_fused_kernel(DecimArgs a, Enc128Args e, FusedRoles fr)
{
    constexpr int LDSDW = ENC128_LDS_BYTES / 4 > DecimLds<L, 2, PACK16>::dwords ? ENC128_LDS_BYTES / 4 : DecimLds<L, 2, PACK16>::dwords;
    __shared__ __attribute__((aligned(16))) int lds[LDSDW];
    __shared__ int s_unit;
    const int nmf = (a.nstreams * a.mf_wps + 3) / 4;
    const int npiece = a.nstreams * a.mf_npieces;
    const int nenc = 2 * e.nlist;
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned key = ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);
        unsigned *cnt = fr.tab + 4096 + 4 * (fr.tag & 1u);
        if (blockIdx.x == 0) { unsigned *nxt = fr.tab + 4096 + 4 * ((fr.tag + 1u) & 1u); nxt[0] = 0u; nxt[1] = 0u; nxt[2] = 0u; }
        const bool first = atomicMax(fr.tab + key, fr.tag) < fr.tag;
        int unit = -1;
        if (first) { const unsigned u = atomicAdd(cnt + 0, 1u); if (u < (unsigned)nmf) unit = (int)u; }
        if (unit < 0) { const unsigned u = atomicAdd(cnt + 1, 1u); if (u < (unsigned)npiece) unit = nmf + (int)u; }
        if (unit < 0) { const unsigned u = atomicAdd(cnt + 2, 1u); if (u < (unsigned)nenc) unit = nmf + npiece + (int)u; }
        if (unit < 0) { const unsigned u = atomicAdd(cnt + 0, 1u); if (u < (unsigned)nmf) unit = (int)u; }
        s_unit = unit;
    }
    __syncthreads();
    const int bx = s_unit;
    if (bx < 0) return;
    const int ndec = nmf + npiece;
    if (bx >= ndec) {
        gf_encode128_wg(e, bx - ndec, reinterpret_cast<unsigned char *>(lds));
        return;
    }
    if (bx >= nmf) {
        const int lx = bx - nmf;
        const int stream = lx / a.mf_npieces, piece = lx - stream * a.mf_npieces;
        if (piece == 0) {
            decim_piece<L, 2, PACK16>(a, lds, stream, 0, a.mf_head, true, false, piece, a.mf_npieces);
        } else {
            const size_t s0 = a.mf_tail_start + (size_t)(piece - 1) * a.mf_tail_seg;
            size_t s1 = s0 + a.mf_tail_seg;
            if (s1 > a.n_used || piece == a.mf_npieces - 1) s1 = a.n_used;
            decim_piece<L, 2, PACK16>(a, lds, stream, s0, s1, false, piece == a.mf_npieces - 1, piece, a.mf_npieces);
        }
        return;
    }
    const int gw = __builtin_amdgcn_readfirstlane(bx * 4 + (int)(threadIdx.x >> 6));
    if (gw >= a.nstreams * a.mf_wps) return;
    mf_wave<L, 0>(a, gw, 0u);
}

template <int L> hipError_t launch_fused(bool pack16, const DecimArgs &a, const Enc128Args &e, unsigned *roles, unsigned tag, hipStream_t stream)
{
    const int ndec = (a.nstreams * a.mf_wps + 3) / 4 + a.nstreams * a.mf_npieces;
    const dim3 grid(ndec + 2 * e.nlist), block(NT);
    FusedRoles fr;
    fr.tab = roles; fr.tag = tag;
    if (pack16) hipLaunchKernelGGL((rx_fused_kernel<L, true>), grid, block, 0, stream, a, e, fr);
    else hipLaunchKernelGGL((rx_fused_kernel<L, false>), grid, block, 0, stream, a, e, fr);
    return hipGetLastError();
}

template <int L, int NG, bool FR> void launch_mf2(bool pack16, const DecimArgs &a, dim3 grid, dim3 block, hipStream_t stream)
{
    if (pack16) hipLaunchKernelGGL((decim_mfma_kernel<L, true, NG, FR>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((decim_mfma_kernel<L, false, NG, FR>), grid, block, 0, stream, a);
}

template <int L> hipError_t launch_mf(bool pack16, const DecimArgs &a, hipStream_t stream)
{
    const int nleg = a.mf_piece_wgs > 0 ? a.mf_piece_wgs : a.nstreams * a.mf_npieces;
    const int nmf = (a.nstreams * a.mf_wps + 3) / 4;
    const dim3 grid(nleg + nmf), block(mf_block_threads(L));
    // ring depth: only the LDS-DMA kernel (decimate16) has one, and only it comes in both depths (DecimArgs::mf_ring);
    // frame_mode: the matrix-core waves store in the frame layout like the VALU pieces beside them
    if constexpr (mf_dma_applies(L)) {
        if (a.mf_ring == 3) {
            if (a.frame_mode) launch_mf2<L, 3, true>(pack16, a, grid, block, stream);
            else launch_mf2<L, 3, false>(pack16, a, grid, block, stream);
            return hipGetLastError();
        }
        if (a.mf_ring == 2) { // (experiment: two waves per SIMD, 72 KiB of ring per workgroup, two workgroups per CU)
            if (a.frame_mode) launch_mf2<L, 2, true>(pack16, a, grid, block, stream);
            else launch_mf2<L, 2, false>(pack16, a, grid, block, stream);
            return hipGetLastError();
        }
    }
    if (a.frame_mode) launch_mf2<L, 4, true>(pack16, a, grid, block, stream);
    else launch_mf2<L, 4, false>(pack16, a, grid, block, stream);
    return hipGetLastError();
}

} // namespace

bool plan_decimate_mfma(int log2decim, int fcpos, size_t n_used, int nstreams, size_t span_override, int n_cu, DecimArgs *a)
{
    // one wave per SIMD on all but one CU of every XCD (MI355X: 8 XCDs x 32 CUs -> 31 workgroups per XCD, 992 waves)
    const int nxcd = n_cu >= 64 && n_cu % 8 == 0 ? 8 : 1;
    const size_t waves1 = (size_t)4 * (size_t)(n_cu > nxcd ? n_cu - nxcd : n_cu);
    const size_t waves3 = 3 * waves1 - waves1 / 13; // a round of three waves per SIMD with some slack (2900 on MI355X)
    if (fcpos != 2 || log2decim < 2 || log2decim > 6) return false; // (decimate2_cen: the VALU kernel is HBM-bound, 63 % against 59 %)
    const size_t W = (size_t)64 << log2decim;     // one period of the schedule = the warm-up
    const size_t head = W > 2048 ? W : 2048;      // VALU head piece: whole passes, >= the warm-up of the first span
    if (n_used <= head) return false;
    const size_t n = n_used - head;
    size_t S;
    if (span_override) {
        S = (span_override + W - 1) / W * W;
    } else {
        const size_t total = n * (size_t)nstreams;
        if (log2decim <= 3) {
            // short cascades (decimate4 / 8): TWO waves per SIMD, 62 workgroups per XCD (1984 waves), the spans sized from
            // the wave count like below (tools/sweep_span.sh: 0.238 / 0.262 ms per 2^28 samples against 0.248 / 0.271 with
            // one round of three waves per SIMD, 0.250 / 0.318 with one wave per SIMD); three per SIMD for bigger banks
            S = (total / (waves3 * 8) + W - 1) / W * W;
            const size_t wps2 = 2 * waves1 / (size_t)nstreams;
            if (wps2 >= 1) {
                size_t S2 = n / (8 * wps2) / W * W;
                if (S2 == 0 || n / (8 * S2) > wps2) S2 += W; // (rounding down must not add a wave)
                if (S2 <= 256 * W && S2 >= 8 * W) S = S2;
            }
        } else {
            // decimate16 and up: with four and more stages per step a single wave keeps its SIMD as busy as three do
            // (tools/sweep_span.sh, tools/bench_streams.sh), so the spans can be long and the warm-up small (3 % at 32 Ki
            // samples and L = 4 against the 9 % of a three-waves-per-SIMD round).  One wave per SIMD: 31 workgroups per XCD
            // (992 waves; with all 32 CUs of an XCD taken -- 1016 waves -- the launch is 3 % slower, with 1056 it takes
            // half as long again: the launch lasts as long as its fullest CU), the spans as long as that allows, sized
            // from the wave count so that the VALU tail stays short.  Banks too big for that (spans beyond the limit of
            // the planner): spans of 32 Ki samples, the waves are dealt dynamically over many rounds.
            const size_t SL = 32768 > 8 * W ? 32768 : 8 * W;
            // (a->mf_ring == 2, decimate16 only: the two-waves-per-SIMD experiment -- 62 workgroups per XCD, spans half as long)
            const size_t wps1 = (mf_dma_applies(log2decim) && a->mf_ring == 2 ? 2 * waves1 : waves1) / (size_t)nstreams;
            S = SL;
            if (wps1 >= 1) {
                size_t S1 = n / (8 * wps1) / W * W;
                if (S1 == 0 || n / (8 * S1) > wps1) S1 += W; // (rounding down must not add a wave)
                // long periods (decimate32 / 64: W = 2048 / 4096) make that rounding coarse: one period less per span and a few
                // waves more than 31 CUs per XCD hold -- they land on the 32nd -- costs ~3 %, a period more of span + warm-up per
                // wave costs W / (S + W) (profiles/r04_decim_paths.txt: decimate64 0.2751 -> 0.2533 ms, decimate32 0.2513 -> 0.2447)
                if (S1 > 8 * W) {
                    const size_t S0 = S1 - W, wps0 = n / (8 * S0);
                    if (wps0 * (size_t)nstreams <= (size_t)4 * (size_t)n_cu && (S0 + W) * 105 < (S1 + W) * 100) S1 = S0;
                }
                if (S1 <= 256 * W) S = S1;
            }
        }
        // small calls: short spans (the launch lasts (W + S) / 32 steps of ~0.2 us whatever the size of the call)
        const size_t smin = (log2decim <= 3 ? 8 : 2) * W;
        if (S < smin) S = smin;
        if (S > 256 * W) S = 256 * W;
    }
    size_t wps = n / (8 * S);
    if (wps == 0 || wps > 0x7fffffffu / (size_t)nstreams) return false;
    if (8 * S * 4 >= 0xffffffffu) return false;   // lane offsets inside a wave are 32 bits
    size_t tail_start = head + wps * 8 * S;
    // the register-ring waves read up to 8 steps (256 samples) past their last span (the LDS-DMA ring clamps its run-ahead)
    if (n_used - tail_start < 256u) {
        if (--wps == 0) return false;
        tail_start = head + wps * 8 * S;
    }
    const size_t tail = n_used - tail_start;
    const size_t seg = 16384;                     // 8 passes of the VALU code per tail piece
    size_t ntail = (tail + seg - 1) / seg;
    if (ntail == 0) ntail = 1;                    // the (possibly empty) last piece stores the bank state
    a->mf_head = head;
    a->mf_span = S;
    a->mf_wps = (int)wps;
    a->mf_tail_start = tail_start;
    a->mf_tail_seg = seg;
    a->mf_npieces = 1 + (int)ntail;
    {
        // piece workgroups (see decim_mfma_kernel): one piece each, unless the matrix-core workgroups own their CUs (LDS-DMA ring) and
        // leave some free: then one EARLY workgroup per free CU with as many pieces as fit beside the matrix part (a piece of 16 Ki
        // samples takes ~21 us, a step of the matrix-core waves ~0.2 us: half of the matrix part's time), the rest one piece each
        const int items = nstreams * a->mf_npieces, nmf = (int)(((size_t)nstreams * wps + 3) / 4);
        a->mf_piece_early = 0; a->mf_piece_share = 0;
        if (mf_dma_applies(log2decim) && n_cu > nmf) {
            int early = n_cu - nmf;
            if (early > items) early = items;
            int share = (int)((W + S) / 6720);
            if (share < 3) share = 3;
            if (share > (items + early - 1) / early) share = (items + early - 1) / early;
            a->mf_piece_early = early; a->mf_piece_share = share;
        }
        const int rest = items - a->mf_piece_early * a->mf_piece_share;
        a->mf_piece_wgs = a->mf_piece_early + (rest > 0 ? rest : 0);
    }
    return true;
}

hipError_t launch_rx_fused(int log2decim, bool pack16, const DecimArgs &a, const Enc128Args &e, unsigned *roles, unsigned tag, hipStream_t stream)
{
    switch (log2decim) {
    case 2: return launch_fused<2>(pack16, a, e, roles, tag, stream);
    case 3: return launch_fused<3>(pack16, a, e, roles, tag, stream);
    case 4: return launch_fused<4>(pack16, a, e, roles, tag, stream);
    }
    return hipErrorInvalidValue; // (decimate32 / 64: 180-220 VGPRs, no room for encoder waves beside them)
}

hipError_t launch_decimate_mfma(int log2decim, bool pack16, const DecimArgs &a, hipStream_t stream)
{
    switch (log2decim) {
    case 2: return launch_mf<2>(pack16, a, stream);
    case 3: return launch_mf<3>(pack16, a, stream);
    case 4: return launch_mf<4>(pack16, a, stream);
    case 5: return launch_mf<5>(pack16, a, stream);
    case 6: return launch_mf<6>(pack16, a, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace sdrhip
