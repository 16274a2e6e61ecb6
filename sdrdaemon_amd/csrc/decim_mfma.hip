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
//    cross-lane traffic except the centre-tap ring below and the final I / Q pairing.
//  * exactness: int16 input x = lo + 256 hi + 128 with signed bytes lo = (x & 255) ^ 128, hi = x >> 8 (the 128
//    becomes a constant in the accumulator); stage outputs |v| <= 2^18 are split as v = b0 + 256 b1 + 65536 b2
//    with signed bytes b = bytes of (v + 0x808080) ^ 0x808080; taps h = h0 + 256 h1.  The limb products of
//    equal weight share an accumulator (|sum| < 2^21: no overflow), the four accumulators are recombined with
//    shifts modulo 2^32 = the reference's wrap-around int32 sum.
//  * the centre tap (x[2k - 30] << 13) needs no multiplier: the even outputs of a stage (even raw samples for the
//    first stage) go as int32 through a 32-entry ring per column and stage in LDS (written by the lane that
//    produced them, read four at a time by the lane that owns outputs k .. k+3) and enter the accumulator as
//    (e << 13) + c.  The ring is private to the wave: no barrier, program order of the wave's DS operations.
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
                        if (d >= 0 && d <= 31) {
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

__device__ __forceinline__ int4_t mfma(int4_t a, int4_t b, int4_t c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// centre-tap ring (LDS, private to a wave): [stage][column][MF_RING_PITCH] int32; entry m of a stage's even
// input plane sits at position m & 31
constexpr int MF_RING_PITCH = 36;                      // dwords per column (32 + pad: 16-byte aligned, spreads the banks)
constexpr int MF_RING_STAGE = 16 * MF_RING_PITCH;      // dwords per stage
template <int NS> constexpr int mf_ring_dwords() { return 4 * NS * MF_RING_STAGE; } // per workgroup of four waves

template <int NS> struct MfState {
    int4_t O[NS][3];        // odd plane window of every stage, one signed byte per entry and limb
    unsigned pend[NS][2];   // odd inputs of stage s >= 1: first half of the block being formed (limbs 0-1, limb 2)
};

struct MfConst {
    int4_t A[2][4]; // tap matrices (limbs h0, h1) in the four rotations
    int cin0;       // accumulator start of stage 0: 128 * (sum of the FIR taps) + (bias << 13)
    int cinN;       // other stages: bias << 13
    int *ring_wr;   // ring + column + 2 * q dwords: where this lane's even entries 2q, 2q+1 of a group of 8 go
    int *ring_rd;   // ring + column + 4 * q dwords: entries 4q .. 4q+3 of a block
    int *ring_rd4b; // entry 4q+4 of the odd-numbered block (wraps to entry 0 of the even one for q = 3)
};

struct MfOut {
    unsigned *p;          // where this lane's next four outputs go
    unsigned *dump;       // 16 bytes per lane that swallow the stores of the warm-up period (no branch in the loop body)
    int store;            // 0 during warm-up
    int norm, trunk;
};

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

// accumulator start of outputs 4q .. 4q+3 of tile I of stage S: c + (e << 13) with e = even-plane entry k - 15
// (IntHalfbandFilterEO1.h:136-142); entries 16 (I-1) + 4q+1 .. 4q+4 of the ring
template <int S, int I> __device__ __forceinline__ int4_t mf_centre(const MfConst &k, int c)
{
    constexpr int BP = (I + 3) & 1; // parity of block I - 1
    // the entries were written by OTHER lanes of this wave: the compiler must not move the reads above ring stores
    // it can prove disjoint per lane (the hardware executes a wave's DS operations in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int *rd = k.ring_rd + S * MF_RING_STAGE + 16 * BP;
    const int4_t v = *reinterpret_cast<const int4_t *>(rd);
    const int e4 = BP ? k.ring_rd4b[S * MF_RING_STAGE] : rd[4];
    int4_t r;
    r[0] = (int)lshl_add((unsigned)v[1], 13, (unsigned)c);
    r[1] = (int)lshl_add((unsigned)v[2], 13, (unsigned)c);
    r[2] = (int)lshl_add((unsigned)v[3], 13, (unsigned)c);
    r[3] = (int)lshl_add((unsigned)e4, 13, (unsigned)c);
    return r;
}

template <int NS, int S, int I> __device__ __forceinline__ void mf_stage(MfState<NS> &st, const MfConst &k, MfOut &oc, int comp)
{
    constexpr int PH = I & 3;
    const int4_t Ah0 = k.A[0][PH], Ah1 = k.A[1][PH];
    const int4_t z = {0, 0, 0, 0};
    int o[4];
    if constexpr (S == 0) {
        int4_t g0 = mfma(Ah0, st.O[0][0], mf_centre<0, I>(k, k.cin0));
        int4_t g1 = mfma(Ah0, st.O[0][1], z);
        int4_t g2 = mfma(Ah1, st.O[0][1], z);
        g1 = mfma(Ah1, st.O[0][0], g1);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (int)lshl_add(opaque(lshl_add((unsigned)g2[r], 8, (unsigned)g1[r])), 8, (unsigned)g0[r]) >> 13;
    } else {
        int4_t g0 = mfma(Ah0, st.O[S][0], mf_centre<S, I>(k, k.cinN));
        int4_t g1 = mfma(Ah0, st.O[S][1], z);
        int4_t g2 = mfma(Ah0, st.O[S][2], z);
        int4_t g3 = mfma(Ah1, st.O[S][2], z);
        g1 = mfma(Ah1, st.O[S][0], g1);
        g2 = mfma(Ah1, st.O[S][1], g2);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o[r] = (int)lshl_add(opaque(lshl_add(opaque(lshl_add((unsigned)g3[r], 8, (unsigned)g2[r])), 8, (unsigned)g1[r])), 8, (unsigned)g0[r]) >> 13;
    }

    if constexpr (S < NS - 1) {
        // outputs 4q .. 4q+3 of this tile: r = 0, 2 are even inputs of stage S+1 (entries 8 I + 2q, + 1 of its even
        // plane: to the ring), r = 1, 3 odd ones (to the window, one signed byte per limb)
        constexpr int SIG = I & 1, NJ = (I >> 1) & 3;
        *reinterpret_cast<int2_t *>(k.ring_wr + (S + 1) * MF_RING_STAGE + 8 * (I & 3)) = (int2_t){o[0], o[2]};
        const unsigned u1 = (unsigned)o[1] + 0x808080u, u3 = (unsigned)o[3] + 0x808080u;
        const unsigned po = perm(u3, u1, 0x05010400u), po2 = perm(u3, u1, 0x0c0c0602u);
        if constexpr (SIG == 0) {
            st.pend[S + 1][0] = po; st.pend[S + 1][1] = po2;
        } else {
            const unsigned X = 0x80808080u;
            st.O[S + 1][0][NJ] = (int)(perm(po, st.pend[S + 1][0], 0x05040100u) ^ X);
            st.O[S + 1][1][NJ] = (int)(perm(po, st.pend[S + 1][0], 0x07060302u) ^ X);
            st.O[S + 1][2][NJ] = (int)(perm(po2, st.pend[S + 1][1], 0x05040100u) ^ X);
            mf_stage<NS, S + 1, (I >> 1)>(st, k, oc, comp);
        }
    } else {
        // lanes n = 2p (I) and 2p + 1 (Q) hold the same outputs: both pack the same dwords and store them to the
        // same place (no divergence, no branch: the loop body stays one basic block and hipcc's vmcnt waits stay
        // exact); the warm-up period stores into the dump slot
        unsigned pk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int other = __builtin_amdgcn_update_dpp(0, o[r], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
            pk[r] = final_pack(comp ? other : o[r], comp ? o[r] : other, oc.norm, oc.trunk);
        }
        unsigned *dst = oc.store ? oc.p : oc.dump;
        *reinterpret_cast<uint4_t *>(dst) = (uint4_t){pk[0], pk[1], pk[2], pk[3]};
        oc.p += oc.store ? 16 : 0;
    }
}

template <int NS> __device__ __forceinline__ void mf_wave(const DecimArgs &a, int gw, int *ring)
{
    constexpr int L = NS;
    constexpr int P = 4 << (NS - 1);    // first-stage steps per period
#ifndef MF_DEPTH
#define MF_DEPTH 4
#endif
    constexpr int D = MF_DEPTH;         // steps of loads in flight
    constexpr size_t W = (size_t)64 << L; // warm-up = one period, raw samples
    static_assert(P % D == 0, "prefetch ring");
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, q = lane >> 4, comp = n & 1, p = n >> 1;
    const int stream = gw / a.mf_wps, ws = gw - stream * a.mf_wps;
    const size_t S = a.mf_span;
    const size_t wave_start = a.mf_head + (size_t)ws * 8 * S; // first stored raw sample of column pair 0
    const char *wbase = reinterpret_cast<const char *>(a.in) + ((size_t)stream * a.in_stride + wave_start - W) * 4;
    const unsigned loff = (unsigned)((size_t)p * S * 4) + 16u * (unsigned)q;
    const int T = (int)((W + S) / 32); // steps
    const int nper = T / P;

    MfConst k;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) k.A[m][ph] = *reinterpret_cast<const int4_t *>(&mf_atab.w[m][ph][lane][0]);
    const int b13 = a.bias << 13;
    k.cin0 = (int)opaque(128u * (unsigned)mf_tap_sum() + (unsigned)b13); // (opaque: hipcc otherwise splits it into (e + bias) << 13 + c, two ops)
    k.cinN = b13;
    {
        int *col = ring + n * MF_RING_PITCH;
        k.ring_wr = col + 2 * q;
        k.ring_rd = col + 4 * q;
        k.ring_rd4b = col + ((20 + 4 * q) & 31);
        for (int i = lane; i < NS * MF_RING_STAGE; i += 64) ring[i] = 0;
    }

    MfOut oc;
    oc.store = 0;
    oc.dump = a.mf_dump + 4 * lane;
    oc.norm = a.norm; oc.trunk = a.trunk;
    {
        unsigned *obase = reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride;
        const size_t first = ((wave_start + (size_t)p * S) >> L) + 4u * (unsigned)q; // this lane's first output
        oc.p = obase + first;
    }

    MfState<NS> st;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int b = 0; b < 3; ++b) st.O[s][b] = (int4_t){0, 0, 0, 0};
        st.pend[s][0] = st.pend[s][1] = 0u;
    }

    const unsigned selc = comp ? 0x07030602u : 0x05010400u;
    const int esh = comp ? 16 : 0;
    uint4_t ld[D][2];
    // step g of this lane's column: 128 bytes at src + 128 g.  No bounds handling: the loads run D steps past the end
    // of the span, i.e. into the next span or (last span of a stream) the first 32 D samples of the tail that
    // plan_decimate_mfma() guarantees
    const char *src = wbase + loff;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ld[d][0] = *reinterpret_cast<const uint4_t *>(src + 128 * d);
        ld[d][1] = *reinterpret_cast<const uint4_t *>(src + 128 * d + 64);
    }

    for (int per = 0; per < nper; ++per) {
        oc.store = per > 0;
        mf_static_for<P>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int slot = i % D;
            const uint4_t r0 = ld[slot][0], r1 = ld[slot][1];
            ld[slot][0] = *reinterpret_cast<const uint4_t *>(src + 128 * (i + D));
            ld[slot][1] = *reinterpret_cast<const uint4_t *>(src + 128 * (i + D) + 64);
            // raw samples 4q .. 4q+3 (r0) and 16 + 4q .. (r1) of the step's 32: x, z even; y, w odd.
            // even ones: entries 2q, 2q+1 and 8 + 2q, 8 + 2q+1 of block i of the first stage's even plane
            int *wr = k.ring_wr + 16 * (i & 1);
            *reinterpret_cast<int2_t *>(wr) = (int2_t){sbfe16(r0.x, esh), sbfe16(r0.z, esh)};
            *reinterpret_cast<int2_t *>(wr + 8) = (int2_t){sbfe16(r1.x, esh), sbfe16(r1.z, esh)};
            const unsigned ao = perm(r0.w, r0.y, selc), bo = perm(r1.w, r1.y, selc);
            st.O[0][0][i & 3] = (int)(perm(bo, ao, 0x05040100u) ^ 0x80808080u);
            st.O[0][1][i & 3] = (int)perm(bo, ao, 0x07060302u);
            mf_stage<NS, 0, i>(st, k, oc, comp);
        });
        src += 128 * P;
    }
}

// grid.x = nstreams * mf_npieces VALU workgroups (head + tail pieces of every stream), then the matrix-core
// workgroups (four waves = four groups of 8 spans each)
template <int L, bool PACK16> __global__ __launch_bounds__(NT, MF_WAVES) void decim_mfma_kernel(DecimArgs a)
{
    constexpr int LDSDW = DecimLds<L, 2, PACK16>::dwords > mf_ring_dwords<L>() ? DecimLds<L, 2, PACK16>::dwords : mf_ring_dwords<L>();
    __shared__ __attribute__((aligned(16))) int lds[LDSDW];
    const int nleg = a.nstreams * a.mf_npieces;
    const int bx = blockIdx.x;
    if (bx < nleg) {
        const int stream = bx / a.mf_npieces, piece = bx - stream * a.mf_npieces;
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
    const int gw = __builtin_amdgcn_readfirstlane((bx - nleg) * 4 + (int)(threadIdx.x >> 6));
    if (gw >= a.nstreams * a.mf_wps) return;
    mf_wave<L>(a, gw, lds + (threadIdx.x >> 6) * (L * MF_RING_STAGE));
}

template <int L> hipError_t launch_mf(bool pack16, const DecimArgs &a, hipStream_t stream)
{
    const int nleg = a.nstreams * a.mf_npieces;
    const int nmf = (a.nstreams * a.mf_wps + 3) / 4;
    const dim3 grid(nleg + nmf), block(NT);
    if (pack16) hipLaunchKernelGGL((decim_mfma_kernel<L, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((decim_mfma_kernel<L, false>), grid, block, 0, stream, a);
    return hipGetLastError();
}

} // namespace

bool plan_decimate_mfma(int log2decim, int fcpos, size_t n_used, int nstreams, size_t span_override, DecimArgs *a)
{
    if (fcpos != 2 || log2decim < 2 || log2decim > 4) return false;
    const size_t W = (size_t)64 << log2decim;     // one period of the schedule = the warm-up
    const size_t head = W > 2048 ? W : 2048;      // VALU head piece: whole passes, >= the warm-up of the first span
    if (n_used <= head) return false;
    const size_t n = n_used - head;
    size_t S;
    if (span_override) {
        S = (span_override + W - 1) / W * W;
    } else {
        // one round of three waves per SIMD (the kernel's 160 VGPRs admit three 4-wave workgroups per CU; the VALU
        // pieces take a few of the 768 slots): ~2900 waves of 8 spans when the call is big enough, longer spans
        // beyond; at least 8 warm-ups per span (<= 12 % overhead)
        S = (n * (size_t)nstreams / (2900 * 8) + W - 1) / W * W;
        if (S < 8 * W) S = 8 * W;
        if (S > 256 * W) S = 256 * W;
    }
    size_t wps = n / (8 * S);
    if (wps == 0 || wps > 0x7fffffffu / (size_t)nstreams) return false;
    if (8 * S * 4 >= 0xffffffffu) return false;   // lane offsets inside a wave are 32 bits
    size_t tail_start = head + wps * 8 * S;
    if (n_used - tail_start < 256) { // the waves read up to 256 samples past their last span (prefetch)
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
    return true;
}

hipError_t launch_decimate_mfma(int log2decim, bool pack16, const DecimArgs &a, hipStream_t stream)
{
    switch (log2decim) {
    case 2: return launch_mf<2>(pack16, a, stream);
    case 3: return launch_mf<3>(pack16, a, stream);
    case 4: return launch_mf<4>(pack16, a, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace sdrhip
