// interp_mfma.hip -- half-band interpolator cascades on the gfx950 matrix cores (MI355X).
//
// Same arithmetic as interp_kernels.hip (Interpolators::interpolate{4..32}_cen, Interpolators.cpp:47-362, over
// IntHalfbandFilterEO1/DB<64|32|16>::myInterpolate, IntHalfbandFilterEO1.h:44-65,149-168), bit for bit:
//     v[2m]   = u[m - O/4]
//     v[2m+1] = (sum_{i < O/4} c[i] * (u[m - (O/2 - 1) + i] + u[m - i])) >> 13
// but every stage is an exact integer matrix product on the int8 matrix cores, built like the decimator of
// decim_mfma.hip (DESIGN.md "K5m"):
//
//  * a wave owns 16 columns = 8 consecutive spans of one stream x {I, Q}.  A tile = 16 consecutive outputs of a stage
//    (8 inputs, both phases) of all columns: D = A x B, the rows of A are the output times, its K slots the taps over
//    a window of the stage's inputs, B holds one signed byte ("limb") of every window entry.  Even rows (the pure
//    delay) carry the single entry 2^13 = 32 * 256 in the high tap limb, so that the same ">> 13" serves both phases.
//  * the 16 outputs of a tile are exactly one 16-entry block of the next stage's input, and lane (column n, q) holds
//    outputs 4q .. 4q+3 = the four bytes of its dword of that block: limb split and re-pack stay inside the lane.  A
//    block feeds two tiles of the next stage (halves 0 / 1: different A matrices), which feed four ... : the cascade is
//    walked depth first, 2^L final tiles (16 x 2^L outputs per column) per block of 16 stream inputs.
//  * windows: {newest block, previous block} (orders 32 and 16: the taps reach back 15 / 7 inputs; K = 32 MFMAs) or
//    {newest, previous, the one before} (order 64: 31 inputs; K = 64); a new block shifts the older ones down
//    (one v_mov per limb), so there is one A matrix per (filter, tap limb, half) and no rotation.
//  * exactness as in decim_mfma.hip: int16 input x = lo + 256 hi + 128; stage outputs as the 19-bit field of
//    acc >>> 13 = b0 + 256 b1 + 65536 b2 - 229248 (signed bytes b0, b1, 3-bit b2); taps h = h0 + 256 h1; limb
//    products of equal weight share an accumulator, Horner recombination modulo 2^32.
//  * the last stage swaps I / Q between lane pairs (one DPP move), packs int16 pairs and stores 16 bytes per lane pair:
//    64 contiguous bytes per span and tile.  No LDS, no barrier in the matrix-core waves.
//  * a span is preceded by 64 inputs (four blocks; the cascade's memory is 43) of warm-up with the stores diverted.
//    Segment 0 and the tail segments of every stream run the VALU code (interp_body.h) in the same launch: they own
//    the bank state.
#include "interp_body.h"

#include <type_traits>
#include <utility>

#ifndef IM_WAVES
#define IM_WAVES 3
#endif

namespace sdrhip {
namespace {

typedef int int2_t __attribute__((ext_vector_type(2)));

constexpr int IM_LIMB_BIAS = 128 + 32768 - 262144; // value of a stage output = its limbs + this
constexpr int IM_WARM_BLOCKS = 4;                  // 64 inputs

__host__ __device__ constexpr int im_coef(int O, int d) { return d < O / 4 ? tap(O, d) : tap(O, O / 2 - 1 - d); }
__host__ __device__ constexpr int im_tap_sum(int O)
{
    int s = 0;
    for (int i = 0; i < O / 4; ++i) s += 2 * tap(O, i);
    return s;
}
// limb m (0: h0, 1: h1) of the coefficient that output row r of half h applies to the input E entries after the start
// of the newest block (E < 0: older blocks); input index of the row: 8 h + (r >> 1)
__host__ __device__ constexpr int im_a(int O, int m, int h, int r, int E)
{
    const int mm = 8 * h + (r >> 1);
    int v = 0;
    if ((r & 1) == 0) {
        if (E == mm - O / 4) v = 8192;
    } else {
        const int d = mm - E;
        if (d >= 0 && d < O / 2) v = im_coef(O, d);
    }
    const int h1 = (v + 128) >> 8, h0 = v - 256 * h1;
    return m == 0 ? h0 : h1;
}

// fragment tables: lane = row r + 16 g; byte t of dword j <-> entry 4 g + t of the block j blocks before the newest
struct ImTab64 { unsigned w[2][2][64][4]; };    // order 64: [tap limb][half][lane][dword]
struct ImTab32 { unsigned w[2][2][2][64][2]; }; // [0: order 32, 1: order 16][tap limb][half][lane][dword]

constexpr ImTab64 im_make64()
{
    ImTab64 T{};
    for (int m = 0; m < 2; ++m)
        for (int h = 0; h < 2; ++h)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    unsigned word = 0;
                    for (int t = 0; t < 4; ++t) {
                        const int v = j < 3 ? im_a(64, m, h, lane & 15, 4 * (lane >> 4) + t - 16 * j) : 0;
                        word |= (unsigned)(v & 0xff) << (8 * t);
                    }
                    T.w[m][h][lane][j] = word;
                }
    return T;
}
constexpr ImTab32 im_make32()
{
    ImTab32 T{};
    for (int f = 0; f < 2; ++f)
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 2; ++h)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 2; ++j) {
                        unsigned word = 0;
                        for (int t = 0; t < 4; ++t) {
                            const int v = im_a(f == 0 ? 32 : 16, m, h, lane & 15, 4 * (lane >> 4) + t - 16 * j);
                            word |= (unsigned)(v & 0xff) << (8 * t);
                        }
                        T.w[f][m][h][lane][j] = word;
                    }
    return T;
}
__device__ const ImTab64 im_tab64 = im_make64();
__device__ const ImTab32 im_tab32 = im_make32();

__device__ __forceinline__ int4_t mfma64(int4_t a, int4_t b, int4_t c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int4_t mfma32(int2_t a, int2_t b, int4_t c)
{
    return __builtin_amdgcn_mfma_i32_16x16x32_i8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// (a << sh) + b = one v_lshl_add_u32; plain C so that hipcc sees the MFMA hazards (decim_mfma.hip)
__device__ __forceinline__ unsigned lshl_add(unsigned a, int sh, unsigned b) { return (a << sh) + b; }
__device__ __forceinline__ unsigned opaque(unsigned v)
{
    asm("" : "+v"(v));
    return v;
}

template <int NS> struct ImState {
    int4_t W0[2];        // first stage (order 64), limbs lo / hi: {newest block, previous, the one before, -}
    int2_t W[NS][3];     // stage s >= 1, three limbs: {newest block, previous}
};

struct ImConst {
    int4_t A64[2][2];    // [tap limb][half]
    int2_t A32[2][2][2]; // [0: order 32, 1: order 16][tap limb][half]
    int4_t c0;           // accumulator start of the first stage: {even row, odd row, even, odd}
    int4_t cN[2];        // order 32, order 16
    unsigned sel_iq;     // final pack {this lane's value, the neighbour's} -> (I lo16, Q lo16)
};

struct ImOut {
    unsigned *p;         // where this lane's next four outputs go
    unsigned *dump;      // sink of the warm-up stores
    int store;
};

template <int NS, int S, int H> __device__ __forceinline__ void im_tile(ImState<NS> &st, const ImConst &k, ImOut &oc)
{
    const int4_t z = {0, 0, 0, 0};
    unsigned acc[4];
    if constexpr (S == 0) {
        const int4_t A0 = k.A64[0][H], A1 = k.A64[1][H];
        int4_t g0 = mfma64(A0, st.W0[0], k.c0);
        int4_t g1 = mfma64(A0, st.W0[1], z);
        int4_t g2 = mfma64(A1, st.W0[1], z);
        g1 = mfma64(A1, st.W0[0], g1);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = lshl_add(opaque(lshl_add((unsigned)g2[r], 8, (unsigned)g1[r])), 8, (unsigned)g0[r]);
    } else {
        constexpr int F = S == 1 ? 0 : 1;
        const int2_t A0 = k.A32[F][0][H], A1 = k.A32[F][1][H];
        int4_t g0 = mfma32(A0, st.W[S][0], k.cN[F]);
        int4_t g1 = mfma32(A0, st.W[S][1], z);
        int4_t g2 = mfma32(A0, st.W[S][2], z);
        int4_t g3 = mfma32(A1, st.W[S][2], z);
        g1 = mfma32(A1, st.W[S][0], g1);
        g2 = mfma32(A1, st.W[S][1], g2);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[r] = lshl_add(opaque(lshl_add(opaque(lshl_add((unsigned)g3[r], 8, (unsigned)g2[r])), 8, (unsigned)g1[r])), 8, (unsigned)g0[r]);
    }

    if constexpr (S < NS - 1) {
        // the tile's outputs 4q .. 4q+3 (19-bit fields) = this lane's dword of the next stage's new block
        const unsigned u0 = acc[0] >> 13, u1 = acc[1] >> 13, u2 = acc[2] >> 13, u3 = acc[3] >> 13;
        const unsigned pa = perm(u1, u0, 0x05010400u), pb = perm(u3, u2, 0x05010400u);   // {b0, b0', b1, b1'}
        const unsigned pa2 = perm(u1, u0, 0x0c0c0602u), pb2 = perm(u3, u2, 0x0c0c0602u); // {b2, b2', 0, 0}
#pragma unroll
        for (int l = 0; l < 3; ++l) st.W[S + 1][l][1] = st.W[S + 1][l][0];
        st.W[S + 1][0][0] = (int)(perm(pb, pa, 0x05040100u) ^ 0x80808080u);
        st.W[S + 1][1][0] = (int)(perm(pb, pa, 0x07060302u) ^ 0x80808080u);
        st.W[S + 1][2][0] = (int)(perm(pb2, pa2, 0x05040100u) ^ 0x04040404u);
        im_tile<NS, S + 1, 0>(st, k, oc);
        im_tile<NS, S + 1, 1>(st, k, oc);
    } else {
        // int16 truncation (Interpolators.cpp: the FixReal casts of the last stage), I / Q pairing, 16-byte store; both
        // lanes of a pair store the same dwords to the same place; the warm-up stores go to the dump slot
        unsigned pk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned o = acc[r] >> 13;
            const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)o, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
            pk[r] = perm(o, other, k.sel_iq);
        }
        unsigned *dst = oc.store ? oc.p : oc.dump;
        *reinterpret_cast<uint4_t *>(dst) = (uint4_t){pk[0], pk[1], pk[2], pk[3]};
        oc.p += oc.store ? 16 : 0;
    }
}

template <int NS> __device__ __forceinline__ void im_wave(const InterpArgs &a, int gw)
{
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, q = lane >> 4, comp = n & 1, p = n >> 1;
    const int stream = gw / a.mf_wps, ws = gw - stream * a.mf_wps;
    const size_t S = a.mf_span;                                        // inputs per span
    const size_t wave_start = a.mf_head + (size_t)ws * 8 * S;          // first input whose outputs column pair 0 stores
    const size_t col_start = wave_start + (size_t)p * S;
    const unsigned *src = reinterpret_cast<const unsigned *>(a.in) + (size_t)stream * a.in_stride + col_start - 16 * IM_WARM_BLOCKS + 4 * q;
    const int nblocks = (int)(S / 16) + IM_WARM_BLOCKS;

    ImConst k;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            k.A64[m][h] = *reinterpret_cast<const int4_t *>(&im_tab64.w[m][h][lane][0]);
#pragma unroll
            for (int f = 0; f < 2; ++f) k.A32[f][m][h] = *reinterpret_cast<const int2_t *>(&im_tab32.w[f][m][h][lane][0]);
        }
    {
        const int ce = (int)(8192u * 128u), co = (int)(128u * (unsigned)im_tap_sum(64)); // x = lo + 256 hi + 128
        k.c0 = (int4_t){ce, co, ce, co};
        const unsigned B = (unsigned)IM_LIMB_BIAS;
        k.cN[0] = (int4_t){(int)(8192u * B), (int)(B * (unsigned)im_tap_sum(32)), (int)(8192u * B), (int)(B * (unsigned)im_tap_sum(32))};
        k.cN[1] = (int4_t){(int)(8192u * B), (int)(B * (unsigned)im_tap_sum(16)), (int)(8192u * B), (int)(B * (unsigned)im_tap_sum(16))};
    }
    k.sel_iq = comp ? 0x05040100u : 0x01000504u;

    ImOut oc;
    oc.store = 0;
    oc.dump = a.mf_dump + 4 * lane;
    oc.p = reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride + (col_start << NS) + 4u * (unsigned)q;

    ImState<NS> st;
    st.W0[0] = st.W0[1] = (int4_t){0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int l = 0; l < 3; ++l) st.W[s][l] = (int2_t){0, 0};

    const unsigned selc = comp ? 0x07030602u : 0x05010400u; // {lo(a), lo(b), hi(a), hi(b)} of this lane's component
    // (the loads run one block past the end of the span: into the next span, or the tail that plan_interpolate_mfma()
    // guarantees)
    uint4_t nxt = *reinterpret_cast<const uint4_t *>(src);
    // every load issued so far (tables, first block) is consumed here, in front of the loop: a first use inside the loop
    // would make hipcc's wait for it part of EVERY iteration, where it is a wait for the previous block's sixteen stores
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            asm volatile("" : "+v"(k.A64[m][h]), "+v"(k.A32[0][m][h]), "+v"(k.A32[1][m][h]));
    asm volatile("" : "+v"(nxt));
#pragma unroll 1
    for (int b = 0; b < nblocks; ++b) {
        const uint4_t r = nxt;
        src += 16;
        nxt = *reinterpret_cast<const uint4_t *>(src);
        oc.store = b >= IM_WARM_BLOCKS;
        // inputs 4q .. 4q+3 of the block
        const unsigned pa = perm(r.y, r.x, selc), pb = perm(r.w, r.z, selc);
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            st.W0[l][2] = st.W0[l][1];
            st.W0[l][1] = st.W0[l][0];
        }
        st.W0[0][0] = (int)(perm(pb, pa, 0x05040100u) ^ 0x80808080u);
        st.W0[1][0] = (int)perm(pb, pa, 0x07060302u);
        im_tile<NS, 0, 0>(st, k, oc);
        im_tile<NS, 0, 1>(st, k, oc);
        // the next block's load is consumed HERE, inside the basic block that issued it: hipcc then waits with the exact
        // vmcnt (the load is older than this block's stores); a use behind the back edge gets vmcnt(0), i.e. a wait for
        // all of the block's stores at the top of every iteration
        asm volatile("" : "+v"(nxt));
    }
}

// grid.x = nstreams * mf_npieces VALU workgroups (segment 0 and the tail segments of every stream), then the
// matrix-core workgroups (four waves = four groups of 8 spans each)
template <int L> __global__ __launch_bounds__(NT, IM_WAVES) void interp_mfma_kernel(InterpArgs a)
{
    __shared__ __attribute__((aligned(16))) int lds[IGeo<L>::ldsDw]; // the VALU segments' stage buffers
    const int nleg = a.nstreams * a.mf_npieces;
    const int bx = blockIdx.x;
    if (bx < nleg) {
        const int stream = bx / a.mf_npieces, piece = bx - stream * a.mf_npieces;
        interp_segment<L>(a, piece == 0 ? 0 : a.mf_tail_seg + piece - 1, stream, lds);
        return;
    }
    const int gw = __builtin_amdgcn_readfirstlane((bx - nleg) * 4 + (int)(threadIdx.x >> 6));
    if (gw >= a.nstreams * a.mf_wps) return;
    im_wave<L>(a, gw);
}

template <int L> hipError_t launch_im(const InterpArgs &a, hipStream_t stream)
{
    const int nleg = a.nstreams * a.mf_npieces;
    const int nmf = (a.nstreams * a.mf_wps + 3) / 4;
    hipLaunchKernelGGL((interp_mfma_kernel<L>), dim3(nleg + nmf), dim3(NT), 0, stream, a);
    return hipGetLastError();
}

} // namespace

// Plans the matrix-core path for n_in inputs per stream: VALU segments of 512 inputs (a->nsub_per_seg = 1) for
// [0, 512) and the tail, spans of a->mf_span inputs in between.  false = use the VALU kernel for the whole call.
bool plan_interpolate_mfma(int log2interp, size_t n_in, int nstreams, size_t span_override, InterpArgs *a)
{
    if (log2interp < 2 || log2interp > 4) return false;
    const size_t SEG = 512, head = SEG;
    if (n_in < head + 2 * SEG) return false;
    const size_t n = n_in - head;
    size_t S;
    if (span_override) {
        S = (span_override + 63) / 64 * 64;
    } else {
        // one round of IM_WAVES waves per SIMD when the call is big enough; the warm-up is 64 inputs per span
        S = (n * (size_t)nstreams / ((size_t)1024 * IM_WAVES * 8 - 256) + 63) / 64 * 64;
        if (S < 1024) S = 1024;
        if (S > 65536) S = 65536;
    }
    size_t wps = n / (8 * S);
    if (wps == 0 || wps > 0x7fffffffu / (size_t)nstreams) return false;
    size_t tail_start = head + wps * 8 * S; // a multiple of 512: 8 S is
    if (n_in - tail_start < 64) {          // the waves read one block past their last span
        if (--wps == 0) return false;
        tail_start = head + wps * 8 * S;
    }
    const size_t nseg = (n_in + SEG - 1) / SEG;
    a->nsub_per_seg = 1;
    a->nseg = (int)nseg;
    a->mf_head = head;
    a->mf_span = S;
    a->mf_wps = (int)wps;
    a->mf_tail_seg = (int)(tail_start / SEG);
    a->mf_npieces = 1 + (int)(nseg - tail_start / SEG);
    if (nseg > 0x7fffffffu || a->mf_npieces < 2) return false; // (the last segment stores the bank state)
    return true;
}

hipError_t launch_interpolate_mfma(int log2interp, const InterpArgs &a, hipStream_t stream)
{
    switch (log2interp) {
    case 2: return launch_im<2>(a, stream);
    case 3: return launch_im<3>(a, stream);
    case 4: return launch_im<4>(a, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace sdrhip
