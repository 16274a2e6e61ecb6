// interp_wave.h -- K5w: the interpolator cascade as barrier-free, wave-private pipelines.
//
// Same arithmetic as interp_body.h (Interpolators::interpolate{4..64}_cen over IntHalfbandFilterEO1/DB<64|32|16>::myInterpolate,
// Interpolators.cpp:47-606, IntHalfbandFilterEO1.h:44-65,149-168), different machine mapping:
//  * a wave (64 lanes; a workgroup is one wave or four independent ones) owns a time slice of one stream and its own 7.5 KB of LDS: no s_barrier anywhere
//    (K5 needs 20 workgroup barriers per macro-cycle); the stage buffers are handed from stage to stage by the wave itself, LDS
//    operations of one wave execute in order, so a write followed by a read needs no synchronisation at all;
//  * blocks of 128 inputs, walked depth first: stage s takes 128 inputs per invocation (lanes 2j / 2j+1 = I / Q of inputs
//    4j .. 4j+3), the last stage 256 (both components per lane: packs int16 I/Q and stores 2 x 16 bytes per lane);
//  * stage 0 (order 64) reads its input as PACKED int16 pairs and runs on v_dot2_i32_i16: 16 dot products + 4.25 v_alignbit per
//    output instead of 16 adds + 16 multiply-adds (the raw samples are int16; every later stage sees 19-bit values);
//  * the last stage multiplies by 8 x tap: (acc >> 13) & 0xffff is then the HIGH half of the accumulator, and one v_perm_b32 both
//    shifts and packs I/Q (the truncation to int16 drops everything above bit 28 anyway);
#ifndef SDRHIP_INTERP_WAVE_H
#define SDRHIP_INTERP_WAVE_H
#include "interp_body.h"

namespace sdrhip {
namespace {

constexpr int WNT = 64;          // one wave
constexpr int WB = 128;          // inputs per block
constexpr int WWARM = 44;        // warm-up of a segment, inputs: the cascade's memory is 43 (32 + 16 / 2 + 8 / 4 + 8 / 8 [+ 8 / 16]); even (packed pairs)
constexpr int WCAP = 256;        // fresh entries a stage buffer (s >= 1) holds
constexpr int WSTR = HIST + WCAP; // 288 dwords = 72 sixteen-byte slots = 8 mod 16: I / Q lane pairs of a ds_read_b128 group hit distinct banks
// a segment's whole input (<= 2 x WPAIRS blocks) is loaded at the wave's start and parked in 4 x WPAIRS AGPRs.  Eight pairs for
// every ratio: the short cascades (interpolate4 / 8: 56-64 VGPRs, four times / twice the input per output) were tried with 4, 6,
// 12 and 16 -- longer segments cost a wave per SIMD and run 4-10 % slower, shorter ones change nothing (experiments_r04 batch 23)
constexpr int WPAIRS = 8;
constexpr int WG = 2;            // blocks staged at a time: one dwordx4 load per lane covers a PAIR of blocks (256 samples)
constexpr int W0HIST = 16;       // packed stage-0 plane: 16 dwords (32 entries) of history + WG x 64 fresh
constexpr int W0STR = 32 + 64 * WG; // ... padded to 16 mod 32 eight-byte slots: the I / Q lanes of a ds_read_b64 group hit distinct banks
static_assert(W0STR % 64 == 32 && W0STR >= W0HIST + 64 * WG, "stage-0 plane stride");

typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

// Q-plane swizzle of the stage buffers s >= 1 (round 5 experiment, OFF by default: WSWZ = 0).  A stage writes its 8 outputs per lane
// as two ds_write_b128; the lanes of an I / Q pair write the same positions of their planes, and the planes lie WSTR = 288 = 0
// (mod 32) dwords apart -- which the ds_read_b128 window reads want (I / Q lanes of a 16-lane service group then hit distinct
// 4-bank columns of the 64 banks), but a ds_write_b128 is serviced in groups of 8 lanes over 32 banks: the I and the Q lane of
// every pair collide, 2-way, on every write.  That is ALL of the kernel's bank conflicts (SQ_LDS_BANK_CONFLICT 19.3 M of
// SQ_LDS_IDX_ACTIVE 68.7 M cycles per 2^28 outputs = 28 %).  WSWZ = 4 stores entry e of the Q plane at position e ^ 4 (the halves
// of every aligned 8-dword group trade places: the Q lane's first ds_write_b128 goes to the upper half while the I lane's goes to
// the lower one; the windows are read as whole 4-dword groups from two lane-constant bases): the write conflicts are gone
// (-10.7 M cycles), but the window groups of ODD index now collide between I and Q lanes on the read side (8.6 M conflict cycles
// left, 14.8 %), and the launch time does not move: 0.2411 / 0.2411 ms against 0.2398 / 0.2419 ms, trace average 243.8 against
// 240.7 us (profiles/r05_k5w_swizzle.txt) -- the LDS array is busy 40 % of the launch, conflicts included, and is not what the
// kernel waits for; the swizzle also costs 8 VGPRs (interpolate8 loses its fifth wave per SIMD: +1.3 %).  A swizzle that is
// conflict-free on both sides exists (brute force over all half-swaps keyed by plane and index bits 3-5: 16 solutions), but every
// one of them needs bit 5 of the entry index, i.e. per-read address arithmetic instead of immediates (3-4 VALU per ds_read_b128 on
// a kernel whose VALU port is the busier resource).  Left in as the A / B partner: make EXTRA=-DWSWZ=4.
#ifndef WSWZ
#define WSWZ 0
#endif
// position of entry `idx` (a multiple of 4 plus x) in plane `comp`: idx ^ (comp ? WSWZ : 0)
__device__ __forceinline__ int wswz(int idx, int comp) { return idx ^ (comp ? WSWZ : 0); }

template <int NS_> struct WGeo {
    static constexpr int NS = NS_;
    static constexpr int base(int s) { return s == 0 ? 0 : 2 * W0STR + (s - 1) * 2 * WSTR; }
    static constexpr int ldsDw = 2 * W0STR + (NS - 1) * 2 * WSTR;
};

// taps of the order-64 stage over delays 0..31 (symmetric), packed for v_dot2_i32_i16 on (older, newer) sample pairs
__host__ __device__ constexpr int h64(int d) { return d < 16 ? T64[d] : T64[31 - d]; }
__host__ __device__ constexpr unsigned tap_pair(int e) { return ((unsigned)h64(2 * e + 1) & 0xffffu) | ((unsigned)h64(2 * e) << 16); }

// The lanes of the wave hand data to EACH OTHER through LDS.  The hardware executes a wave's LDS operations in order, so no
// instruction is needed -- but the compiler reasons per thread: it may move a lane's read above a write of the same lane that
// it can prove disjoint (and did: the second half of a stage's output write sank below the next stage's window reads).  A
// wavefront-scope fence + wave barrier pins the order of the accesses; neither emits an instruction.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int wdot2(unsigned a, unsigned taps, int acc)
{
    typedef short short2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, taps), acc, false);
}

// stage 0 (order 64) on packed int16 planes: `valid` inputs of the block -> 2 * valid entries at the start of stage 1's buffer
template <class G, bool FULL> __device__ __forceinline__ void wstage0(int *lds, int lane, int blk, int valid, int (&o)[8])
{
    const int j = lane >> 1, comp = lane & 1, m0 = 4 * j;
    if (!FULL && m0 >= valid) return;
    // window dword t <-> entries u[m0 - 32 + 2t], u[m0 - 32 + 2t + 1] of block blk of the group
    const unsigned *pl = reinterpret_cast<const unsigned *>(lds) + comp * W0STR + 64 * blk + 2 * j;
    unsigned W[18];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint2_t v = *reinterpret_cast<const uint2_t *>(pl + 2 * t);
        W[2 * t] = v.x; W[2 * t + 1] = v.y;
    }
    int acc[4] = {0, 0, 0, 0};
    if constexpr (G::NS == 3) {
        // interpolate8: the odd alignment A(t) <-> entries 2t - 1, 2t of the window is formed where it is used (twice each: 15 more
        // v_perm per invocation, 17 fewer live registers: 56 + 32 -> five waves per SIMD, +3.4 %; the other ratios run faster with
        // four, tools/experiments_r04/exp19.sh)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned a0 = __builtin_amdgcn_alignbit(W[16 - e], W[15 - e], 16), a2 = __builtin_amdgcn_alignbit(W[17 - e], W[16 - e], 16);
            acc[0] = wdot2(a0, tap_pair(e), acc[0]);
            acc[1] = wdot2(W[16 - e], tap_pair(e), acc[1]);
            acc[2] = wdot2(a2, tap_pair(e), acc[2]);
            acc[3] = wdot2(W[17 - e], tap_pair(e), acc[3]);
        }
    } else {
        unsigned A[18]; // the odd alignment: A[t] <-> entries 2t - 1, 2t of the window
#pragma unroll
        for (int t = 1; t < 18; ++t) A[t] = __builtin_amdgcn_alignbit(W[t], W[t - 1], 16);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[0] = wdot2(A[16 - e], tap_pair(e), acc[0]);
            acc[1] = wdot2(W[16 - e], tap_pair(e), acc[1]);
            acc[2] = wdot2(A[17 - e], tap_pair(e), acc[2]);
            acc[3] = wdot2(W[17 - e], tap_pair(e), acc[3]);
        }
    }
    // v[2m] = u[m - 16]: entries 16 .. 19 of the window
    o[0] = (int)(short)(W[8] & 0xffffu); o[2] = (int)W[8] >> 16; o[4] = (int)(short)(W[9] & 0xffffu); o[6] = (int)W[9] >> 16;
    o[1] = acc[0] >> 13; o[3] = acc[1] >> 13; o[5] = acc[2] >> 13; o[7] = acc[3] >> 13;
    int *nx = lds + G::base(1) + comp * WSTR + HIST + 2 * m0;
    *reinterpret_cast<int4_t *>(nx + wswz(0, comp)) = (int4_t){o[0], o[1], o[2], o[3]};
    *reinterpret_cast<int4_t *>(nx + wswz(4, comp)) = (int4_t){o[4], o[5], o[6], o[7]};
}

// a middle stage (1 <= S < NS - 1): `valid` inputs at in_off of its buffer -> 2 * valid entries at the start of the next buffer
template <class G, int S, bool FULL> __device__ __forceinline__ void wstage(int *lds, int lane, int in_off, int valid, int (&o)[8])
{
    constexpr int O = stage_order(S), K = O / 4, S2 = O / 2, R = 4;
    const int j = lane >> 1, comp = lane & 1, m0 = j * R;
    if (!FULL && m0 >= valid) return;
    // window x <-> u[m0 - O/2 + x]; the window's 4-dword groups alternate between the two swizzled bases (HIST + in_off - S2 is a
    // multiple of 8, so the group parity is that of j + x / 4)
    static_assert(S2 % 8 == 0 && HIST % 8 == 0 && WB % 8 == 0, "aligned 8-dword groups");
    const int *plane = lds + G::base(S) + comp * WSTR + HIST + in_off - S2;
    const int *plE = plane + wswz(m0, comp), *plO = plane + (wswz(m0 + 4, comp) - 4);
    int w[R + S2];
#pragma unroll
    for (int x = 0; x < R + S2; x += 4) {
        const int4_t v = *reinterpret_cast<const int4_t *>(((x >> 2) & 1 ? plO : plE) + x);
        w[x] = v.x; w[x + 1] = v.y; w[x + 2] = v.z; w[x + 3] = v.w;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int acc = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) acc = mad24(w[r + 1 + i] + w[r + S2 - i], tap(O, i), acc);
        o[2 * r] = w[r + K]; // u[m - O/4]
        o[2 * r + 1] = acc >> 13;
    }
    int *nx = lds + G::base(S + 1) + comp * WSTR + HIST + 2 * m0;
    *reinterpret_cast<int4_t *>(nx + wswz(0, comp)) = (int4_t){o[0], o[1], o[2], o[3]};
    *reinterpret_cast<int4_t *>(nx + wswz(4, comp)) = (int4_t){o[4], o[5], o[6], o[7]};
}

// the last stage: both components per lane, taps x 8 (the int16 result is the accumulator's high half), 2 x 16-byte stores
template <class G, int S, bool FULL> __device__ __forceinline__ void wstage_last(int *lds, int lane, int in_off, int valid, const IOut &oc)
{
    constexpr int O = stage_order(S), K = O / 4, S2 = O / 2, R = 4;
    const int m0 = lane * R;
    if (!FULL && (m0 >= valid || !oc.store)) return;
    int ev[2][R], od[2][R];
#pragma unroll
    for (int comp = 0; comp < 2; ++comp) {
        const int *plane = lds + G::base(S) + comp * WSTR + HIST + in_off - S2;
        const int *plE = plane + wswz(m0, comp), *plO = plane + (wswz(m0 + 4, comp) - 4);
        int w[R + S2];
#pragma unroll
        for (int x = 0; x < R + S2; x += 4) {
            const int4_t v = *reinterpret_cast<const int4_t *>(((x >> 2) & 1 ? plO : plE) + x);
            w[x] = v.x; w[x + 1] = v.y; w[x + 2] = v.z; w[x + 3] = v.w;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int acc = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) acc = mad24(w[r + 1 + i] + w[r + S2 - i], 8 * tap(O, i), acc);
            ev[comp][r] = w[r + K];
            od[comp][r] = acc; // (acc >> 13) & 0xffff == bits 16..31 of 8 * sum
        }
    }
    unsigned pk[2 * R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        pk[2 * r] = __builtin_amdgcn_perm((unsigned)ev[1][r], (unsigned)ev[0][r], 0x05040100u);     // (I lo16, Q lo16)
        pk[2 * r + 1] = __builtin_amdgcn_perm((unsigned)od[1][r], (unsigned)od[0][r], 0x07060302u); // (I hi16, Q hi16)
    }
    size_t idx = oc.out_pos + 2 * (size_t)m0; // chain output index
    if (oc.stuff64) idx = (idx >> 5) * 64 + (idx & 31);
    unsigned *dst = oc.out + idx;
    if (FULL || m0 + R <= valid) {
        *reinterpret_cast<uint4_t *>(dst) = (uint4_t){pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<uint4_t *>(dst + 4) = (uint4_t){pk[4], pk[5], pk[6], pk[7]};
        if (oc.stuff64) {
            *reinterpret_cast<uint4_t *>(dst + 32) = (uint4_t){0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4_t *>(dst + 36) = (uint4_t){0u, 0u, 0u, 0u};
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2 * R; ++q)
            if (m0 + q / 2 < valid) {
                dst[q] = pk[q];
                if (oc.stuff64) dst[32 + q] = 0u;
            }
    }
}

// history of stage S (S >= 1) after its n fresh entries were consumed: entries [n - 32, n) -> [0, 32) of both planes, one
// ds_read_b32 + one ds_write_b32 of the whole wave (the wave reads before it writes: source and destination may overlap).
// (Writing them from the producer's registers instead -- they are still there -- costs two ds_write_b128 per invocation, and an
// LDS store is priced by instruction, not by active lane: 26 cycles of the CU's LDS pipe against 6, and the LDS pipe is one
// of the three resources this kernel keeps ~50 % busy: tools/store_lds_mix.hip.)
template <class G, int S> __device__ __forceinline__ void whist(int *lds, int lane, int n)
{
    int *pl = lds + G::base(S) + (lane >> 5) * WSTR;
    const int e = lane & 31, q = lane >> 5;
    const int v = pl[wswz(n + e, q)]; // = fresh entry n - 32 + e (or history entry n + e when n < 32)
    wave_sync();
    pl[wswz(e, q)] = v;
}

// FULL: a whole block with its stores (valid = 128 per stage invocation, 256 in the last stage): every guard is a compile-time
// constant and the walk is straight-line code apart from the LDS-only history writes -- which is what lets hipcc count the
// block's eight stores behind the prefetch load exactly (s_waitcnt vmcnt(8) instead of a drain of the stores, see the loop)
template <class G, int S, bool FULL> __device__ __forceinline__ void wdescend(int *lds, int lane, int in_off, int valid_rt, IOut &oc)
{
    const int valid = FULL ? (S == G::NS - 1 ? WCAP : WB) : valid_rt;
    if constexpr (S == G::NS - 1) {
        wstage_last<G, S, FULL>(lds, lane, in_off, valid, oc);
        if (FULL || oc.store) oc.out_pos += 2 * (size_t)valid;
    } else {
        int o[8];
        if constexpr (S == 0) wstage0<G, FULL>(lds, lane, in_off, valid, o); // (stage 0: in_off = the block's index in its group)
        else wstage<G, S, FULL>(lds, lane, in_off, valid, o);
        wave_sync(); // the stage's outputs, written by all lanes, are read by other lanes next
        const int n = 2 * valid;
        if constexpr (S + 1 == G::NS - 1) {
            wdescend<G, S + 1, FULL>(lds, lane, 0, n, oc);
        } else if constexpr (FULL) {
            wdescend<G, S + 1, true>(lds, lane, 0, WB, oc);
            wdescend<G, S + 1, true>(lds, lane, WB, WB, oc);
        } else {
            wdescend<G, S + 1, false>(lds, lane, 0, n < WB ? n : WB, oc);
            if (n > WB) wdescend<G, S + 1, false>(lds, lane, WB, n - WB, oc);
        }
        wave_sync(); // (the consumers' reads of the old history stay in front of its rewrite)
        whist<G, S + 1>(lds, lane, n);
        wave_sync();
    }
}

// bank state <-> LDS (same record as K5: per stage 2 planes x 32 int32 entries; stage 0 lives packed here)
template <class G, int S = 0> __device__ __forceinline__ void wstate_fetch(int (&sv)[G::NS], int lane, const int32_t *st, bool zero)
{
    sv[S] = zero ? 0 : st[S * 2 * INT_HIST + lane];
    if constexpr (S + 1 < G::NS) wstate_fetch<G, S + 1>(sv, lane, st, zero);
}
template <class G, int S = 0> __device__ __forceinline__ void wstate_put(int *lds, int lane, const int (&sv)[G::NS])
{
    if constexpr (S == 0) reinterpret_cast<short *>(lds + (lane >> 5) * W0STR)[lane & 31] = (short)sv[S];
    else lds[G::base(S) + (lane >> 5) * WSTR + wswz(lane & 31, lane >> 5)] = sv[S];
    if constexpr (S + 1 < G::NS) wstate_put<G, S + 1>(lds, lane, sv);
}
template <class G, int S = 0> __device__ __forceinline__ void wstate_store(const int *lds, int lane, int32_t *st)
{
    if constexpr (S == 0) st[lane] = (int)reinterpret_cast<const short *>(lds + (lane >> 5) * W0STR)[lane & 31];
    else st[S * 2 * INT_HIST + lane] = lds[G::base(S) + (lane >> 5) * WSTR + wswz(lane & 31, lane >> 5)];
    if constexpr (S + 1 < G::NS) wstate_store<G, S + 1>(lds, lane, st);
}

// Where a wave's input comes from.  Contiguous: the stream is a.in.  Gather (the Tx pipe without the decoder's copy, InterpArgs::gmap):
// sample g of the stream is dword `col` of block `blk` of frame `f`, (f, blk, col) = (g / 16129, 1 + g % 16129 / 127, g % 127), and the
// block lies where the decoder's map says -- in the received frames (arrival order) or among the restored blocks.
template <bool GATHER> struct WSrc;
template <> struct WSrc<false> {
    const unsigned *in;
    __device__ __forceinline__ WSrc(const InterpArgs &a, int stream) : in(reinterpret_cast<const unsigned *>(a.in) + (size_t)stream * a.in_stride) {}
    __device__ __forceinline__ bool aligned16() const { return (reinterpret_cast<uintptr_t>(in) & 15u) == 0; }
    __device__ __forceinline__ unsigned ld1(size_t g) const { return SDRHIP_STREAM_LOAD(in + g); }
    // the lane's 4 samples of pair p of the segment: one 16-byte load
    __device__ __forceinline__ void seek(size_t) {}
    __device__ __forceinline__ uint4_t ld4(size_t g) { return SDRHIP_STREAM_LOAD(reinterpret_cast<const uint4_t *>(in + g)); }
};
template <> struct WSrc<true> {
    const unsigned *gmap;
    const uint8_t *grx, *grest;
    unsigned fbase, flast;   // first / last frame (absolute index) of the stream
    unsigned f, blk, col;    // where the lane's running position (seek / ld4) stands
    __device__ __forceinline__ WSrc(const InterpArgs &a, int stream)
        : gmap(a.gmap), grx(a.grx), grest(a.grest), fbase((unsigned)stream * (unsigned)a.gframes), flast((unsigned)stream * (unsigned)a.gframes + (unsigned)a.gframes - 1u), f(0), blk(1), col(0) {}
    __device__ __forceinline__ bool aligned16() const { return true; }
    __device__ __forceinline__ const unsigned *block(unsigned ff, unsigned bb) const
    {
        const unsigned code = gmap[(size_t)ff * 128u + bb];
        const uint8_t *p = (code >> 31) ? grest + (size_t)(code & 0x7fffffffu) * 508u : grx + (size_t)code * 512u + 4u;
        return reinterpret_cast<const unsigned *>(p);
    }
    __device__ __forceinline__ unsigned ld1(size_t g) const
    {
        const unsigned gi = (unsigned)g, fr = gi / 16129u, r = gi - fr * 16129u, b = r / 127u;
        return SDRHIP_STREAM_LOAD(block(fbase + fr, b + 1u) + (r - b * 127u));
    }
    __device__ __forceinline__ void seek(size_t g)
    {
        const unsigned gi = (unsigned)g, fr = gi / 16129u, r = gi - fr * 16129u, b = r / 127u;
        f = fbase + fr; blk = b + 1u; col = r - b * 127u;
    }
    // the 4 samples at the running position (they may straddle a block, and with it a frame), then the position moves on by 256
    // samples = two blocks and two samples (a pair of K5w's input blocks further)
    __device__ __forceinline__ uint4_t ld4(size_t)
    {
        const unsigned *pa = block(f, blk);
        unsigned nb = blk + 1u, nf = f;
        if (nb > 127u) { nb = 1u; nf = f + 1u; }
        if (nf > flast) nf = flast; // (past the stream's last frame: never selected -- the segment's samples end in front of it)
        const unsigned *pb = block(nf, nb) - 127; // (indexed with col + k >= 127)
        uint4_t v;
        v.x = SDRHIP_STREAM_LOAD(pa + col);
        v.y = SDRHIP_STREAM_LOAD((col + 1u < 127u ? pa : pb) + col + 1u);
        v.z = SDRHIP_STREAM_LOAD((col + 2u < 127u ? pa : pb) + col + 2u);
        v.w = SDRHIP_STREAM_LOAD((col + 3u < 127u ? pa : pb) + col + 3u);
        col += 2u; blk += 2u;
        if (col >= 127u) { col -= 127u; blk += 1u; }
        if (blk > 127u) { blk -= 127u; f += 1u; }
        return v;
    }
};

// one segment (seg of a.nseg, a.nsub_per_seg blocks of 128 inputs each) of one stream, on one wave; L >= 2
//
// WHERE THE INPUT LOADS GO decides this kernel (tools/store_load_mix.hip, tools/experiments_r04/): a streaming store pattern
// that runs 5.7 TB/s on its own drops to 3.5 TB/s when 6 % of the bytes are loads issued 512 B at a time between the stores --
// whether the loads are awaited or not, hit in the cache or not, even when OTHER waves issue them.  Clustered, they are almost
// free: a wave's whole input as 1 KiB loads back to back at its start costs 10 %.  So a wave loads the input of its whole
// segment (<= 16 blocks = 8 KiB: eight global_load_dwordx4) in front of everything else, parks it in 32 accumulation registers
// (nothing else uses them, so nothing moves them) and issues nothing but stores from then on; a pair of blocks at a time comes
// back through v_accvgpr_read, is de-interleaved into packed I / Q pairs and staged in LDS.
template <int L, bool GATHER = false> __device__ __forceinline__ void interp_wave_segment(const InterpArgs &a, int seg, int stream, int *lds)
{
    constexpr int NS = (L == 6) ? 5 : L;
    static_assert(NS >= 2, "interpolate2 has a single stage: K5");
    using G = WGeo<NS>;
    const int lane = threadIdx.x & 63;
    WSrc<GATHER> src(a, stream);
    const size_t seg_len = (size_t)a.nsub_per_seg * WB;
    const size_t seg_start = (size_t)seg * seg_len;
    size_t seg_end = seg_start + seg_len;
    if (seg_end > a.n_in) seg_end = a.n_in;

    unsigned *p0 = reinterpret_cast<unsigned *>(lds);
    // ---- the segment's whole blocks, pair by pair, into AGPRs (pairs beyond the segment: not loaded, never read)
    const bool al16 = src.aligned16();
    const int npairs = al16 ? (int)((seg_end - seg_start) / (2 * WB) < (size_t)WPAIRS ? (seg_end - seg_start) / (2 * WB) : (size_t)WPAIRS) : 0;
    unsigned A[4 * WPAIRS];
    const int32_t *stc = a.state_cur + (size_t)stream * INT_STATE_WORDS;
    uint2_t wv = (uint2_t){0u, 0u};
    int sv[NS];
    {
        uint4_t v[WPAIRS];
        src.seek(seg_start + 4 * (size_t)lane);
#pragma unroll
        for (int p = 0; p < WPAIRS; ++p) { // all of them back to back: ONE cluster of reads per wave
            v[p] = (uint4_t){0u, 0u, 0u, 0u};
            if (p < npairs) v[p] = src.ld4(seg_start + (size_t)p * 2 * WB + 4 * lane);
        }
        // the warm-up's samples and the bank state: behind the cluster and in flight with it (three round trips in a row --
        // cluster, state, warm-up -- at the start of every wave before; tools/experiments_r04 batch 24)
        if (seg != 0) {
            const size_t wg = (seg_start - WWARM) + 2 * (size_t)lane;
            if (2 * lane < WWARM) wv.x = src.ld1(wg);
            if (2 * lane + 1 < WWARM) wv.y = src.ld1(wg + 1);
        }
        wstate_fetch<G>(sv, lane, stc, seg != 0);
#pragma unroll
        for (int p = 0; p < WPAIRS; ++p)
            asm volatile("v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                         : "=a"(A[4 * p]), "=a"(A[4 * p + 1]), "=a"(A[4 * p + 2]), "=a"(A[4 * p + 3]) : "v"(v[p].x), "v"(v[p].y), "v"(v[p].z), "v"(v[p].w));
    }

    wstate_put<G>(lds, lane, sv);
    wave_sync();

    IOut oc;
    oc.out = reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride;
    oc.stuff64 = (L == 6);
    oc.out_pos = seg_start << NS;
    oc.store = true;

    // history of stage 0 after n inputs were consumed: entry by entry (n may be odd at the ragged end of a call)
    auto hist0 = [&](int n) {
        short *pl = reinterpret_cast<short *>(p0 + (lane >> 5) * W0STR);
        const int e = lane & 31;
        wave_sync();
        const short v = pl[n + e];
        wave_sync();
        pl[e] = v;
        wave_sync();
    };
    // ---- one block at a time with its own loads and every guard: the warm-up of a segment (in front of every store of the
    // wave), and what the pairs do not cover -- an odd block, a ragged end, an unaligned input: the last segment of a call
    auto single_v = [&](uint2_t v, int cnt, bool store) {
        p0[W0HIST + lane] = __builtin_amdgcn_perm(v.y, v.x, 0x05040100u);
        p0[W0STR + W0HIST + lane] = __builtin_amdgcn_perm(v.y, v.x, 0x07060302u);
        wave_sync();
        oc.store = store;
        wdescend<G, 0, false>(lds, lane, 0, cnt, oc);
        hist0(cnt);
    };
    auto single = [&](size_t pos, int cnt, bool store) {
        const int m = 2 * lane;
        uint2_t v = (uint2_t){0u, 0u};
        if (m < cnt) v.x = src.ld1(pos + m);
        if (m + 1 < cnt) v.y = src.ld1(pos + m + 1);
        single_v(v, cnt, store);
    };

    size_t pos = seg_start;
    if (seg != 0) single_v(wv, WWARM, false); // histories of the slice from the 44 inputs in front of it, stores suppressed
    oc.store = true;
    for (int p = 0; p < npairs; ++p) {
        // the pair's 256 samples: lane t holds samples 4t .. 4t+3 -> packed dwords 2t, 2t+1 of both planes
        uint4_t v;
        bool got = false;
#pragma unroll
        for (int q = 0; q < WPAIRS; ++q)
            if (!got && p == q) { // (wave-uniform: a scalar branch per candidate)
                asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                             : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w) : "a"(A[4 * q]), "a"(A[4 * q + 1]), "a"(A[4 * q + 2]), "a"(A[4 * q + 3]));
                got = true;
            }
        const uint2_t pi = (uint2_t){__builtin_amdgcn_perm(v.y, v.x, 0x05040100u), __builtin_amdgcn_perm(v.w, v.z, 0x05040100u)};
        const uint2_t pq = (uint2_t){__builtin_amdgcn_perm(v.y, v.x, 0x07060302u), __builtin_amdgcn_perm(v.w, v.z, 0x07060302u)};
        *reinterpret_cast<uint2_t *>(p0 + W0HIST + 2 * lane) = pi;
        *reinterpret_cast<uint2_t *>(p0 + W0STR + W0HIST + 2 * lane) = pq;
        wave_sync();
        wdescend<G, 0, true>(lds, lane, 0, WB, oc);
        wdescend<G, 0, true>(lds, lane, 1, WB, oc);
        // history of stage 0: the pair's last 32 samples, straight from the registers of lanes 56 .. 63
        wave_sync();
        if (lane >= 56) {
            *reinterpret_cast<uint2_t *>(p0 + 2 * (lane - 56)) = pi;
            *reinterpret_cast<uint2_t *>(p0 + W0STR + 2 * (lane - 56)) = pq;
        }
        wave_sync();
        pos += 2 * WB;
    }
    while (pos < seg_end) {
        const int cnt = (int)((seg_end - pos) < (size_t)WB ? (seg_end - pos) : (size_t)WB);
        single(pos, cnt, true);
        pos += cnt;
    }
    if (seg == a.nseg - 1) {
        int32_t *stn = a.state_next + (size_t)stream * INT_STATE_WORDS;
        wave_sync();
        wstate_store<G>(lds, lane, stn);
        // the stages this ratio does not use travel as they are (loads first, then the stores: as a copy loop every iteration was a
        // global round trip of its own, at the end of the wave that ends the launch)
        constexpr int REST = (INT_STAGES - NS) * 2 * INT_HIST;
        static_assert(REST % WNT == 0, "whole waves");
        int32_t keep[REST / WNT > 0 ? REST / WNT : 1];
#pragma unroll
        for (int k = 0; k < REST / WNT; ++k) keep[k] = stc[NS * 2 * INT_HIST + k * WNT + lane];
#pragma unroll
        for (int k = 0; k < REST / WNT; ++k) stn[NS * 2 * INT_HIST + k * WNT + lane] = keep[k];
    }
}

} // namespace
} // namespace sdrhip
#endif
