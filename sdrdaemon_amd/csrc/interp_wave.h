// interp_wave.h -- K5w: the interpolator cascade as barrier-free, wave-private pipelines.
//
// Same arithmetic as interp_body.h (Interpolators::interpolate{4..64}_cen over IntHalfbandFilterEO1/DB<64|32|16>::myInterpolate,
// Interpolators.cpp:47-606, IntHalfbandFilterEO1.h:44-65,149-168), different machine mapping:
//  * a workgroup IS one wave (64 lanes) that owns a time slice of one stream and its own 7.5 KB of LDS: no s_barrier anywhere
//    (K5 needs 20 workgroup barriers per macro-cycle); the stage buffers are handed from stage to stage by the wave itself, LDS
//    operations of one wave execute in order, so a write followed by a read needs no synchronisation at all;
//  * blocks of 128 inputs, walked depth first: stage s takes 128 inputs per invocation (lanes 2j / 2j+1 = I / Q of inputs
//    4j .. 4j+3), the last stage 256 (both components per lane: packs int16 I/Q and stores 2 x 16 bytes per lane);
//  * stage 0 (order 64) reads its input as PACKED int16 pairs and runs on v_dot2_i32_i16: 16 dot products + 4.25 v_alignbit per
//    output instead of 16 adds + 16 multiply-adds (the raw samples are int16; every later stage sees 19-bit values);
//  * the last stage multiplies by 8 x tap: (acc >> 13) & 0xffff is then the HIGH half of the accumulator, and one v_perm_b32 both
//    shifts and packs I/Q (the truncation to int16 drops everything above bit 28 anyway);
//  * the 32-entry history of a stage buffer is rewritten from the PRODUCER's registers after the consumer is done (two
//    ds_write_b128 on four lane pairs) instead of an LDS read -> wait -> write round trip per invocation.
#ifndef SDRHIP_INTERP_WAVE_H
#define SDRHIP_INTERP_WAVE_H
#include "interp_body.h"

namespace sdrhip {
namespace {

constexpr int WNT = 64;          // one wave
constexpr int WB = 128;          // inputs per block
constexpr int WCAP = 256;        // fresh entries a stage buffer (s >= 1) holds
constexpr int WSTR = HIST + WCAP; // 288 dwords = 72 sixteen-byte slots = 8 mod 16: I / Q lane pairs of a ds_read_b128 group hit distinct banks
constexpr int W0HIST = 16;       // packed stage-0 plane: 16 dwords (32 entries) of history + 64 fresh
constexpr int W0STR = 96;        // ... padded to 48 eight-byte slots = 16 mod 32: the I / Q lanes of a ds_read_b64 group hit distinct banks

typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

template <int NS_> struct WGeo {
    static constexpr int NS = NS_;
    static constexpr int base(int s) { return s == 0 ? 0 : 2 * W0STR + (s - 1) * 2 * WSTR; }
    static constexpr int ldsDw = 2 * W0STR + (NS - 1) * 2 * WSTR;
};

// taps of the order-64 stage over delays 0..31 (symmetric), packed for v_dot2_i32_i16 on (older, newer) sample pairs
__host__ __device__ constexpr int h64(int d) { return d < 16 ? T64[d] : T64[31 - d]; }
__host__ __device__ constexpr unsigned tap_pair(int e) { return ((unsigned)h64(2 * e + 1) & 0xffffu) | ((unsigned)h64(2 * e) << 16); }

__device__ __forceinline__ int wdot2(unsigned a, unsigned taps, int acc)
{
    typedef short short2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, taps), acc, false);
}

// stage 0 (order 64) on packed int16 planes: `valid` inputs of the block -> 2 * valid entries at the start of stage 1's buffer
template <class G> __device__ __forceinline__ void wstage0(int *lds, int lane, int valid, int (&o)[8])
{
    const int j = lane >> 1, comp = lane & 1, m0 = 4 * j;
    if (m0 >= valid) return;
    // window dword t <-> entries u[m0 - 32 + 2t], u[m0 - 32 + 2t + 1]
    const unsigned *pl = reinterpret_cast<const unsigned *>(lds) + comp * W0STR + 2 * j;
    unsigned W[18];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint2_t v = *reinterpret_cast<const uint2_t *>(pl + 2 * t);
        W[2 * t] = v.x; W[2 * t + 1] = v.y;
    }
    unsigned A[18]; // the odd alignment: A[t] <-> entries 2t - 1, 2t of the window
#pragma unroll
    for (int t = 1; t < 18; ++t) A[t] = __builtin_amdgcn_alignbit(W[t], W[t - 1], 16);
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc[0] = wdot2(A[16 - e], tap_pair(e), acc[0]);
        acc[1] = wdot2(W[16 - e], tap_pair(e), acc[1]);
        acc[2] = wdot2(A[17 - e], tap_pair(e), acc[2]);
        acc[3] = wdot2(W[17 - e], tap_pair(e), acc[3]);
    }
    // v[2m] = u[m - 16]: entries 16 .. 19 of the window
    o[0] = (int)(short)(W[8] & 0xffffu); o[2] = (int)W[8] >> 16; o[4] = (int)(short)(W[9] & 0xffffu); o[6] = (int)W[9] >> 16;
    o[1] = acc[0] >> 13; o[3] = acc[1] >> 13; o[5] = acc[2] >> 13; o[7] = acc[3] >> 13;
    int *nx = lds + G::base(1) + comp * WSTR + HIST + 2 * m0;
    *reinterpret_cast<int4_t *>(nx) = (int4_t){o[0], o[1], o[2], o[3]};
    *reinterpret_cast<int4_t *>(nx + 4) = (int4_t){o[4], o[5], o[6], o[7]};
}

// a middle stage (1 <= S < NS - 1): `valid` inputs at in_off of its buffer -> 2 * valid entries at the start of the next buffer
template <class G, int S> __device__ __forceinline__ void wstage(int *lds, int lane, int in_off, int valid, int (&o)[8])
{
    constexpr int O = stage_order(S), K = O / 4, S2 = O / 2, R = 4;
    const int j = lane >> 1, comp = lane & 1, m0 = j * R;
    if (m0 >= valid) return;
    const int *pl = lds + G::base(S) + comp * WSTR + HIST + in_off + m0 - S2; // window x <-> u[m0 - O/2 + x]
    int w[R + S2];
#pragma unroll
    for (int x = 0; x < R + S2; x += 4) {
        const int4_t v = *reinterpret_cast<const int4_t *>(pl + x);
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
    *reinterpret_cast<int4_t *>(nx) = (int4_t){o[0], o[1], o[2], o[3]};
    *reinterpret_cast<int4_t *>(nx + 4) = (int4_t){o[4], o[5], o[6], o[7]};
}

// the last stage: both components per lane, taps x 8 (the int16 result is the accumulator's high half), 2 x 16-byte stores
template <class G, int S> __device__ __forceinline__ void wstage_last(int *lds, int lane, int in_off, int valid, const IOut &oc)
{
    constexpr int O = stage_order(S), K = O / 4, S2 = O / 2, R = 4;
    const int m0 = lane * R;
    if (m0 >= valid || !oc.store) return;
    int ev[2][R], od[2][R];
#pragma unroll
    for (int comp = 0; comp < 2; ++comp) {
        const int *pl = lds + G::base(S) + comp * WSTR + HIST + in_off + m0 - S2;
        int w[R + S2];
#pragma unroll
        for (int x = 0; x < R + S2; x += 4) {
            const int4_t v = *reinterpret_cast<const int4_t *>(pl + x);
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
    if (m0 + R <= valid) {
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

// history of stage S (S >= 1) after its n fresh entries were consumed: entries [n - 32, n) -> [0, 32).  The producer's lane
// pairs still hold them (o[0..7] = entries 8j .. 8j+7): straight from the registers when whole lanes line up, through LDS
// otherwise (one instruction each: the whole wave reads before it writes, source and destination may overlap)
template <class G, int S> __device__ __forceinline__ void whist(int *lds, int lane, int n, const int (&o)[8])
{
    if (n >= HIST && (n & 7) == 0) {
        const int j = lane >> 1, comp = lane & 1;
        const int e0 = 8 * j - (n - HIST);
        if (e0 >= 0 && 8 * j < n) {
            int *pl = lds + G::base(S) + comp * WSTR + e0;
            *reinterpret_cast<int4_t *>(pl) = (int4_t){o[0], o[1], o[2], o[3]};
            *reinterpret_cast<int4_t *>(pl + 4) = (int4_t){o[4], o[5], o[6], o[7]};
        }
    } else {
        int *pl = lds + G::base(S) + (lane >> 5) * WSTR;
        const int e = lane & 31;
        const int v = pl[n + e]; // = fresh entry n - 32 + e (or history entry n + e when n < 32)
        pl[e] = v;
    }
}

// depth-first walk: stage S consumes `valid` inputs at in_off of its buffer
template <class G, int S> __device__ __forceinline__ void wdescend(int *lds, int lane, int in_off, int valid, IOut &oc)
{
    if constexpr (S == G::NS - 1) {
        wstage_last<G, S>(lds, lane, in_off, valid, oc);
        if (oc.store) oc.out_pos += 2 * (size_t)valid;
    } else {
        int o[8]; // (read back by whist only on the lanes that computed them)
        if constexpr (S == 0) wstage0<G>(lds, lane, valid, o);
        else wstage<G, S>(lds, lane, in_off, valid, o);
        const int n = 2 * valid;
        if constexpr (S + 1 == G::NS - 1) {
            wdescend<G, S + 1>(lds, lane, 0, n, oc);
        } else {
            wdescend<G, S + 1>(lds, lane, 0, n < WB ? n : WB, oc);
            if (n > WB) wdescend<G, S + 1>(lds, lane, WB, n - WB, oc);
        }
        whist<G, S + 1>(lds, lane, n, o);
    }
}

// bank state <-> LDS (same record as K5: per stage 2 planes x 32 int32 entries; stage 0 lives packed here)
template <class G, int S = 0> __device__ __forceinline__ void wstate_load(int *lds, int lane, const int32_t *st, bool zero)
{
    const int v = zero ? 0 : st[S * 2 * INT_HIST + lane];
    if constexpr (S == 0) reinterpret_cast<short *>(lds + (lane >> 5) * W0STR)[lane & 31] = (short)v;
    else lds[G::base(S) + (lane >> 5) * WSTR + (lane & 31)] = v;
    if constexpr (S + 1 < G::NS) wstate_load<G, S + 1>(lds, lane, st, zero);
}
template <class G, int S = 0> __device__ __forceinline__ void wstate_store(const int *lds, int lane, int32_t *st)
{
    if constexpr (S == 0) st[lane] = (int)reinterpret_cast<const short *>(lds + (lane >> 5) * W0STR)[lane & 31];
    else st[S * 2 * INT_HIST + lane] = lds[G::base(S) + (lane >> 5) * WSTR + (lane & 31)];
    if constexpr (S + 1 < G::NS) wstate_store<G, S + 1>(lds, lane, st);
}

// one segment (seg of a.nseg, a.nsub_per_seg blocks of 128 inputs each) of one stream, on one wave; L >= 2
template <int L> __device__ __forceinline__ void interp_wave_segment(const InterpArgs &a, int seg, int stream, int *lds)
{
    constexpr int NS = (L == 6) ? 5 : L;
    static_assert(NS >= 2, "interpolate2 has a single stage: K5");
    using G = WGeo<NS>;
    const int lane = threadIdx.x;
    const unsigned *in = reinterpret_cast<const unsigned *>(a.in) + (size_t)stream * a.in_stride;
    const size_t seg_len = (size_t)a.nsub_per_seg * WB;
    const size_t seg_start = (size_t)seg * seg_len;
    size_t seg_end = seg_start + seg_len;
    if (seg_end > a.n_in) seg_end = a.n_in;

    const int32_t *stc = a.state_cur + (size_t)stream * INT_STATE_WORDS;
    wstate_load<G>(lds, lane, stc, seg != 0);

    IOut oc;
    oc.out = reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride;
    oc.stuff64 = (L == 6);
    oc.out_pos = seg_start << NS;

    bool warm = (seg != 0);
    size_t pos = warm ? seg_start - WARM : 0;
    const bool al8 = (reinterpret_cast<uintptr_t>(in) & 7u) == 0; // (pos is even)
    auto issue = [&](size_t p, int cnt) -> uint2_t {
        const int m = 2 * lane;
        uint2_t v = (uint2_t){0u, 0u};
        if (m + 1 < cnt && al8) v = SDRHIP_STREAM_LOAD(reinterpret_cast<const uint2_t *>(in + p + m));
        else {
            if (m < cnt) v.x = SDRHIP_STREAM_LOAD(in + p + m);
            if (m + 1 < cnt) v.y = SDRHIP_STREAM_LOAD(in + p + m + 1);
        }
        return v;
    };
    int cnt = warm ? WARM : (int)((seg_end - pos) < (size_t)WB ? (seg_end - pos) : (size_t)WB);
    uint2_t ld = issue(pos, cnt);
    unsigned *p0 = reinterpret_cast<unsigned *>(lds);
    while (true) {
        // de-interleave: packed I pairs and Q pairs (dword `lane` of the block's fresh part)
        const unsigned pi = __builtin_amdgcn_perm(ld.y, ld.x, 0x05040100u), pq = __builtin_amdgcn_perm(ld.y, ld.x, 0x07060302u);
        p0[W0HIST + lane] = pi;
        p0[W0STR + W0HIST + lane] = pq;
        const size_t next_pos = pos + cnt;
        const bool more = next_pos < seg_end;
        int next_cnt = 0;
        if (more) {
            next_cnt = (int)((seg_end - next_pos) < (size_t)WB ? (seg_end - next_pos) : (size_t)WB);
            ld = issue(next_pos, next_cnt); // in flight while this block computes
        }
        oc.store = !warm;
        wdescend<G, 0>(lds, lane, 0, cnt, oc);
        // history of stage 0: the last 32 inputs
        if (cnt == WB) {
            if (lane >= 48) { p0[lane - 48] = pi; p0[W0STR + lane - 48] = pq; }
        } else {
            short *pl = reinterpret_cast<short *>(p0 + (lane >> 5) * W0STR);
            const int e = lane & 31;
            const short v = pl[cnt + e];
            pl[e] = v;
        }
        if (!more) break;
        pos = next_pos;
        cnt = next_cnt;
        warm = false;
    }
    if (seg == a.nseg - 1) {
        int32_t *stn = a.state_next + (size_t)stream * INT_STATE_WORDS;
        wstate_store<G>(lds, lane, stn);
        for (int i = NS * 2 * INT_HIST + lane; i < INT_STAGES * 2 * INT_HIST; i += WNT) stn[i] = stc[i];
    }
}

} // namespace
} // namespace sdrhip
#endif
