// interp_body.h -- the VALU / LDS interpolator cascade (DESIGN.md "K5") as device code, included by interp_kernels.hip.
#ifndef SDRHIP_INTERP_BODY_H
#define SDRHIP_INTERP_BODY_H
#include "sdrhip_internal.h"

#include <cstring>

namespace sdrhip {
namespace {

constexpr int NT = 256;
constexpr int HIST = 32;
constexpr int WARM = 64;
constexpr int MC = 512; // inputs per macro-cycle (1024 when the cascade has a single stage)

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

// HBFIRFilterTraits<64|32|16>::hbCoeffs, HBFilterTraits.cpp:210-228, 62-72, 25-31
constexpr int T64[16] = {-7, 11, -20, 32, -49, 71, -101, 140, -190, 256, -345, 469, -656, 978, -1698, 5201};
constexpr int T32[8] = {-30, 63, -135, 261, -469, 830, -1605, 5176};
constexpr int T16[4] = {-85, 380, -1246, 5041};

__host__ __device__ constexpr int stage_order(int s) { return s == 0 ? 64 : (s == 1 ? 32 : 16); }
__host__ __device__ constexpr int tap(int order, int i) { return order == 64 ? T64[i] : (order == 32 ? T32[i] : T16[i]); }

__device__ __forceinline__ int mad24(int a, int t, int acc)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(t), "v"(acc));
    return d;
}

template <int NS_> struct IGeo {
    static constexpr int NS = NS_;
    static constexpr int mc = (NS == 1) ? 1024 : MC;                                  // inputs per macro-cycle
    static constexpr int cap(int s) { return s == 0 ? mc : 1024; }                     // fresh entries a stage buffer holds
    static constexpr int stride(int s) { return HIST + cap(s); }                       // 544 / 1056 dwords: 8 mod 16 slots
    static constexpr int base(int s) { return s == 0 ? 0 : base(s - 1) + 2 * stride(s - 1); }
    static constexpr int ldsDw = base(NS);
};

struct IOut {
    unsigned *out;
    size_t out_pos; // chain output index of the next final output of this stream
    bool store;
    bool stuff64;   // interpolate64_cen layout
};

// one invocation of a non-last stage: `valid` inputs at buffer offset in_off -> 2 * valid entries
// at the start of the next stage's buffer
template <class G, int S> __device__ __forceinline__ void istage(int *lds, int tid, int in_off, int valid)
{
    constexpr int O = stage_order(S), K = O / 4, S2 = O / 2, R = 4;
    const int j = tid >> 1, comp = tid & 1;
    const int m0 = j * R;
    if (m0 >= valid) return;
    const int *pl = lds + G::base(S) + comp * G::stride(S) + HIST + in_off + m0 - S2; // window x <-> u[m0 - O/2 + x]
    int w[R + S2];
#pragma unroll
    for (int x = 0; x < R + S2; x += 4) {
        int4_t v = *reinterpret_cast<const int4_t *>(pl + x);
        w[x] = v.x; w[x + 1] = v.y; w[x + 2] = v.z; w[x + 3] = v.w;
    }
    int o[2 * R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int acc = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) acc = mad24(w[r + 1 + i] + w[r + S2 - i], tap(O, i), acc);
        o[2 * r] = w[r + K]; // u[m - O/4]
        o[2 * r + 1] = acc >> 13;
    }
    int *nx = lds + G::base(S + 1) + comp * G::stride(S + 1) + HIST + 2 * m0;
    *reinterpret_cast<int4_t *>(nx) = (int4_t){o[0], o[1], o[2], o[3]};
    *reinterpret_cast<int4_t *>(nx + 4) = (int4_t){o[4], o[5], o[6], o[7]};
}

// the last stage: both components per thread, int16 packing, 2 x 16-byte stores
template <class G, int S> __device__ __forceinline__ void istage_last(int *lds, int tid, int in_off, int valid, const IOut &oc)
{
    constexpr int O = stage_order(S), K = O / 4, S2 = O / 2, R = 4;
    const int m0 = tid * R;
    if (m0 >= valid || !oc.store) return;
    int o[2][2 * R];
#pragma unroll
    for (int comp = 0; comp < 2; ++comp) {
        const int *pl = lds + G::base(S) + comp * G::stride(S) + HIST + in_off + m0 - S2;
        int w[R + S2];
#pragma unroll
        for (int x = 0; x < R + S2; x += 4) {
            int4_t v = *reinterpret_cast<const int4_t *>(pl + x);
            w[x] = v.x; w[x + 1] = v.y; w[x + 2] = v.z; w[x + 3] = v.w;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int acc = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) acc = mad24(w[r + 1 + i] + w[r + S2 - i], tap(O, i), acc);
            o[comp][2 * r] = w[r + K];
            o[comp][2 * r + 1] = acc >> 13;
        }
    }
    unsigned pk[2 * R];
#pragma unroll
    for (int q = 0; q < 2 * R; ++q) pk[q] = __builtin_amdgcn_perm((unsigned)o[1][q], (unsigned)o[0][q], 0x05040100u); // (I lo16, Q lo16)
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

// history of stage S: entries [consumed, consumed + 32) -> [0, 32) of both planes (one wave:
// the whole wave reads before it writes, source and destination may overlap)
template <class G, int S> __device__ __forceinline__ void slide(int *lds, int tid, int consumed)
{
    if (tid >= 64) return;
    int *pl = lds + G::base(S) + (tid >> 5) * G::stride(S);
    const int e = tid & 31;
    const int v = pl[HIST + consumed - HIST + e];
    pl[e] = v;
}

// depth-first walk: stage S consumes `valid` inputs at in_off of its buffer
template <class G, int S> __device__ __forceinline__ void descend(int *lds, int tid, int in_off, int valid, IOut &oc)
{
    if constexpr (S == G::NS - 1) {
        istage_last<G, S>(lds, tid, in_off, valid, oc);
        if (oc.store) oc.out_pos += 2 * (size_t)valid;
        __syncthreads();
    } else {
        istage<G, S>(lds, tid, in_off, valid);
        __syncthreads();
        const int n = 2 * valid;
        if constexpr (S + 1 == G::NS - 1) {
            descend<G, S + 1>(lds, tid, 0, n, oc);
        } else {
            descend<G, S + 1>(lds, tid, 0, n < MC ? n : MC, oc);
            if (n > MC) descend<G, S + 1>(lds, tid, MC, n - MC, oc);
        }
        slide<G, S + 1>(lds, tid, n);
        __syncthreads();
    }
}

template <class G, int S = 0> __device__ __forceinline__ void state_load(int *lds, int tid, const int32_t *st, bool zero)
{
    if (tid < 64) lds[G::base(S) + (tid >> 5) * G::stride(S) + (tid & 31)] = zero ? 0 : st[S * 2 * INT_HIST + tid];
    if constexpr (S + 1 < G::NS) state_load<G, S + 1>(lds, tid, st, zero);
}
template <class G, int S = 0> __device__ __forceinline__ void state_store(const int *lds, int tid, int32_t *st)
{
    if (tid < 64) st[S * 2 * INT_HIST + tid] = lds[G::base(S) + (tid >> 5) * G::stride(S) + (tid & 31)];
    if constexpr (S + 1 < G::NS) state_store<G, S + 1>(lds, tid, st);
}

// L = log2 interpolation (6 = the reference's 5-stage + zero stuffing variant)
// one segment (seg of a.nseg, a.nsub_per_seg macro-cycles each) of one stream; lds: IGeo<NS>::ldsDw dwords
template <int L> __device__ __forceinline__ void interp_segment(const InterpArgs &a, int seg, int stream, int *lds)
{
    constexpr int NS = (L == 6) ? 5 : L;
    using G = IGeo<NS>;
    constexpr int CI = G::mc;
    static_assert(G::ldsDw * 4 <= 64 * 1024, "LDS budget");

    const int tid = threadIdx.x;
    const unsigned *in = reinterpret_cast<const unsigned *>(a.in) + (size_t)stream * a.in_stride;
    const size_t seg_len = (size_t)a.nsub_per_seg * CI;
    const size_t seg_start = (size_t)seg * seg_len;
    size_t seg_end = seg_start + seg_len;
    if (seg_end > a.n_in) seg_end = a.n_in;

    const int32_t *stc = a.state_cur + (size_t)stream * INT_STATE_WORDS;
    state_load<G>(lds, tid, stc, seg != 0);

    IOut oc;
    oc.out = reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride;
    oc.stuff64 = (L == 6);
    oc.out_pos = seg_start << NS;

    bool warm = (seg != 0);
    size_t pos = warm ? seg_start - WARM : 0;
    unsigned ldv[CI / NT];
    auto issue = [&](size_t p, int cnt) {
#pragma unroll
        for (int n = 0; n < CI / NT; ++n) {
            const int m = tid + n * NT;
            ldv[n] = (m < cnt) ? SDRHIP_STREAM_LOAD(in + p + m) : 0u;
        }
    };
    int cnt = warm ? WARM : (int)((seg_end - pos) < (size_t)CI ? (seg_end - pos) : (size_t)CI);
    issue(pos, cnt);
    __syncthreads();
    while (true) {
#pragma unroll
        for (int n = 0; n < CI / NT; ++n) {
            const int m = tid + n * NT;
            lds[G::base(0) + HIST + m] = (int)(short)(ldv[n] & 0xffffu);
            lds[G::base(0) + G::stride(0) + HIST + m] = (int)ldv[n] >> 16;
        }
        const size_t next_pos = pos + cnt;
        const bool more = next_pos < seg_end;
        int next_cnt = 0;
        if (more) {
            next_cnt = (int)((seg_end - next_pos) < (size_t)CI ? (seg_end - next_pos) : (size_t)CI);
            issue(next_pos, next_cnt); // in flight while this macro-cycle computes
        }
        __syncthreads();
        oc.store = !warm;
        descend<G, 0>(lds, tid, 0, cnt, oc);
        slide<G, 0>(lds, tid, cnt);
        __syncthreads();
        if (!more) break;
        pos = next_pos;
        cnt = next_cnt;
        warm = false;
    }
    if (seg == a.nseg - 1) {
        int32_t *stn = a.state_next + (size_t)stream * INT_STATE_WORDS;
        state_store<G>(lds, tid, stn);
        for (int i = NS * 2 * INT_HIST + tid; i < INT_STAGES * 2 * INT_HIST; i += NT) stn[i] = stc[i];
    }
}


} // namespace
} // namespace sdrhip
#endif
