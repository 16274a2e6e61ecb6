// decim_kernels.hip -- cascaded integer half-band decimators for gfx950 (MI355X).
//
// Replaces the per-sample ring-buffer loops of Decimators::decimate{8..64}_{inf,sup} and
// decimate{2..64}_cen (Decimators.cpp:94-1305) over IntHalfbandFilterEO1/DB<64>::myDecimate
// (IntHalfbandFilterEO1.h:34-42,100-147 / IntHalfbandFilterDB.h:32-49,79-107).
//
// Design (see DESIGN.md "K1"):
//  * grid = (segments, streams); a 256-thread workgroup walks one segment of one stream in
//    sub-chunks of C0 first-stage inputs.  All stage inputs live in LDS as four planes per
//    stage ({I,Q} x {even,odd} input parity, i.e. the polyphase split of the half-band
//    filter); 32 entries of history sit in front of every plane and are carried from
//    sub-chunk to sub-chunk, so nothing is recomputed inside a segment.
//  * output k of a stage = 32-tap FIR over the odd-parity plane + centre tap from the even
//    plane.  A thread produces R consecutive outputs of one component from a register
//    window of R+32 plane entries fetched with ds_read_b128 (planes are padded so that the
//    per-thread stride is an odd number of 16-byte slots: conflict-free).
//  * the first stage of the centred modes reads the raw int16 samples packed two per dword
//    and uses v_dot2c_i32_i16 (2 taps per lane-op, exact: |acc| < 2^30); later stages use
//    v_add_u32 + v_mad_i32_i24 (inputs are |x| <= 2^18, so the 24-bit multiply is exact in
//    the low 32 bits, i.e. the reference's wrap-around int32 arithmetic).
//  * segment 0 loads the filter histories from the bank's state; every other segment
//    rebuilds them by processing 64 * 2^L raw samples before its first sample (>= the
//    62 * (2^L - 1) samples that reach the last stage's history) and discarding the
//    outputs.  The workgroup of the last segment stores the new state (double buffered).
#include "sdrhip_internal.h"

namespace sdrhip {
namespace {

constexpr int NT = 256; // threads per workgroup

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

// HBFIRFilterTraits<64>::hbCoeffs = (int32_t)(literal * 2^14), HBFilterTraits.cpp:210-228
constexpr int C64[16] = {-7, 11, -20, 32, -49, 71, -101, 140, -190, 256, -345, 469, -656, 978, -1698, 5201};
// symmetric 32-tap view: H32(i), i = 0..31 multiplies odd-plane entry k - i
__host__ __device__ constexpr int H32(int i) { return i < 16 ? C64[i] : C64[31 - i]; }
__host__ __device__ constexpr unsigned pack_taps(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Geometry of one kernel variant.  C0 = first-stage inputs per sub-chunk, NS = half-band
// stages, PACK16 = first stage reads packed int16 planes.
template <int C0_, int NS_, bool PACK16_> struct Geo {
    static constexpr int C0 = C0_, NS = NS_;
    static constexpr bool PACK16 = PACK16_;
    static constexpr int n(int s) { return C0 >> (s + 1); } // outputs per component per sub-chunk
    static constexpr bool last(int s) { return s == NS - 1; }
    // outputs per thread: the last stage handles both components in one thread (it packs I/Q)
    static constexpr int R(int s) { return last(s) ? cmax(4, n(s) / NT) : cmax(8, 2 * n(s) / NT); }
    static constexpr int T(int s) { return last(s) ? n(s) / R(s) : 2 * n(s) / R(s); }
    static constexpr bool packed(int s) { return PACK16 && s == 0; }
    static constexpr int histDw(int s) { return packed(s) ? 16 : 32; }
    static constexpr int newDw(int s) { return packed(s) ? n(s) / 2 : n(s); }
    static constexpr int blk(int s) { return packed(s) ? R(s) / 2 : R(s); } // dwords per thread block
    static constexpr int pad(int s) { return ((blk(s) / 4) % 2 == 0) ? 4 : 0; }
    static constexpr int planeDw(int s) { return (histDw(s) + newDw(s)) / blk(s) * (blk(s) + pad(s)); }
    static constexpr int stageBase(int s) { return s == 0 ? 0 : stageBase(s - 1) + 4 * planeDw(s - 1); }
    static constexpr int ldsDw = stageBase(NS);
    // dword address inside a plane of stage s
    static constexpr int addr(int s, int d) { return d + pad(s) * (d / blk(s)); }
};

template <class G, int S> __device__ __forceinline__ int plane_addr(int d)
{
    return d + G::pad(S) * (d / G::blk(S));
}

__device__ __forceinline__ int dot2(unsigned a, unsigned taps, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, taps), acc, false);
}

struct OutCtx {
    int16_t *out;       // stream base
    size_t out_base;    // index of the sub-chunk's first final output
    int valid;          // final outputs of this sub-chunk that exist
    bool store;         // false during warm-up
    int norm, trunk;
    int frame_mode, frame_blocks;
    uint64_t frame_sample_base;
};

__device__ __forceinline__ unsigned final_pack(int i, int q, int norm, int trunk)
{
    // `x << norm_shift >> trunk_shift` then FixReal truncation, Decimators.cpp:112-113
    int a = (int)((unsigned)i << norm) >> trunk;
    int b = (int)((unsigned)q << norm) >> trunk;
    return ((unsigned)a & 0xffffu) | ((unsigned)b << 16);
}

__device__ __forceinline__ void store_final(const OutCtx &oc, int k, unsigned v)
{
    if (!oc.frame_mode) {
        reinterpret_cast<unsigned *>(oc.out)[oc.out_base + k] = v;
    } else {
        // UDPSinkFEC::write framing (UDPSinkFEC.cpp:134-155): 127 samples per super block,
        // block 0 of a frame is the meta block
        uint64_t g = oc.frame_sample_base + oc.out_base + (uint64_t)k;
        uint64_t f = g / 16129u;
        unsigned w = (unsigned)(g - f * 16129u);
        unsigned b = w / 127u, i = w - b * 127u;
        size_t dw = ((size_t)f * oc.frame_blocks + 1 + b) * 128u + 1 + i;
        reinterpret_cast<unsigned *>(oc.out)[dw] = v;
    }
}

// ------------------------------------------------------------------------------------------
// one half-band stage over the current sub-chunk
template <class G, int S> __device__ __forceinline__ void run_stage(int *lds, int tid, int cnt0, int bias, const OutCtx &oc)
{
    constexpr int R = G::R(S);
    constexpr int T = G::T(S);
    constexpr bool LAST = G::last(S);
    constexpr bool PK = G::packed(S);
    constexpr int PLANE = G::planeDw(S);
    const int valid = cnt0 >> (S + 1); // outputs of this stage that exist in this sub-chunk
    if (tid >= T) return;
    const int tl = LAST ? tid : tid % (T / 2);
    const int comp0 = LAST ? 0 : tid / (T / 2);
    const int k0 = tl * R;
    if (k0 >= valid) return;
    int *st = lds + G::stageBase(S);

    int res[LAST ? 2 : 1][R];
#pragma unroll
    for (int ci = 0; ci < (LAST ? 2 : 1); ++ci) {
        const int comp = comp0 + ci;
        const int *pe = st + (comp * 2 + 0) * PLANE;
        const int *po = st + (comp * 2 + 1) * PLANE;
        if constexpr (PK) {
            // window dword j holds odd-plane buffer entries k0 + 2j, k0 + 2j + 1
            constexpr int WO = (R + 32) / 2, WE = (R / 2 + 4);
            unsigned wo[WO], we[WE];
            const int base = plane_addr<G, S>(k0 / 2);
#pragma unroll
            for (int j = 0; j < WO; j += 4) {
                uint4_t v = *reinterpret_cast<const uint4_t *>(po + base + G::addr(S, j));
                wo[j] = v.x; wo[j + 1] = v.y; wo[j + 2] = v.z; wo[j + 3] = v.w;
            }
            // even-plane window starts at buffer entry k0 + 16 (dword k0/2 + 8)
#pragma unroll
            for (int j = 0; j < WE; j += 4) {
                uint4_t v = *reinterpret_cast<const uint4_t *>(pe + base + G::addr(S, 8 + j));
                we[j] = v.x; we[j + 1] = v.y; we[j + 2] = v.z; we[j + 3] = v.w;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int acc = bias << 13;
                if (r & 1) {
                    // entries r+1 .. r+32 = dwords (r+1)/2 .. (r+1)/2+15 ; entry x <-> tap 32+r-x
                    acc = dot2(we[(r + 1) / 2], pack_taps(8192, 0), acc);
#pragma unroll
                    for (int p = 0; p < 16; ++p)
                        acc = dot2(wo[(r + 1) / 2 + p], pack_taps(H32(31 - 2 * p), H32(30 - 2 * p)), acc);
                } else {
                    acc = dot2(we[r / 2], pack_taps(0, 8192), acc);
                    acc = dot2(wo[r / 2], pack_taps(0, H32(31)), acc);
#pragma unroll
                    for (int p = 1; p < 16; ++p)
                        acc = dot2(wo[r / 2 + p], pack_taps(H32(32 - 2 * p), H32(31 - 2 * p)), acc);
                    acc = dot2(wo[r / 2 + 16], pack_taps(H32(0), 0), acc);
                }
                res[ci][r] = acc >> 13;
            }
        } else {
            int wo[R + 32], we[R + 4];
            const int base = plane_addr<G, S>(k0);
#pragma unroll
            for (int x = 0; x < R + 32; x += 4) {
                int4_t v = *reinterpret_cast<const int4_t *>(po + base + G::addr(S, x));
                wo[x] = v.x; wo[x + 1] = v.y; wo[x + 2] = v.z; wo[x + 3] = v.w;
            }
#pragma unroll
            for (int x = 0; x < R + 4; x += 4) {
                int4_t v = *reinterpret_cast<const int4_t *>(pe + base + G::addr(S, 16 + x));
                we[x] = v.x; we[x + 1] = v.y; we[x + 2] = v.z; we[x + 3] = v.w;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // acc = sum c[i] * (s[n-2i] + s[n-62+2i]) + ((s[n-31] + bias) << 13), n = 2k+1
                int acc = (int)((unsigned)(we[r + 1] + bias) << 13);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc += __mul24(wo[r + 32 - i] + wo[r + 1 + i], C64[i]);
                res[ci][r] = acc >> 13;
            }
        }
    }

    if constexpr (LAST) {
        if (!oc.store) return;
        if (!oc.frame_mode && (R % 4 == 0) && k0 + R <= oc.valid) {
            unsigned *dst = reinterpret_cast<unsigned *>(oc.out) + oc.out_base + k0;
#pragma unroll
            for (int r = 0; r < R; r += 4) {
                uint4_t v;
                v.x = final_pack(res[0][r], res[1][r], oc.norm, oc.trunk);
                v.y = final_pack(res[0][r + 1], res[1][r + 1], oc.norm, oc.trunk);
                v.z = final_pack(res[0][r + 2], res[1][r + 2], oc.norm, oc.trunk);
                v.w = final_pack(res[0][r + 3], res[1][r + 3], oc.norm, oc.trunk);
                *reinterpret_cast<uint4_t *>(dst + r) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (k0 + r < oc.valid) store_final(oc, k0 + r, final_pack(res[0][r], res[1][r], oc.norm, oc.trunk));
        }
    } else {
        // outputs k0 .. k0+R-1 are inputs k0.. of stage S+1: even -> E plane, odd -> O plane,
        // plane entry 32 + k/2
        constexpr int NPL = G::planeDw(S + 1);
        int *nx = lds + G::stageBase(S + 1) + comp0 * 2 * NPL;
        const int e0 = 32 + k0 / 2;
#pragma unroll
        for (int j = 0; j < R / 2; j += 4) {
            int4_t ve, vo;
            ve.x = res[0][2 * j]; ve.y = res[0][2 * j + 2]; ve.z = res[0][2 * j + 4]; ve.w = res[0][2 * j + 6];
            vo.x = res[0][2 * j + 1]; vo.y = res[0][2 * j + 3]; vo.z = res[0][2 * j + 5]; vo.w = res[0][2 * j + 7];
            const int a = plane_addr<G, S + 1>(e0 + j);
            *reinterpret_cast<int4_t *>(nx + a) = ve;
            *reinterpret_cast<int4_t *>(nx + NPL + a) = vo;
        }
    }
}

template <class G, int S = 0> __device__ __forceinline__ void run_all_stages(int *lds, int tid, int cnt0, int bias, const OutCtx &oc)
{
    run_stage<G, S>(lds, tid, cnt0, bias, oc);
    __syncthreads();
    if constexpr (S + 1 < G::NS) run_all_stages<G, S + 1>(lds, tid, cnt0, bias, oc);
}

// ------------------------------------------------------------------------------------------
// history access at entry granularity (entry e of plane p of stage s)
template <class G> __device__ __forceinline__ int hist_get(const int *lds, int s, int p, int e)
{
    // runtime stage index: small switch-free arithmetic via constexpr tables is not possible,
    // so walk the (at most six) stages
    int v = 0;
#define SDRHIP_CASE(S_)                                                                                         \
    if constexpr (S_ < G::NS)                                                                                   \
        if (s == S_) {                                                                                          \
            const int *pl = lds + G::stageBase(S_) + p * G::planeDw(S_);                                        \
            if constexpr (G::packed(S_)) {                                                                      \
                const short *ps = reinterpret_cast<const short *>(pl);                                          \
                v = ps[2 * plane_addr<G, S_>(e >> 1) + (e & 1)];                                                \
            } else {                                                                                            \
                v = pl[plane_addr<G, S_>(e)];                                                                   \
            }                                                                                                   \
        }
    SDRHIP_CASE(0) SDRHIP_CASE(1) SDRHIP_CASE(2) SDRHIP_CASE(3) SDRHIP_CASE(4) SDRHIP_CASE(5)
#undef SDRHIP_CASE
    return v;
}

template <class G> __device__ __forceinline__ void hist_put(int *lds, int s, int p, int e, int v)
{
#define SDRHIP_CASE(S_)                                                                                         \
    if constexpr (S_ < G::NS)                                                                                   \
        if (s == S_) {                                                                                          \
            int *pl = lds + G::stageBase(S_) + p * G::planeDw(S_);                                              \
            if constexpr (G::packed(S_)) {                                                                      \
                short *ps = reinterpret_cast<short *>(pl);                                                      \
                ps[2 * plane_addr<G, S_>(e >> 1) + (e & 1)] = (short)v;                                         \
            } else {                                                                                            \
                pl[plane_addr<G, S_>(e)] = v;                                                                   \
            }                                                                                                   \
        }
    SDRHIP_CASE(0) SDRHIP_CASE(1) SDRHIP_CASE(2) SDRHIP_CASE(3) SDRHIP_CASE(4) SDRHIP_CASE(5)
#undef SDRHIP_CASE
}

// ------------------------------------------------------------------------------------------
// FC: 0 inf, 1 sup (fs/4 rotate + sum of four raw samples first), 2 cen
template <int L, int FC, bool PACK16, int C0> __global__ __launch_bounds__(NT) void decim_kernel(DecimArgs a)
{
    constexpr bool CEN = (FC == 2);
    constexpr int NS = CEN ? L : L - 2;
    constexpr int RAWSH = CEN ? 0 : 2;           // raw samples per first-stage input = 1 << RAWSH
    constexpr int CRAW = C0 << RAWSH;            // raw samples per sub-chunk
    constexpr int WRAW = 64 << L;                // warm-up length in raw samples
    static_assert(WRAW <= CRAW, "warm-up must fit one sub-chunk");
    using G = Geo<C0, NS, PACK16>;
    static_assert(G::ldsDw * 4 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) int lds[];

    const int tid = threadIdx.x;
    const int seg = blockIdx.x;
    const int stream = blockIdx.y;
    const int16_t *in = a.in + 2 * (size_t)stream * a.in_stride;
    const size_t seg_raw = (size_t)a.nsub_per_seg * CRAW;
    const size_t seg_start = (size_t)seg * seg_raw;
    size_t seg_end = seg_start + seg_raw;
    if (seg_end > a.n_used) seg_end = a.n_used;
    const bool last_seg = (seg == a.nseg - 1);

    // ---- history init: state for segment 0, zeros (rebuilt by the warm-up) otherwise
    const int32_t *stc = a.state_cur + (size_t)stream * DEC_STATE_WORDS;
    for (int i = tid; i < NS * 4 * DEC_HIST; i += NT) {
        const int s = i / (4 * DEC_HIST), p = (i / DEC_HIST) & 3, e = i % DEC_HIST;
        hist_put<G>(lds, s, p, e, seg == 0 ? stc[i] : 0);
    }
    __syncthreads();

    OutCtx oc;
    oc.norm = a.norm; oc.trunk = a.trunk;
    oc.frame_mode = a.frame_mode; oc.frame_blocks = a.frame_blocks; oc.frame_sample_base = a.frame_sample_base;
    oc.out = a.frame_mode ? reinterpret_cast<int16_t *>(reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride)
                          : a.out + 2 * (size_t)stream * a.out_stride;

    bool warm = (seg != 0);
    size_t pos = warm ? seg_start - WRAW : 0;
    while (pos < seg_end) {
        const int cnt_raw = warm ? WRAW : (int)((seg_end - pos) < (size_t)CRAW ? (seg_end - pos) : (size_t)CRAW);
        const int cnt0 = cnt_raw >> RAWSH;
        // ---- load + de-interleave the sub-chunk into the first stage's planes
        {
            const unsigned *src = reinterpret_cast<const unsigned *>(in) + pos;
            int *st = lds;
            constexpr int P0 = G::planeDw(0);
#pragma unroll 4
            for (int q = tid; q < CRAW / 4; q += NT) {
                if (4 * q >= cnt_raw) break;
                uint4_t v;
                if (4 * q + 3 < cnt_raw) {
                    v = *reinterpret_cast<const uint4_t *>(src + 4 * q);
                } else { // ragged tail of a call whose length is not a multiple of 4
                    v.x = src[4 * q];
                    v.y = (4 * q + 1 < cnt_raw) ? src[4 * q + 1] : 0u;
                    v.z = (4 * q + 2 < cnt_raw) ? src[4 * q + 2] : 0u;
                    v.w = 0u;
                }
                if constexpr (CEN && PACK16) {
                    // samples 4q (even), 4q+1 (odd), 4q+2 (even), 4q+3 (odd): packed int16 pairs
                    const int ad = plane_addr<G, 0>(16 + q);
                    st[0 * P0 + ad] = (int)__builtin_amdgcn_perm(v.z, v.x, 0x05040100u); // I even
                    st[1 * P0 + ad] = (int)__builtin_amdgcn_perm(v.w, v.y, 0x05040100u); // I odd
                    st[2 * P0 + ad] = (int)__builtin_amdgcn_perm(v.z, v.x, 0x07060302u); // Q even
                    st[3 * P0 + ad] = (int)__builtin_amdgcn_perm(v.w, v.y, 0x07060302u); // Q odd
                } else if constexpr (CEN) {
                    const int ad = plane_addr<G, 0>(32 + 2 * q); // two consecutive entries, same block
                    st[0 * P0 + ad] = (int)(short)(v.x & 0xffff); st[0 * P0 + ad + 1] = (int)(short)(v.z & 0xffff);
                    st[1 * P0 + ad] = (int)(short)(v.y & 0xffff); st[1 * P0 + ad + 1] = (int)(short)(v.w & 0xffff);
                    st[2 * P0 + ad] = (int)v.x >> 16; st[2 * P0 + ad + 1] = (int)v.z >> 16;
                    st[3 * P0 + ad] = (int)v.y >> 16; st[3 * P0 + ad + 1] = (int)v.w >> 16;
                } else {
                    const int I0 = (short)(v.x & 0xffff), Q0 = (int)v.x >> 16, I1 = (short)(v.y & 0xffff), Q1 = (int)v.y >> 16;
                    const int I2 = (short)(v.z & 0xffff), Q2 = (int)v.z >> 16, I3 = (short)(v.w & 0xffff), Q3 = (int)v.w >> 16;
                    int x, y;
                    if constexpr (FC == 0) { // Decimators.cpp:351-352
                        x = I0 - Q1 + Q3 - I2; y = Q0 - Q2 + I1 - I3;
                    } else {                 // Decimators.cpp:384-385
                        x = Q0 - I1 - Q2 + I3; y = -I0 - Q1 + I2 + Q3;
                    }
                    const int ad = plane_addr<G, 0>(32 + (q >> 1));
                    st[(0 + (q & 1)) * P0 + ad] = x;
                    st[(2 + (q & 1)) * P0 + ad] = y;
                }
            }
        }
        __syncthreads();

        oc.out_base = pos >> L;
        oc.valid = cnt_raw >> L;
        oc.store = !warm;
        run_all_stages<G>(lds, tid, cnt0, a.bias, oc); // ends with a barrier

        // ---- slide the histories: entries [valid, valid+32) -> [0, 32) of every plane
        {
            constexpr int NK = (NS * 4 * DEC_HIST + NT - 1) / NT;
            int keep[NK];
#pragma unroll
            for (int n = 0; n < NK; ++n) {
                const int i = tid + n * NT;
                const int s = i / (4 * DEC_HIST), p = (i / DEC_HIST) & 3, e = i % DEC_HIST;
                keep[n] = (i < NS * 4 * DEC_HIST) ? hist_get<G>(lds, s, p, e + (cnt0 >> (s + 1))) : 0;
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NK; ++n) {
                const int i = tid + n * NT;
                const int s = i / (4 * DEC_HIST), p = (i / DEC_HIST) & 3, e = i % DEC_HIST;
                if (i < NS * 4 * DEC_HIST) hist_put<G>(lds, s, p, e, keep[n]);
            }
            __syncthreads();
        }
        pos += cnt_raw;
        warm = false;
    }

    // ---- new filter state (double buffered: other workgroups still read state_cur)
    if (last_seg) {
        int32_t *stn = a.state_next + (size_t)stream * DEC_STATE_WORDS;
        for (int i = tid; i < DEC_STAGES * 4 * DEC_HIST; i += NT) {
            const int s = i / (4 * DEC_HIST), p = (i / DEC_HIST) & 3, e = i % DEC_HIST;
            stn[i] = (s < NS) ? hist_get<G>(lds, s, p, e) : stc[i];
        }
    }
}

template <int L, int FC, bool PACK16, int C0> hipError_t launch_variant(const DecimArgs &a, hipStream_t stream)
{
    constexpr bool CEN = (FC == 2);
    constexpr int NS = CEN ? L : L - 2;
    using G = Geo<C0, NS, PACK16>;
    constexpr size_t lds_bytes = (size_t)G::ldsDw * 4;
    static bool attr_set = false;
    auto kern = decim_kernel<L, FC, PACK16, C0>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(a.nseg, a.nstreams);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds_bytes, stream, a);
    return hipGetLastError();
}

constexpr int C0_DEFAULT = 4096;
template <int L> constexpr int c0_for() { return (64 << L) > C0_DEFAULT ? (64 << L) : C0_DEFAULT; }

} // namespace

void plan_decimate(int log2decim, int fcpos, size_t n_used, int nstreams, int *nsub_per_seg, int *nseg)
{
    const bool cen = (fcpos == 2);
    (void)log2decim;
    const size_t craw = cen ? (size_t)C0_DEFAULT : (size_t)C0_DEFAULT * 4;
    size_t nsub = (n_used + craw - 1) / craw;
    if (nsub == 0) nsub = 1;
    // enough workgroups to fill 256 CUs (two resident per CU), but at most 8 sub-chunks of
    // warm-up-free work per segment once the chip is full
    size_t per = 8;
    while (per > 1 && ((nsub + per - 1) / per) * (size_t)nstreams < 2048) per >>= 1;
    *nsub_per_seg = (int)per;
    *nseg = (int)((nsub + per - 1) / per);
}

hipError_t launch_decimate(int log2decim, int fcpos, bool pack16, const DecimArgs &a, hipStream_t stream)
{
#define SDRHIP_CEN(L_)                                                                                          \
    case L_:                                                                                                    \
        return pack16 ? launch_variant<L_, 2, true, c0_for<L_>()>(a, stream)                                    \
                      : launch_variant<L_, 2, false, c0_for<L_>()>(a, stream);
#define SDRHIP_ROT(L_, FC_)                                                                                     \
    case L_:                                                                                                    \
        return launch_variant<L_, FC_, false, C0_DEFAULT>(a, stream);
    if (fcpos == 2) {
        switch (log2decim) {
            SDRHIP_CEN(1) SDRHIP_CEN(2) SDRHIP_CEN(3) SDRHIP_CEN(4) SDRHIP_CEN(5) SDRHIP_CEN(6)
        }
    } else if (fcpos == 0) {
        switch (log2decim) { SDRHIP_ROT(3, 0) SDRHIP_ROT(4, 0) SDRHIP_ROT(5, 0) SDRHIP_ROT(6, 0) }
    } else {
        switch (log2decim) { SDRHIP_ROT(3, 1) SDRHIP_ROT(4, 1) SDRHIP_ROT(5, 1) SDRHIP_ROT(6, 1) }
    }
#undef SDRHIP_CEN
#undef SDRHIP_ROT
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// filter-less variants: decimate1 (Decimators.cpp:22-35), decimate2_inf/sup (:38-91),
// decimate4_inf/sup (:127-170).  One thread per group of four input samples.
__global__ void decim_simple_kernel(int log2decim, int fcpos, const int16_t *in, size_t in_stride, int16_t *out,
                                    size_t out_stride, size_t n_in, int norm, int trunk)
{
    const int stream = blockIdx.y;
    const unsigned *src = reinterpret_cast<const unsigned *>(in) + (size_t)stream * in_stride;
    unsigned *dst = reinterpret_cast<unsigned *>(out) + (size_t)stream * out_stride;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (log2decim == 0) {
        for (size_t i = g; i < n_in; i += (size_t)gridDim.x * blockDim.x) {
            unsigned v = src[i];
            int a = (short)(v & 0xffff), b = (int)v >> 16;
            dst[i] = (((unsigned)a << norm) & 0xffffu) | (((unsigned)b << norm) << 16);
        }
        return;
    }
    const size_t n_resize = n_in >> log2decim;
    const size_t ngroups = n_in / 4;
    for (size_t q = g; q < (n_in + 3) / 4; q += (size_t)gridDim.x * blockDim.x) {
        if (q >= ngroups) { // out.resize() elements the loop never writes stay zero (fresh vector)
            if (log2decim == 1 && 2 * q < n_resize) dst[2 * q] = 0;
            continue;
        }
        const unsigned v0 = src[4 * q], v1 = src[4 * q + 1], v2 = src[4 * q + 2], v3 = src[4 * q + 3];
        const int I0 = (short)(v0 & 0xffff), Q0 = (int)v0 >> 16, I1 = (short)(v1 & 0xffff), Q1 = (int)v1 >> 16;
        const int I2 = (short)(v2 & 0xffff), Q2 = (int)v2 >> 16, I3 = (short)(v3 & 0xffff), Q3 = (int)v3 >> 16;
        if (log2decim == 1) {
            int xa, ya, xb, yb;
            if (fcpos == 0) { xa = I0 - Q1; ya = Q0 + I1; xb = Q3 - I2; yb = -Q2 - I3; }
            else { xa = Q0 - I1; ya = -I0 - Q1; xb = I3 - Q2; yb = I2 + Q3; }
            dst[2 * q] = final_pack(xa, ya, norm, trunk);
            dst[2 * q + 1] = final_pack(xb, yb, norm, trunk);
        } else {
            int x, y;
            if (fcpos == 0) { x = I0 - Q1 + Q3 - I2; y = Q0 - Q2 + I1 - I3; }
            else { x = Q0 - I1 - Q2 + I3; y = -I0 - Q1 + I2 + Q3; }
            dst[q] = final_pack(x, y, norm, trunk);
        }
    }
}

hipError_t launch_decimate_simple(int log2decim, int fcpos, const int16_t *in, size_t in_stride, int16_t *out,
                                  size_t out_stride, size_t n_in, int nstreams, int norm, int trunk, hipStream_t stream)
{
    size_t work = log2decim == 0 ? n_in : (n_in + 3) / 4;
    size_t blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(decim_simple_kernel, dim3((unsigned)blocks, nstreams), dim3(256), 0, stream, log2decim, fcpos, in,
                       in_stride, out, out_stride, n_in, norm, trunk);
    return hipGetLastError();
}

} // namespace sdrhip
