// decim_body.h -- device code of the VALU half-band decimator cascade (one piece of one stream per workgroup).
// Included by decim_kernels.hip (the all-VALU kernels) and decim_mfma.hip (the matrix-core kernels, which run
// the head and the tail of every stream through this code for the bank state's sake).
#pragma once
#include "sdrhip_internal.h"

namespace sdrhip {
namespace {

constexpr int NT = 256;    // threads per workgroup
constexpr int P0 = 2048;   // first-stage inputs per pass
constexpr int FULL = 512;  // fresh entries per plane that trigger a stage s >= 1
constexpr int HE = 32;     // history entries in front of every plane

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef int int2_t __attribute__((ext_vector_type(2)));
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

// HBFIRFilterTraits<64>::hbCoeffs = (int32_t)(literal * 2^14), HBFilterTraits.cpp:210-228
constexpr int C64[16] = {-7, 11, -20, 32, -49, 71, -101, 140, -190, 256, -345, 469, -656, 978, -1698, 5201};
// symmetric 32-tap view: H32(i) multiplies odd-plane entry k - i
__host__ __device__ constexpr int H32(int i) { return i < 16 ? C64[i] : C64[31 - i]; }
__host__ __device__ constexpr unsigned pack_taps(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }

// plane strides in dwords; all are 32 mod 64 dwords (= 8 mod 16 slots of 16 bytes)
constexpr int PK0_DW = 16 + 512 + 16; // packed int16 first stage: 32 + 1024 entries, 2 per dword (+ pad)
constexpr int I0_DW = 32 + 1024;      // int32 first stage
constexpr int SN_DW = 32 + FULL;      // stages >= 1

template <int NS_, bool PK_> struct Geo {
    static constexpr int NS = NS_;
    static constexpr bool PK = PK_;
    static constexpr int stride(int s) { return s == 0 ? (PK ? PK0_DW : I0_DW) : SN_DW; }
    static constexpr int base(int s) { return s == 0 ? 0 : base(s - 1) + 4 * stride(s - 1); }
    static constexpr int ldsDw = base(NS);
    // plane p = parity * 2 + comp  (E_I, E_Q, O_I, O_Q)
    static constexpr int plane(int s, int parity, int comp) { return base(s) + (parity * 2 + comp) * stride(s); }
};

__device__ __forceinline__ int dot2(unsigned a, unsigned taps, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, taps), acc, false);
}

__device__ __forceinline__ int dot2_link(unsigned a, unsigned taps, int acc) { return dot2(a, taps, acc); }

// first link of a dot2 chain: the three-address form (d = a . b + c) takes the chain's initial value from a
// VGPR that stays put; hipcc only emits the accumulate-in-place v_dot2c and a v_mov per chain in front of it
__device__ __forceinline__ int dot2_init(unsigned a, unsigned taps, int c)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(taps), "v"(c));
    return d;
}

// one v_mad_i32_i24 (hipcc otherwise splits the FIR into v_mul_i32_i24 + v_add3_u32 trees,
// ~25 % more lane-ops).  The tap is wave-uniform: one SGPR operand, within the constant-bus limit.
__device__ __forceinline__ int mad24(int a, int tap, int acc)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(tap), "v"(acc));
    return d;
}

struct OutCtx {
    unsigned *out;     // stream base (dwords = IQ samples, or frame area)
    size_t out_pos;    // index of the next final output of this stream
    bool store;        // false during warm-up
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

__device__ __forceinline__ void store_one(const OutCtx &oc, size_t k, unsigned v)
{
    if (!oc.frame_mode) {
        oc.out[k] = v;
    } else {
        // UDPSinkFEC::write framing (UDPSinkFEC.cpp:134-155): 127 samples per super block,
        // block 0 of a frame is the meta block
        uint64_t g = oc.frame_sample_base + k;
        uint64_t f = g / 16129u;
        unsigned w = (unsigned)(g - f * 16129u);
        unsigned b = w / 127u, i = w - b * 127u;
        oc.out[((size_t)f * oc.frame_blocks + 1 + b) * 128u + 1 + i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// one invocation of half-band stage S: `valid` outputs exist; results go to the planes of
// stage S+1 behind `fill_next` fresh entries, or (last stage) to global memory.
template <class G, int S> __device__ __forceinline__ void run_stage(int *lds, int tid, int valid, int fill_next, int bias, const OutCtx &oc)
{
    constexpr bool PK = G::PK && S == 0;
    constexpr bool LAST = (S == G::NS - 1);
    constexpr int R = (S == 0) ? 8 : 4;
    const int j = tid >> 1, comp = tid & 1;
    const int k0 = j * R;
    if (k0 >= valid) return;
    const int *pe = lds + G::plane(S, 0, 0) + comp * G::stride(S);
    const int *po = lds + G::plane(S, 1, 0) + comp * G::stride(S);
    int res[R];
    if constexpr (PK) {
        // window dword d holds odd-plane buffer entries k0 + 2d, k0 + 2d + 1
        constexpr int WO = (R + 32) / 2, WE = R / 2 + 4;
        unsigned wo[WO], we[WE];
#pragma unroll
        for (int d = 0; d < WO; d += 4) {
            uint4_t v = *reinterpret_cast<const uint4_t *>(po + k0 / 2 + d);
            wo[d] = v.x; wo[d + 1] = v.y; wo[d + 2] = v.z; wo[d + 3] = v.w;
        }
#pragma unroll
        for (int d = 0; d < WE; d += 4) { // even-plane window starts at buffer entry k0 + 16
            uint4_t v = *reinterpret_cast<const uint4_t *>(pe + k0 / 2 + 8 + d);
            we[d] = v.x; we[d + 1] = v.y; we[d + 2] = v.z; we[d + 3] = v.w;
        }
        // The R accumulation chains advance in lockstep (link t of every chain, then link t + 1): consecutive
        // instructions of a wave are then independent.  Written chain after chain, every v_dot2c waits for
        // the one before it, and the SIMD only stays busy while other waves happen to have VALU work.
        const int bias13 = bias << 13;
        int acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) // window entries r+1 .. r+32 = dwords (r+1)/2 ..; entry x <-> tap 32 + r - x
            acc[r] = (r & 1) ? dot2_init(we[(r + 1) / 2], pack_taps(8192, 0), bias13) : dot2_init(we[r / 2], pack_taps(0, 8192), bias13);
#pragma unroll
        for (int t = 0; t < 17; ++t) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r & 1) { // 16 links: dwords (r+1)/2 + p, p = 0..15
                    if (t < 16) acc[r] = dot2_link(wo[(r + 1) / 2 + t], pack_taps(H32(31 - 2 * t), H32(30 - 2 * t)), acc[r]);
                } else {     // 17 links: a half-used dword at either end
                    const unsigned taps = t == 0 ? pack_taps(0, H32(31)) : (t == 16 ? pack_taps(H32(0), 0) : pack_taps(H32(32 - 2 * t), H32(31 - 2 * t)));
                    acc[r] = dot2_link(wo[r / 2 + t], taps, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) res[r] = acc[r] >> 13;
    } else {
        int wo[R + 32], we[R + 4];
#pragma unroll
        for (int x = 0; x < R + 32; x += 4) {
            int4_t v = *reinterpret_cast<const int4_t *>(po + k0 + x);
            wo[x] = v.x; wo[x + 1] = v.y; wo[x + 2] = v.z; wo[x + 3] = v.w;
        }
#pragma unroll
        for (int x = 0; x < R + 4; x += 4) {
            int4_t v = *reinterpret_cast<const int4_t *>(pe + k0 + 16 + x);
            we[x] = v.x; we[x + 1] = v.y; we[x + 2] = v.z; we[x + 3] = v.w;
        }
        // acc = sum c[i] * (s[n-2i] + s[n-62+2i]) + ((s[n-31] + bias) << 13), n = 2k+1; the R chains in
        // lockstep (tap i of every output, then tap i + 1), see above
        int acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = (int)((unsigned)(we[r + 1] + bias) << 13);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = mad24(wo[r + 32 - i] + wo[r + 1 + i], C64[i], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) res[r] = acc[r] >> 13;
    }

    if constexpr (LAST) {
        // lanes 2j (I) and 2j+1 (Q) hold the same outputs: fetch the partner's values with a
        // quad-perm DPP move, the I lane packs and stores
        unsigned o[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int other = __builtin_amdgcn_update_dpp(0, res[r], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
            o[r] = final_pack(res[r], other, oc.norm, oc.trunk);
        }
        if (comp != 0 || !oc.store) return;
        const size_t k = oc.out_pos + k0;
        if (!oc.frame_mode && k0 + R <= valid) {
#pragma unroll
            for (int r = 0; r < R; r += 4) *reinterpret_cast<uint4_t *>(oc.out + k + r) = (uint4_t){o[r], o[r + 1], o[r + 2], o[r + 3]};
        } else if (!oc.frame_mode) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (k0 + r < valid) oc.out[k + r] = o[r];
        } else {
            // UDPSinkFEC::write framing (UDPSinkFEC.cpp:134-155): 127 samples per super block, block 0
            // of a frame is the meta block.  One division per thread, then the (frame, block, index)
            // position advances with carries.
            const uint64_t g = oc.frame_sample_base + k;
            const uint64_t f = g / 16129u;
            const unsigned w = (unsigned)(g - f * 16129u);
            unsigned b = w / 127u, i = w - b * 127u;
            size_t dw = ((size_t)f * oc.frame_blocks + 1 + b) * 128u + 1 + i;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (k0 + r < valid) oc.out[dw] = o[r];
                ++i; ++dw;
                if (i == 127u) {
                    i = 0; ++b; dw += 1; // skip the next super block's 4-byte header
                    if (b == 127u) { b = 0; dw += ((size_t)oc.frame_blocks - 127u) * 128u; } // next frame: skip its recovery blocks and block 0
                }
            }
        }
    } else {
        // outputs k0.. are inputs k0.. of stage S+1: even -> E plane, odd -> O plane, entry k/2
        int *ne = lds + G::plane(S + 1, 0, 0) + comp * G::stride(S + 1) + HE + fill_next + k0 / 2;
        int *no = lds + G::plane(S + 1, 1, 0) + comp * G::stride(S + 1) + HE + fill_next + k0 / 2;
        if constexpr (R == 8) {
            *reinterpret_cast<int4_t *>(ne) = (int4_t){res[0], res[2], res[4], res[6]};
            *reinterpret_cast<int4_t *>(no) = (int4_t){res[1], res[3], res[5], res[7]};
        } else {
            *reinterpret_cast<int2_t *>(ne) = (int2_t){res[0], res[2]};
            *reinterpret_cast<int2_t *>(no) = (int2_t){res[1], res[3]};
        }
    }
}

// history of stage S: entries [consumed, consumed + 32) -> [0, 32) of its four planes.
// One wave moves two planes (read, then write, in lockstep: source and destination may overlap).
template <class G, int S> __device__ __forceinline__ void slide(int *lds, int tid, int consumed)
{
    if (tid >= 128) return;
    const int p = tid >> 5, e = tid & 31;
    int *pl = lds + G::base(S) + p * G::stride(S);
    if constexpr (G::PK && S == 0) {
        short *ps = reinterpret_cast<short *>(pl);
        const short v = ps[e + consumed]; // (the store below depends on the load: whole wave reads first)
        ps[e] = v;
    } else {
        const int v = pl[e + consumed];
        pl[e] = v;
    }
}

template <class G, int S> __device__ __forceinline__ void hist_put(int *lds, int p, int e, int v)
{
    int *pl = lds + G::base(S) + p * G::stride(S);
    if constexpr (G::PK && S == 0) reinterpret_cast<short *>(pl)[e] = (short)v;
    else pl[e] = v;
}
template <class G, int S> __device__ __forceinline__ int hist_get(const int *lds, int p, int e)
{
    const int *pl = lds + G::base(S) + p * G::stride(S);
    if constexpr (G::PK && S == 0) return reinterpret_cast<const short *>(pl)[e];
    else return pl[e];
}

// state word index -> (stage, kernel plane): state plane = comp * 2 + parity, kernel plane = parity * 2 + comp
template <class G, int S = 0> __device__ __forceinline__ void state_load(int *lds, int tid, const int32_t *st, bool zero)
{
    if (tid < 128) {
        const int sp = tid >> 5, e = tid & 31;
        const int kp = (sp & 1) * 2 + (sp >> 1);
        hist_put<G, S>(lds, kp, e, zero ? 0 : st[S * 4 * DEC_HIST + sp * DEC_HIST + e]);
    }
    if constexpr (S + 1 < G::NS) state_load<G, S + 1>(lds, tid, st, zero);
}
template <class G, int S = 0> __device__ __forceinline__ void state_store(const int *lds, int tid, int32_t *st)
{
    if (tid < 128) {
        const int sp = tid >> 5, e = tid & 31;
        const int kp = (sp & 1) * 2 + (sp >> 1);
        st[S * 4 * DEC_HIST + sp * DEC_HIST + e] = hist_get<G, S>(lds, kp, e);
    }
    if constexpr (S + 1 < G::NS) state_store<G, S + 1>(lds, tid, st);
}

// stages 1 .. NS-1 of one pass (each phase = slide of the previous stage + maybe this stage)
template <class G, int S> __device__ __forceinline__ void later_stages(int *lds, int tid, int (&fill)[6], int consumed_prev, bool flush, int bias, OutCtx &oc)
{
    if constexpr (S < G::NS) {
        slide<G, S - 1>(lds, tid, consumed_prev);
        fill[S] += consumed_prev / 2;
        const int have = fill[S];
        const bool run = have > 0 && (have >= FULL || flush);
        if (run) {
            run_stage<G, S>(lds, tid, have, fill[S + 1 < 6 ? S + 1 : 5], bias, oc);
            if (S == G::NS - 1 && oc.store) oc.out_pos += have;
            fill[S] = 0;
        }
        __syncthreads();
        if (run) later_stages<G, S + 1>(lds, tid, fill, have, flush, bias, oc);
    } else {
        slide<G, G::NS - 1>(lds, tid, consumed_prev);
        __syncthreads();
    }
}

// FC: 0 inf, 1 sup (fs/4 rotate + sum of four raw samples first), 2 cen
// One piece [seg_start, seg_end) of one stream (raw sample indices, multiples of the pass size except at the
// end of the call).  from_state: the piece starts where the bank state was saved (no warm-up), otherwise the
// histories are rebuilt from the WRAW preceding raw samples.  store_state: the piece ends the call.
// meta_id / meta_n: this workgroup writes the meta blocks of frames meta_id, meta_id + meta_n, ...
template <int L, int FC, bool PACK16>
__device__ __forceinline__ void decim_piece(const DecimArgs &a, int *lds, int stream, size_t seg_start, size_t seg_end, bool from_state,
                                            bool store_state, int meta_id, int meta_n)
{
    constexpr bool CEN = (FC == 2);
    constexpr int NS = CEN ? L : L - 2;
    constexpr int RAWSH = CEN ? 0 : 2;   // raw samples per first-stage input = 1 << RAWSH
    constexpr int PRAW = P0 << RAWSH;    // raw samples per pass
    constexpr int NLD = PRAW / 4 / NT;   // dwordx4 loads per thread and pass
    constexpr int WRAW = 64 << L;        // warm-up length in raw samples
    using G = Geo<NS, PACK16>;
    const int tid = threadIdx.x;
    const unsigned *in = reinterpret_cast<const unsigned *>(a.in) + (size_t)stream * a.in_stride;

    const int32_t *stc = a.state_cur + (size_t)stream * DEC_STATE_WORDS;
    state_load<G>(lds, tid, stc, !from_state);

    OutCtx oc;
    oc.norm = a.norm; oc.trunk = a.trunk;
    oc.frame_mode = a.frame_mode; oc.frame_blocks = a.frame_blocks; oc.frame_sample_base = a.frame_sample_base;
    oc.out = reinterpret_cast<unsigned *>(a.out) + (size_t)stream * a.out_stride;
    oc.out_pos = seg_start >> L;

    // fused Rx pipe: meta block + super block headers of the frames this call starts, one frame per
    // workgroup (frame i of the stream by segment i mod nseg); nothing else writes those dwords
    if (a.frame_mode) {
        // (i) the 24-byte records, a frame per THREAD: two 64-bit divisions and a CRC each -- one frame at a time, with the record
        // formed by a whole wave, this loop took ~1.2 us per frame, and the matrix-core launch leaves it to three workgroups per
        // stream: 130 frames = the launch's tail (profiles/r05_rx_direct.txt); (ii) zero fill and block headers, a frame per pass
        for (int fi = meta_id + tid * meta_n; fi < a.meta_count; fi += NT * meta_n) {
            unsigned w[6];
            frame_meta_words_thread(a.meta_w, a.meta_idx0, a.meta_rate, fi, w);
            unsigned *fr = oc.out + (size_t)(a.meta_first + fi) * a.frame_blocks * 128u;
            fr[0] = (a.meta_frame_count0 + (unsigned)fi) & 0xffffu;
#pragma unroll
            for (int k = 0; k < 6; ++k) fr[1 + k] = w[k];
        }
        if (tid < 128) {
            for (int fi = meta_id; fi < a.meta_count; fi += meta_n) {
                unsigned *fr = oc.out + (size_t)(a.meta_first + fi) * a.frame_blocks * 128u;
                const unsigned fidx = (a.meta_frame_count0 + (unsigned)fi) & 0xffffu;
                if (tid >= 7) fr[tid] = 0u; // block 0 behind the record: zeros (512 bytes = 128 dwords)
                if (tid >= 1) fr[(size_t)tid * 128] = fidx | ((unsigned)tid << 16);
            }
        }
    }

    int fill[6] = {0, 0, 0, 0, 0, 0};
    bool warm = !from_state;
    size_t pos = warm ? seg_start - WRAW : 0;
    size_t region_end = warm ? seg_start : seg_end;

    uint4_t ld[NLD];
    auto issue = [&](size_t p, size_t rend) {
        const unsigned *src = in + p;
        const size_t left = rend - p;
        if (left >= (size_t)PRAW) { // full pass (wave-uniform): no per-lane bounds checks
#pragma unroll
            for (int n = 0; n < NLD; ++n) ld[n] = SDRHIP_STREAM_LOAD(reinterpret_cast<const uint4_t *>(src + 4 * (tid + n * NT)));
            return;
        }
        const unsigned rem = (unsigned)left;
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const unsigned q = (unsigned)(tid + n * NT);
            uint4_t v = (uint4_t){0u, 0u, 0u, 0u};
            if (4 * q + 3 < rem) {
                v = SDRHIP_STREAM_LOAD(reinterpret_cast<const uint4_t *>(src + 4 * q));
            } else if (4 * q < rem) { // ragged tail of a call whose length is not a multiple of 4
                v.x = src[4 * q];
                if (4 * q + 1 < rem) v.y = src[4 * q + 1];
                if (4 * q + 2 < rem) v.z = src[4 * q + 2];
            }
            ld[n] = v;
        }
    };
    issue(pos, region_end);
    __syncthreads();

    while (true) {
        const size_t left = region_end - pos;
        const int cnt_raw = left < (size_t)PRAW ? (int)left : PRAW;
        const int cnt0 = cnt_raw >> RAWSH;
        // ---- commit the prefetched samples to the first stage's planes (de-interleave / rotate)
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int q = tid + n * NT;
            const uint4_t v = ld[n];
            if constexpr (CEN && PACK16) {
                // samples 4q (even), 4q+1 (odd), 4q+2 (even), 4q+3 (odd) as packed int16 pairs
                lds[G::plane(0, 0, 0) + 16 + q] = (int)__builtin_amdgcn_perm(v.z, v.x, 0x05040100u); // E_I
                lds[G::plane(0, 0, 1) + 16 + q] = (int)__builtin_amdgcn_perm(v.z, v.x, 0x07060302u); // E_Q
                lds[G::plane(0, 1, 0) + 16 + q] = (int)__builtin_amdgcn_perm(v.w, v.y, 0x05040100u); // O_I
                lds[G::plane(0, 1, 1) + 16 + q] = (int)__builtin_amdgcn_perm(v.w, v.y, 0x07060302u); // O_Q
            } else if constexpr (CEN) {
                *reinterpret_cast<int2_t *>(&lds[G::plane(0, 0, 0) + HE + 2 * q]) = (int2_t){(int)(short)(v.x & 0xffff), (int)(short)(v.z & 0xffff)};
                *reinterpret_cast<int2_t *>(&lds[G::plane(0, 0, 1) + HE + 2 * q]) = (int2_t){(int)v.x >> 16, (int)v.z >> 16};
                *reinterpret_cast<int2_t *>(&lds[G::plane(0, 1, 0) + HE + 2 * q]) = (int2_t){(int)(short)(v.y & 0xffff), (int)(short)(v.w & 0xffff)};
                *reinterpret_cast<int2_t *>(&lds[G::plane(0, 1, 1) + HE + 2 * q]) = (int2_t){(int)v.y >> 16, (int)v.w >> 16};
            } else {
                const int I0 = (short)(v.x & 0xffff), Q0 = (int)v.x >> 16, I1 = (short)(v.y & 0xffff), Q1 = (int)v.y >> 16;
                const int I2 = (short)(v.z & 0xffff), Q2 = (int)v.z >> 16, I3 = (short)(v.w & 0xffff), Q3 = (int)v.w >> 16;
                int x, y;
                if constexpr (FC == 0) { // Decimators.cpp:351-352
                    x = I0 - Q1 + Q3 - I2; y = Q0 - Q2 + I1 - I3;
                } else {                 // Decimators.cpp:384-385
                    x = Q0 - I1 - Q2 + I3; y = -I0 - Q1 + I2 + Q3;
                }
                lds[G::plane(0, q & 1, 0) + HE + (q >> 1)] = x;
                lds[G::plane(0, q & 1, 1) + HE + (q >> 1)] = y;
            }
        }
        // ---- what comes next, and its loads in flight while this pass computes
        const size_t next_pos = pos + cnt_raw;
        const bool flush = next_pos >= region_end;
        bool more = !flush;
        size_t n_pos = next_pos, n_end = region_end;
        if (flush && warm && seg_start < seg_end) { more = true; n_pos = seg_start; n_end = seg_end; }
        if (more) issue(n_pos, n_end);
        __syncthreads();

        oc.store = !warm;
        const int valid0 = cnt0 >> 1;
        run_stage<G, 0>(lds, tid, valid0, fill[1], a.bias, oc);
        if (NS == 1 && oc.store) oc.out_pos += valid0;
        __syncthreads();
        later_stages<G, 1>(lds, tid, fill, valid0, flush, a.bias, oc);

        if (!more) break;
        if (flush) { warm = false; region_end = seg_end; }
        pos = n_pos;
    }

    // ---- new filter state (double buffered: other workgroups still read state_cur)
    if (store_state) {
        int32_t *stn = a.state_next + (size_t)stream * DEC_STATE_WORDS;
        state_store<G>(lds, tid, stn);
        for (int i = NS * 4 * DEC_HIST + tid; i < DEC_STAGES * 4 * DEC_HIST; i += NT) stn[i] = stc[i];
    }
}


template <int L, int FC, bool PACK16> struct DecimLds {
    static constexpr int NS = (FC == 2) ? L : L - 2;
    static constexpr int dwords = Geo<NS, PACK16>::ldsDw;
    static_assert(dwords * 4 <= 64 * 1024, "LDS budget");
};

} // namespace
} // namespace sdrhip
