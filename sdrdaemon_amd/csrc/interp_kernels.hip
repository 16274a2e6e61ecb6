// interp_kernels.hip -- cascaded integer half-band interpolators for gfx950 (MI355X).
//
// Replaces Interpolators::interpolate{2..64}_cen (Interpolators.cpp:23-606) over
// IntHalfbandFilterEO1/DB<64|32|16>::myInterpolate (IntHalfbandFilterEO1.h:44-65,149-168;
// DB twin IntHalfbandFilterDB.h:51-72,109-128 -- identical arithmetic):
//     v[2m]   = u[m - O/4]
//     v[2m+1] = (sum_{i < O/4} c[i] * (u[m - (O/2 - 1) + i] + u[m - i])) >> 13
// stage orders 64, 32, 16, 16, 16 (Interpolators.h:31-33); int32 between stages, int16
// truncation at the end.  interpolate64_cen is reproduced as the reference has it: five
// stages and 32 zero samples after every 32 outputs (Interpolators.cpp:363-606).
//
// Same structure as the decimator kernel: grid = (segments, streams); a 256-thread
// workgroup walks its segment in sub-chunks of CI inputs; every stage's inputs sit in LDS
// (I and Q planes, int32) behind 32 entries of history that are carried between sub-chunks;
// a thread produces the 2R outputs of R consecutive inputs of one component from a register
// window of R + O/2 entries; the last stage handles both components, packs int16 I/Q and
// stores 16-byte vectors.  Segment 0 takes the histories from the bank state, other
// segments rebuild them from the 64 preceding inputs (the cascade's memory is 43 inputs).
#include "sdrhip_internal.h"

namespace sdrhip {
namespace {

constexpr int NT = 256;
constexpr int HIST = 32;
constexpr int WARM = 64;

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

// HBFIRFilterTraits<64|32|16>::hbCoeffs, HBFilterTraits.cpp:210-228, 62-72, 25-31
constexpr int T64[16] = {-7, 11, -20, 32, -49, 71, -101, 140, -190, 256, -345, 469, -656, 978, -1698, 5201};
constexpr int T32[8] = {-30, 63, -135, 261, -469, 830, -1605, 5176};
constexpr int T16[4] = {-85, 380, -1246, 5041};

__host__ __device__ constexpr int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ constexpr int stage_order(int s) { return s == 0 ? 64 : (s == 1 ? 32 : 16); }
__host__ __device__ constexpr int tap(int order, int i) { return order == 64 ? T64[i] : (order == 32 ? T32[i] : T16[i]); }

template <int CI_, int NS_> struct IGeo {
    static constexpr int CI = CI_, NS = NS_;
    static constexpr int n(int s) { return CI << s; } // inputs of stage s per component per sub-chunk
    static constexpr bool last(int s) { return s == NS - 1; }
    static constexpr int R(int s) { return last(s) ? imax(4, n(s) / NT) : imax(8, 2 * n(s) / NT); }
    static constexpr int T(int s) { return last(s) ? n(s) / R(s) : 2 * n(s) / R(s); }
    static constexpr int planeDw(int s) { return HIST + n(s); }
    static constexpr int stageBase(int s) { return s == 0 ? 0 : stageBase(s - 1) + 2 * planeDw(s - 1); }
    static constexpr int ldsDw = stageBase(NS);
};

struct IOut {
    int16_t *out;
    size_t out_base; // first output sample index of this sub-chunk
    size_t out_limit; // outputs of this stream that exist
    bool store;
    bool stuff64;    // interpolate64_cen layout
};

template <class G, int S> __device__ __forceinline__ void istage(int *lds, int tid, int cnt, const IOut &oc)
{
    constexpr int R = G::R(S), T = G::T(S);
    constexpr bool LAST = G::last(S);
    constexpr int O = stage_order(S), K = O / 4, S2 = O / 2;
    constexpr int PLANE = G::planeDw(S);
    const int valid = cnt << S; // inputs of this stage that exist in this sub-chunk
    if (tid >= T) return;
    const int tl = LAST ? tid : tid % (T / 2);
    const int comp0 = LAST ? 0 : tid / (T / 2);
    const int m0 = tl * R;
    if (m0 >= valid) return;
    int *st = lds + G::stageBase(S);

    int ev[LAST ? 2 : 1][R], od[LAST ? 2 : 1][R];
#pragma unroll
    for (int ci = 0; ci < (LAST ? 2 : 1); ++ci) {
        const int *pl = st + (comp0 + ci) * PLANE + HIST + m0 - S2; // window entry x <-> u[m0 - O/2 + x]
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
            for (int i = 0; i < K; ++i) acc += __mul24(w[r + 1 + i] + w[r + S2 - i], tap(O, i));
            ev[ci][r] = w[r + K + 0 + (S2 - 2 * K)]; // u[m - K] <-> x = r + O/2 - K
            od[ci][r] = acc >> 13;
        }
    }
    if constexpr (LAST) {
        if (!oc.store) return;
        unsigned o[2 * R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            o[2 * r] = ((unsigned)ev[0][r] & 0xffffu) | ((unsigned)ev[1][r] << 16);
            o[2 * r + 1] = ((unsigned)od[0][r] & 0xffffu) | ((unsigned)od[1][r] << 16);
        }
        size_t idx = oc.out_base + 2 * (size_t)m0; // chain output index
        if (oc.stuff64) idx = (idx >> 5) * 64 + (idx & 31);
        unsigned *dst = reinterpret_cast<unsigned *>(oc.out) + idx;
        if (m0 + R <= valid) {
#pragma unroll
            for (int j = 0; j < 2 * R; j += 4) *reinterpret_cast<uint4_t *>(dst + j) = (uint4_t){o[j], o[j + 1], o[j + 2], o[j + 3]};
            if (oc.stuff64) {
#pragma unroll
                for (int j = 0; j < 2 * R; j += 4) *reinterpret_cast<uint4_t *>(dst + 32 + j) = (uint4_t){0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2 * R; ++j)
                if (m0 + j / 2 < valid) {
                    dst[j] = o[j];
                    if (oc.stuff64) dst[32 + j] = 0u;
                }
        }
    } else {
        int *nx = lds + G::stageBase(S + 1) + comp0 * G::planeDw(S + 1) + HIST + 2 * m0;
#pragma unroll
        for (int r = 0; r < R; r += 2)
            *reinterpret_cast<int4_t *>(nx + 2 * r) = (int4_t){ev[0][r], od[0][r], ev[0][r + 1], od[0][r + 1]};
    }
}

template <class G, int S = 0> __device__ __forceinline__ void irun(int *lds, int tid, int cnt, const IOut &oc)
{
    istage<G, S>(lds, tid, cnt, oc);
    __syncthreads();
    if constexpr (S + 1 < G::NS) irun<G, S + 1>(lds, tid, cnt, oc);
}

template <class G> __device__ __forceinline__ int *hist_ptr(int *lds, int s, int comp, int e)
{
    int *p = lds;
#define SDRHIP_ICASE(S_)                                                                                        \
    if constexpr (S_ < G::NS)                                                                                   \
        if (s == S_) p = lds + G::stageBase(S_) + comp * G::planeDw(S_) + e;
    SDRHIP_ICASE(0) SDRHIP_ICASE(1) SDRHIP_ICASE(2) SDRHIP_ICASE(3) SDRHIP_ICASE(4)
#undef SDRHIP_ICASE
    return p;
}

// L = log2 interpolation (6 = the reference's 5-stage + zero stuffing variant)
template <int L, int CI> __global__ __launch_bounds__(NT) void interp_kernel(InterpArgs a)
{
    constexpr int NS = (L == 6) ? 5 : L;
    using G = IGeo<CI, NS>;
    static_assert(G::ldsDw * 4 <= 64 * 1024, "LDS budget");
    static_assert(WARM <= CI, "warm-up must fit one sub-chunk");
    __shared__ __attribute__((aligned(16))) int lds[G::ldsDw];

    const int tid = threadIdx.x;
    const int seg = blockIdx.x, stream = blockIdx.y;
    const unsigned *in = reinterpret_cast<const unsigned *>(a.in) + (size_t)stream * a.in_stride;
    const size_t seg_len = (size_t)a.nsub_per_seg * CI;
    const size_t seg_start = (size_t)seg * seg_len;
    size_t seg_end = seg_start + seg_len;
    if (seg_end > a.n_in) seg_end = a.n_in;
    const bool last_seg = (seg == a.nseg - 1);

    const int32_t *stc = a.state_cur + (size_t)stream * INT_STATE_WORDS;
    for (int i = tid; i < NS * 2 * HIST; i += NT) {
        const int s = i / (2 * HIST), comp = (i / HIST) & 1, e = i % HIST;
        *hist_ptr<G>(lds, s, comp, e) = (seg == 0) ? stc[i] : 0;
    }
    __syncthreads();

    IOut oc;
    oc.out = a.out + 2 * (size_t)stream * a.out_stride;
    oc.stuff64 = (L == 6);
    oc.out_limit = a.n_in << L;

    bool warm = (seg != 0);
    size_t pos = warm ? seg_start - WARM : 0;
    while (pos < seg_end) {
        const int cnt = warm ? WARM : (int)((seg_end - pos) < (size_t)CI ? (seg_end - pos) : (size_t)CI);
        for (int m = tid; m < cnt; m += NT) {
            const unsigned v = in[pos + m];
            lds[G::stageBase(0) + HIST + m] = (int)(short)(v & 0xffffu);
            lds[G::stageBase(0) + G::planeDw(0) + HIST + m] = (int)v >> 16;
        }
        __syncthreads();
        oc.out_base = pos << NS;
        oc.store = !warm;
        irun<G>(lds, tid, cnt, oc);
        // slide the histories: entries [valid, valid + 32) -> [0, 32)
        {
            constexpr int NK = (NS * 2 * HIST + NT - 1) / NT;
            int keep[NK];
#pragma unroll
            for (int n = 0; n < NK; ++n) {
                const int i = tid + n * NT;
                const int s = i / (2 * HIST), comp = (i / HIST) & 1, e = i % HIST;
                keep[n] = (i < NS * 2 * HIST) ? *hist_ptr<G>(lds, s, comp, e + (cnt << s)) : 0;
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NK; ++n) {
                const int i = tid + n * NT;
                const int s = i / (2 * HIST), comp = (i / HIST) & 1, e = i % HIST;
                if (i < NS * 2 * HIST) *hist_ptr<G>(lds, s, comp, e) = keep[n];
            }
            __syncthreads();
        }
        pos += cnt;
        warm = false;
    }
    if (last_seg) {
        int32_t *stn = a.state_next + (size_t)stream * INT_STATE_WORDS;
        for (int i = tid; i < INT_STAGES * 2 * HIST; i += NT) {
            const int s = i / (2 * HIST), comp = (i / HIST) & 1, e = i % HIST;
            stn[i] = (s < NS) ? *hist_ptr<G>(lds, s, comp, e) : stc[i];
        }
    }
}

template <int L> constexpr int ci_for() { return L == 1 ? 2048 : (L == 2 ? 1024 : (L == 3 ? 512 : (L == 4 ? 256 : 128))); }

template <int L> hipError_t launch_l(const InterpArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL((interp_kernel<L, ci_for<L>()>), dim3(a.nseg, a.nstreams), dim3(NT), 0, stream, a);
    return hipGetLastError();
}

} // namespace

void plan_interpolate(int log2interp, size_t n_in, int nstreams, int *nsub_per_seg, int *nseg)
{
    const size_t ci = log2interp == 1 ? 2048 : (log2interp == 2 ? 1024 : (log2interp == 3 ? 512 : (log2interp == 4 ? 256 : 128)));
    size_t nsub = (n_in + ci - 1) / ci;
    if (nsub == 0) nsub = 1;
    size_t per = 16; // warm-up is 64 inputs: 16 sub-chunks per segment keep its cost below 3 %
    while (per > 1 && ((nsub + per - 1) / per) * (size_t)nstreams < 2048) per >>= 1;
    *nsub_per_seg = (int)per;
    *nseg = (int)((nsub + per - 1) / per);
}

hipError_t launch_interpolate(int log2interp, const InterpArgs &a, hipStream_t stream)
{
    switch (log2interp) {
    case 1: return launch_l<1>(a, stream);
    case 2: return launch_l<2>(a, stream);
    case 3: return launch_l<3>(a, stream);
    case 4: return launch_l<4>(a, stream);
    case 5: return launch_l<5>(a, stream);
    case 6: return launch_l<6>(a, stream);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// meta block + super block headers of the frames started by an Rx call (UDPSinkFEC.cpp:87-132,
// 150-152): block 0 = header {frameIndex, 0, 0} + 24-byte MetaDataFEC + zero fill; blocks
// 1..127 get {frameIndex, blockIndex, 0}.  One workgroup per (frame, stream).
__global__ void frame_meta_kernel(uint8_t *work, size_t stream_bytes, int frame_blocks, int first_frame, unsigned frame_count0,
                                  const uint8_t *meta24)
{
    const int f = first_frame + blockIdx.x;
    const int stream = blockIdx.y;
    unsigned *fr = reinterpret_cast<unsigned *>(work + (size_t)stream * stream_bytes + (size_t)f * frame_blocks * 512);
    const unsigned fidx = (frame_count0 + blockIdx.x) & 0xffffu;
    const int t = threadIdx.x;
    if (t < 128) {
        const unsigned *m = reinterpret_cast<const unsigned *>(meta24);
        fr[t] = t == 0 ? fidx : (t <= 6 ? m[t - 1] : 0u); // block 0: 512 bytes = 128 dwords
        if (t >= 1) fr[(size_t)t * 128] = fidx | ((unsigned)t << 16);
    }
}

hipError_t launch_frame_meta(uint8_t *work, size_t stream_bytes, int frame_blocks, int nstreams, int first_frame, int nframes,
                             unsigned frame_count0, const uint8_t *meta24, hipStream_t stream)
{
    if (nframes <= 0) return hipSuccess;
    hipLaunchKernelGGL(frame_meta_kernel, dim3(nframes, nstreams), dim3(128), 0, stream, work, stream_bytes, frame_blocks, first_frame,
                       frame_count0, meta24);
    return hipGetLastError();
}

} // namespace sdrhip
