// gf_kernels.hip -- GF(2^8) block arithmetic of CM256 for gfx950 (MI355X).
//
// Replaces the gf256_muladd_mem loops behind CM256::cm256_encode (UDPSinkFEC.cpp:246,
// 31 x 128 multiply-adds + 128 XORs of 508 bytes per frame at 128+32) and
// CM256::cm256_decode (SDRdaemonFECBuffer.cpp:197).  Both are one operation here:
//     out[row] = XOR_col  coef[row][col] * in[col]        (508-byte blocks, bytes in GF(256))
// with the encoder's Cauchy rows, or with the per-erasure-pattern decode matrix that
// gf256.cpp derives on the host.
//
// Multiply by a wave-uniform constant m without MFMA and without per-byte gathers: the data
// dword is split into three selector dwords (bits 0-2, 3-5, 6-7 of each byte; computed once
// per data dword and reused by every row) and the product is three v_perm_b32 byte-table
// lookups into the 8+8+4 byte tables of m, XOR-ed together (GF multiplication by a constant
// is GF(2)-linear).  The 32-byte table of each of the 256 constants lives in LDS and is
// fetched with uniform-address ds_read_b128/b32 (broadcast); the coefficients of the
// workgroup's rows sit next to it.  A lane owns one 16-byte slab of a block, a half-wave one
// 508-byte block, a wave GF_FRAMES_PER_GROUP = 2 frames, and accumulates RB rows at a time in
// registers; a workgroup = 4 waves = 4 x RB rows of the same frames.
#include "sdrhip_internal.h"

#include <type_traits>
#include <utility>

namespace sdrhip {
namespace {

typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
#include "gf_encode128_body.h"
// rows accumulated per wave: RB = 4, 6 or 8, a workgroup (4 waves) = 4 x RB rows of the same frames.  The
// launch picks the smallest tile that covers the matrix in one workgroup row (24 erasures -> RB = 6): the
// column slabs are then read and split into selectors once instead of once per 16 rows.
static_assert(GF_FRAMES_PER_GROUP == 2, "lane mapping below assumes one frame per half-wave");

struct Sel { unsigned a, b, c; };

__device__ __forceinline__ Sel make_sel(unsigned x)
{
    Sel s;
    s.a = x & 0x07070707u;
    s.b = (x >> 3) & 0x07070707u;
    s.c = (x >> 6) & 0x03030303u;
    return s;
}

// product of the four bytes of the data dword (given as selectors) with the constant whose
// tables are (ta_lo, ta_hi | tb_lo, tb_hi | tc)
__device__ __forceinline__ unsigned mulc(const Sel &s, const uint4_t &t, unsigned tc)
{
    unsigned pa = __builtin_amdgcn_perm(t.y, t.x, s.a);
    unsigned pb = __builtin_amdgcn_perm(t.w, t.z, s.b);
    unsigned pc = __builtin_amdgcn_perm(0u, tc, s.c);
    return __builtin_amdgcn_bitop3_b32(pa, pb, pc, 0x96); // pa ^ pb ^ pc in one v_bitop3_b32 (hipcc emits two v_xor)
}

__device__ __forceinline__ uint4_t load_slab(const uint8_t *p, int l)
{
    // 16 bytes at p + 16 l of a 508-byte block; the last lane's slab is 12 bytes
    const unsigned *q = reinterpret_cast<const unsigned *>(p) + 4 * l;
    uint4_t v;
    v.x = q[0]; v.y = q[1]; v.z = q[2];
    v.w = (l < 31) ? q[3] : 0u;
    return v;
}

__device__ __forceinline__ void store_slab(uint8_t *p, int l, const uint4_t &v)
{
    unsigned *q = reinterpret_cast<unsigned *>(p) + 4 * l;
    q[0] = v.x; q[1] = v.y; q[2] = v.z;
    if (l < 31) q[3] = v.w;
}

template <int RB> __global__ __launch_bounds__(GF_NT) void gf_apply_kernel(GfArgs a)
{
    constexpr int ROWS_PER_WG = 4 * RB;
    __shared__ __attribute__((aligned(16))) unsigned tab[256 * 8];              // 8 KB
    __shared__ __attribute__((aligned(16))) uint8_t coef[ROWS_PER_WG * 256];    // rows x cols (cols <= 256)

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int h = lane >> 5, l = lane & 31;
    const int group = blockIdx.x;
    const int row0 = blockIdx.y * ROWS_PER_WG;

    // this half-wave's frame (both frames of a group share one coefficient matrix)
    int fr = a.frame_list ? a.frame_list[group * GF_FRAMES_PER_GROUP + h] : group * GF_FRAMES_PER_GROUP + h;
    if (fr >= a.nframes) fr = -1;
    // coefficient matrix of this group: shared (0), one per group, or an indirect slot of a pattern cache
    const int cm = a.group_cm ? a.group_cm[group] : (a.coef_per_frame ? group : 0);
    const size_t mstride = a.matrix_rows ? (size_t)a.matrix_rows : (size_t)a.rows; // rows between consecutive matrices
    const int cols = a.cols;

    for (int i = tid; i < 256 * 8; i += GF_NT) tab[i] = reinterpret_cast<const unsigned *>(a.tab)[i];
    {
        const uint8_t *cg = a.coef + ((size_t)cm * mstride + row0) * cols;
        const int nrows = (a.rows - row0) < ROWS_PER_WG ? (a.rows - row0) : ROWS_PER_WG;
        for (int i = tid; i < ROWS_PER_WG * cols; i += GF_NT) {
            int r = i / cols;
            coef[r * 256 + (i - r * cols)] = (r < nrows) ? cg[i] : 0;
        }
    }
    __syncthreads();

    const int r0 = wave * RB; // this wave's rows inside the workgroup tile
    if (row0 + r0 >= a.rows) return;

    uint4_t acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = (uint4_t){0u, 0u, 0u, 0u};

    const int16_t *csrc = a.col_src ? a.col_src + (size_t)cm * cols : nullptr;
    const uint8_t *fbase = a.in + (size_t)(fr < 0 ? 0 : fr) * a.in_frame_bytes + a.in_off;
    uint4_t xn = (uint4_t){0u, 0u, 0u, 0u};
    {
        const int sb = csrc ? csrc[0] : 0;
        if (fr >= 0 && sb >= 0) xn = load_slab(fbase + (size_t)sb * a.in_pitch, l);
    }
    for (int j = 0; j < cols; ++j) {
        const uint4_t x = xn;
        if (j + 1 < cols) { // next column's slab in flight while this one is multiplied
            const int sb = csrc ? csrc[j + 1] : j + 1;
            xn = (uint4_t){0u, 0u, 0u, 0u};
            if (fr >= 0 && sb >= 0) xn = load_slab(fbase + (size_t)sb * a.in_pitch, l);
        }
        const Sel s0 = make_sel(x.x), s1 = make_sel(x.y), s2 = make_sel(x.z), s3 = make_sel(x.w);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const unsigned m = coef[(r0 + rb) * 256 + j];
            const uint4_t t = *reinterpret_cast<const uint4_t *>(&tab[m * 8]);
            const unsigned tc = tab[m * 8 + 4];
            acc[rb].x ^= mulc(s0, t, tc);
            acc[rb].y ^= mulc(s1, t, tc);
            acc[rb].z ^= mulc(s2, t, tc);
            acc[rb].w ^= mulc(s3, t, tc);
        }
    }

    if (fr < 0) return;
    const int16_t *rdst = a.row_dst ? a.row_dst + (size_t)cm * mstride : nullptr;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = row0 + r0 + rb;
        if (r >= a.rows) break;
        const int db = rdst ? rdst[r] : r;
        if (db < 0) continue;
        store_slab(a.out + (size_t)fr * a.out_frame_bytes + (size_t)db * a.out_pitch + a.out_off, l, acc[rb]);
    }
}

__global__ __launch_bounds__(GF_NT, 4) void gf_encode128_kernel(Enc128Args a) // 105 VGPRs: 4 waves per SIMD
{
    __shared__ __attribute__((aligned(16))) unsigned char ldsraw[ENC128_LDS_BYTES];
    gf_encode128_wg(a, (int)blockIdx.x, ldsraw);
}

// the additive-FFT encoder (rows <= 32): a workgroup per frame
#include "gf_encode128_fft.h"
__global__ __launch_bounds__(GF_NT, FFT_WAVES_PER_EU) void gf_encode128_fft_kernel(Enc128Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char ldsraw[ENC128_FFT_KERNEL_LDS];
    gf_encode128_fft_unit(a, (int)blockIdx.x, ldsraw);
}

// ... in half-frame workgroups (Enc128Args::half_units): grid = 2 x frames, 128 threads
__global__ __launch_bounds__(128, FFT_WAVES_PER_EU) void gf_encode128_fft_half_kernel(Enc128Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char ldsraw[ENC128_FFT_HALF_LDS];
    gf_encode128_fft_half_wg(a, (int)blockIdx.x, ldsraw);
}

// The Rx pipe's last launch: the encoder's workgroups and, behind them, K2's (the frames left open at either end of the call,
// meta blocks, headers).  One launch instead of two: K2 used to run first because the encoder reads what K2 writes (block 0 of
// the frames the call starts, the tail of the frame the call completes); now the encoder derives / fetches both itself
// (Enc128Args::meta_*, lin_straddle), so the two roles touch disjoint bytes (the encoder's write-back of block 0 repeats K2's).
#include "frame_pack_body.h"
__global__ __launch_bounds__(GF_NT, 4) void gf_encode128_pack_kernel(Enc128Args a, FrameArgs f, unsigned pack_bx)
{
    __shared__ __attribute__((aligned(16))) unsigned char ldsraw[ENC128_LDS_BYTES];
    const unsigned nenc = 2u * (unsigned)a.nlist;
    if (blockIdx.x < nenc) {
        gf_encode128_wg(a, (int)blockIdx.x, ldsraw);
    } else {
        const unsigned u = blockIdx.x - nenc;
        frame_pack_wg(f, (int)(u / pack_bx), u % pack_bx, pack_bx);
    }
}
// ... with the additive-FFT encoder: nlist encoder workgroups in front of K2's
__global__ __launch_bounds__(GF_NT, FFT_WAVES_PER_EU) void gf_encode128_fft_pack_kernel(Enc128Args a, FrameArgs f, unsigned pack_bx)
{
    __shared__ __attribute__((aligned(16))) unsigned char ldsraw[ENC128_FFT_KERNEL_LDS];
    const unsigned nenc = (unsigned)a.nlist;
    if (blockIdx.x < nenc) {
        gf_encode128_fft_unit(a, (int)blockIdx.x, ldsraw);
    } else {
        const unsigned u = blockIdx.x - nenc;
        frame_pack_wg(f, (int)(u / pack_bx), u % pack_bx, pack_bx);
    }
}

// scatter copy of 508-byte blocks: dst[f][map[f][p]] = src[f][p] for map >= 0
__global__ void block_scatter_kernel(const uint8_t *src, size_t src_frame_bytes, int src_pitch, int src_off, uint8_t *dst,
                                     size_t dst_frame_bytes, int dst_pitch, int dst_off, const int16_t *map, int nblocks,
                                     int nframes)
{
    const int f = blockIdx.y;
    const int p = blockIdx.x * (blockDim.x / 32) + (threadIdx.x >> 5);
    const int l = threadIdx.x & 31;
    if (f >= nframes || p >= nblocks) return;
    const int d = map[(size_t)f * nblocks + p];
    if (d < 0) return;
    uint4_t v = load_slab(src + (size_t)f * src_frame_bytes + (size_t)p * src_pitch + src_off, l);
    store_slab(dst + (size_t)f * dst_frame_bytes + (size_t)d * dst_pitch + dst_off, l, v);
}

// headers of the recovery super blocks: {frameIndex (from block 0 of the frame), 128 + r, 0}; generic
// encoder only (gf_encode128_kernel writes them itself).  frame_list as in GfArgs (flat, -1 = none).
__global__ void fec_header_kernel(const uint8_t *frames, size_t in_frame_bytes, uint8_t *rec, size_t out_frame_bytes, int nb_fec,
                                  int first_index, int nframes, const int32_t *frame_list, int nlist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlist * nb_fec) return;
    const int fi = i / nb_fec, r = i - fi * nb_fec;
    const int f = frame_list ? frame_list[fi] : fi;
    if (f < 0 || f >= nframes) return;
    const unsigned h0 = *reinterpret_cast<const unsigned *>(frames + (size_t)f * in_frame_bytes);
    *reinterpret_cast<unsigned *>(rec + (size_t)f * out_frame_bytes + (size_t)r * 512) = (h0 & 0xffffu) | ((unsigned)(first_index + r) << 16);
}


// ------------------------------------------------------------------------------------------
// Decode planning on the device (SDRdaemonFECBuffer::writeAndRead, .cpp:143-213, + CM256Decoder::Initialize /
// Decode / DecodeM1 of the library behind cm256_decode): one workgroup per frame, no host involvement.
//
// From the 128 block indices of a frame in arrival order (the headers' blockIndex, or a caller-supplied array) it
// derives: where every received original goes (pmap: payload block, zmap: the meta block 0), which originals were
// erased (ascending, one per received recovery block, in array order: the pairing cm256 uses), and the
// n_rec x 128 matrix that turns the received blocks into the erased ones,
//     recovered_t = XOR_p coef[t][p] * block[p].
// The system to solve is M[i][t] = (y_t ^ x_0) / (x_i ^ y_t) (x_i = index of the i-th received recovery block,
// y_t = t-th erased original, x_0 = 128): a Cauchy matrix with scaled columns, whose inverse has the closed form
//     Minv[t][i] = PX_i PY_t / ((x_i ^ y_t) QX_i QY_t (y_t ^ x_0)),
//     PX_i = prod_k (x_i ^ y_k), PY_t = prod_k (x_k ^ y_t), QX_i = prod_{k != i} (x_i ^ x_k), QY_t = prod_{k != t} (y_t ^ y_k)
// (characteristic 2: no signs), O(N) logarithm sums per row / column instead of a Gauss-Jordan elimination per
// pattern on the host.  Columns of received originals: coef[t][p] = XOR_i Minv[t][i] * (y'_p ^ x_0) / (x_i ^ y'_p).
// Mirrored quirks: exactly one received recovery block takes DecodeM1's XOR shortcut whatever its row; a repeated
// original or recovery index is the library's "decode error" (-5): the frame keeps what was received (holes = 0).
struct DecPlanArgs {
    const uint8_t *rx;         // frames: [nframes][128][512], header byte 2 = block index
    size_t rx_frame_bytes;
    const uint8_t *indices;    // optional [nframes][128] (device); NULL = read the headers
    const uint8_t *explog;     // exp[512] bytes, then log[256] as uint16 (device)
    uint8_t *coef;             // [nframes][128][128]
    int16_t *pmap, *zmap;      // [nframes][128] destination of position p (payload block / block 0) or -1
    int16_t *pdst, *zdst;      // [nframes][128] destination of row t (payload block / block 0) or -1
    int32_t *nrec;             // [nframes][2]: rows to apply, row that recovers block 0 exists (0 / 1)
    uint8_t *payload_out;      // zero fill of the frames that stay incomplete
    size_t payload_frame_bytes;
    uint8_t *block0_out;       // may be NULL
    int nframes;
    uint8_t *plan2;            // [nframes][DEC128_PLAN_BYTES] records for gf_decode128_kernel (NULL: dense path only)
    int strict;                // ctx option dec_strict: deliver only what the reference's copy-back loop delivers (see the planner)
    int max_rows;              // the caller's promise (ctx option dec_max_rows): no frame carries more recovery blocks
    unsigned *stats;           // [0] += frames that broke the promise (they stay as received; sdrhip_ctx_get_counter)
};

// Per-frame record of the syndrome decoder (gf_decode128_kernel): where every original lies in the received array, which
// recovery rows came, which originals they restore, and the N x N inverse (N <= 32; bigger repairs take the dense kernel).
constexpr int DEC128_MAXN = 32;
struct Dec128Plan {
    int16_t inv[128];                         // position of original j in the received array, -1 = erased
    uint8_t rowidx[128];                      // recovery row r (block 128 + r) -> i (array order), 255 = not received
    uint8_t rpos[DEC128_MAXN];                // position of the i-th received recovery block
    uint8_t ydst[DEC128_MAXN];                // original restored by row t of the inverse (ascending)
    uint8_t minv[DEC128_MAXN * DEC128_MAXN];  // Minv[t][i] at [i][t % 4][t / 4]: the eight constants a wave (rows w, w + 4, ...) needs for syndrome i are 8 consecutive bytes
    int32_t n;                                // erased originals to restore here (0: none, or not this kernel's frame)
    int32_t m1;                               // cm256's DecodeM1: one recovery block, XOR of everything received
    int32_t maxrow;                           // highest recovery row among the received ones
    int32_t pad;
};
constexpr int DEC128_PLAN_BYTES = (int)sizeof(Dec128Plan);
static_assert(sizeof(Dec128Plan) % 16 == 0 && sizeof(Dec128Plan) == DECODE_PLAN2_BYTES, "record size (sdrhip_internal.h)");

__global__ __launch_bounds__(128) void gf_decode_plan_kernel(DecPlanArgs a)
{
    constexpr int K = 128;
    __shared__ __attribute__((aligned(16))) uint8_t s_explog[1024]; // exp[512], then log[256] (uint16): gf_explog as it is
    uint8_t *const s_exp = s_explog;
    uint16_t *const s_log = reinterpret_cast<uint16_t *>(s_explog + 512);
    __shared__ uint8_t s_idx[K], s_x[K], s_y[K], s_rank[K], s_rpos[K];
    __shared__ int s_cnt[K];
    __shared__ int s_lpx[K], s_lqx[K], s_lpy[K], s_lqy[K];
    __shared__ uint8_t s_linv[K * K]; // log of Minv[t][i] (a Cauchy inverse has no zero entry)
    __shared__ uint8_t s_le[K * K];   // log of (y'_p ^ x_0) / (x_i ^ y'_p) at [i][p]
    __shared__ int s_bad;
    __shared__ unsigned long long s_mask[2][3];
    const int f = blockIdx.x, p = threadIdx.x;
    // (one 16-byte load per thread, issued with the header byte: as two copy loops of bytes / halfwords the table arrived in six
    // global round trips one after the other)
    uint4_t el = {0u, 0u, 0u, 0u};
    if (p < 64) el = reinterpret_cast<const uint4_t *>(a.explog)[p];
    // (the header bytes non-temporal: with the default policy their 128 lines per frame allocate in the caches and evict dirty lines of
    // the launch in front -- gf_decode128_fft.h, DEC_HDR_NT: 8.1 against 4.4 us for this round trip)
    const int b = a.indices ? a.indices[(size_t)f * K + p]
                            : __builtin_amdgcn_raw_buffer_load_b8(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.rx) + (size_t)f * a.rx_frame_bytes, 0, 0x7fffffff, 0x00020000),
                                                                  (unsigned)p * 512u + 2u, 0, 2);
    if (p < 64) reinterpret_cast<uint4_t *>(s_explog)[p] = el;
    s_idx[p] = (uint8_t)b;
    s_cnt[p] = 0;
    if (p == 0) s_bad = 0;
    __syncthreads();
    if (b < K) atomicAdd(&s_cnt[b], 1);
    __syncthreads();
    // ranks: of this position among the recovery blocks (array order), of original index p among the missing ones -- population
    // counts of the two waves' ballots below the lane (a 128-step scan of LDS per thread before round 4: a third of this kernel)
    const int wv = p >> 6, ln = p & 63;
    const unsigned long long br = __ballot(b >= K), bm = __ballot(s_cnt[p] == 0), bd = __ballot(s_cnt[p] > 1);
    if (ln == 0) { s_mask[wv][0] = br; s_mask[wv][1] = bm; s_mask[wv][2] = bd; }
    __syncthreads();
    const unsigned long long below = (1ull << ln) - 1ull;
    const int nrec = __popcll(s_mask[0][0]) + __popcll(s_mask[1][0]), nmiss = __popcll(s_mask[0][1]) + __popcll(s_mask[1][1]);
    const int rrank = (wv ? __popcll(s_mask[0][0]) : 0) + __popcll(br & below);
    const int mrank = (wv ? __popcll(s_mask[0][1]) : 0) + __popcll(bm & below);
    const int dup = (s_mask[0][2] | s_mask[1][2]) != 0ull;
    const bool is_rec = b >= K;
    if (is_rec) { s_x[rrank] = (uint8_t)b; s_rank[p] = (uint8_t)rrank; s_rpos[rrank] = (uint8_t)p; }
    if (s_cnt[p] == 0 && mrank < nrec) s_y[mrank] = (uint8_t)p; // erased originals, ascending (nmiss == nrec without repeats)
    __syncthreads();

    int16_t *pmap = a.pmap + (size_t)f * K, *zmap = a.zmap + (size_t)f * K, *pdst = a.pdst + (size_t)f * K, *zdst = a.zdst + (size_t)f * K;
    pmap[p] = (b >= 1 && b < K) ? (int16_t)(b - 1) : (int16_t)-1;
    zmap[p] = b == 0 ? (int16_t)0 : (int16_t)-1;

    const int N = nrec;
    bool ok = N > 0 && !dup;
    if (N > a.max_rows) {
        // more recovery blocks than the caller promised (dec_max_rows sized the launches behind this one): the frame is left
        // as received -- like an undecodable one -- and COUNTED, never half repaired
        ok = false;
        if (p == 0) atomicAdd(a.stats, 1u);
    }
    Dec128Plan *pl = a.plan2 ? reinterpret_cast<Dec128Plan *>(a.plan2 + (size_t)f * DEC128_PLAN_BYTES) : nullptr;
    const bool syn = pl && N <= DEC128_MAXN; // this frame's repair (if any) is the syndrome kernel's
    if (pl) {
        // (the copy of the received originals is gf_decode128_kernel's job for EVERY frame; a repeated original: any copy)
        pl->inv[p] = (int16_t)-1;
        pl->rowidx[p] = 255;
        if (p == 0) { pl->n = 0; pl->m1 = 0; pl->maxrow = 0; pl->pad = 0; }
    }
    __syncthreads();
    if (pl) {
        if (b < K) pl->inv[b] = (int16_t)p;
        else if (ok && syn) { pl->rowidx[b - K] = (uint8_t)rrank; pl->rpos[rrank] = (uint8_t)p; atomicMax(&pl->maxrow, b - K); }
    }
    if (ok && N >= 2) {
        // logarithm sums of the four products
        if (p < N) {
            const int xi = s_x[p], yt = s_y[p];
            int lpx = 0, lqx = 0, lpy = 0, lqy = 0, sing = 0;
            for (int k = 0; k < N; ++k) {
                lpx += s_log[xi ^ s_y[k]];
                lpy += s_log[s_x[k] ^ yt];
                if (k != p) {
                    const int dx = xi ^ s_x[k];
                    sing |= dx == 0; // the same recovery block twice: singular
                    lqx += s_log[dx];
                    lqy += s_log[yt ^ s_y[k]];
                }
            }
            if (sing) atomicOr(&s_bad, 1);
            s_lpx[p] = lpx % 255; s_lqx[p] = lqx % 255; s_lpy[p] = lpy % 255; s_lqy[p] = lqy % 255;
        }
        __syncthreads();
        ok = !s_bad;
        if (ok) {
            for (int e = p; e < N * N; e += K) {
                const int t = e / N, i = e - t * N;
                const int yt = s_y[t], xi = s_x[i];
                const int num = s_lpx[i] + s_lpy[t];
                const int den = s_log[xi ^ yt] + s_lqx[i] + s_lqy[t] + s_log[yt ^ K];
                s_linv[t * K + i] = (uint8_t)((num + 4 * 255 - den) % 255);
            }
            if (!is_rec && !syn)
                for (int i = 0; i < N; ++i) s_le[i * K + p] = (uint8_t)((s_log[b ^ K] + 255 - s_log[s_x[i] ^ b]) % 255);
        }
        __syncthreads();
    }

    int32_t *meta = a.nrec + (size_t)f * 2;
    if (!ok) {
        // nothing to repair, or cm256's "decode error": the frame keeps what was received (SDRdaemonFECBuffer.cpp:199)
        pdst[p] = -1; zdst[p] = -1;
        if (p == 0) { meta[0] = 0; meta[1] = 0; }
        if (nmiss > 0) { // some original never arrived: zero fill (initDecodeSlot, .cpp:109), the scatter pass comes after
            unsigned *po = reinterpret_cast<unsigned *>(a.payload_out + (size_t)f * a.payload_frame_bytes);
            for (int i = p; i < 127 * 127; i += K) po[i] = 0u;
            if (a.block0_out && p < 127) reinterpret_cast<unsigned *>(a.block0_out + (size_t)f * 508)[p] = 0u;
        }
        return;
    }
    // Strict mode (ctx option dec_strict): the reference copies back only the descriptors [128 - recoveryCount, 128)
    // (SDRdaemonFECBuffer.cpp:204-211: it relies on the recovery blocks arriving last); cm256 restores erased original t into the
    // t-th recovery block in ARRAY order, so a block restored into a recovery block that sits further up in the array is never
    // copied and the frame keeps a hole (zeros) there.  hole(t) = that case; by default every restored block is delivered.
    const bool hole = a.strict && p < N && (int)s_rpos[p] < K - N;
    // destination of the recovered rows
    {
        int16_t pd = -1, zd = -1;
        if (p < N) {
            const int y = s_y[p];
            if (y >= 1) pd = (int16_t)(y - 1);
            else zd = 0;
        }
        pdst[p] = pd; zdst[p] = zd;
        if (p == 0) { meta[0] = syn ? 0 : N; meta[1] = syn ? 0 : s_y[0] == 0; } // (ascending: block 0, when erased, is row 0)
    }
    if (syn) {
        // the syndrome kernel applies the N x N inverse itself: no N x 128 product matrix
        if (p < N) pl->ydst[p] = (uint8_t)(s_y[p] | (hole ? 0x80 : 0)); // (bit 7: a hole of strict mode, the block is written as zeros)
        if (N == 1) { if (p == 0) { pl->minv[0] = 1; pl->m1 = 1; } }
        else for (int e = p; e < N * N; e += K) { const int t = e / N, i = e - t * N; pl->minv[i * DEC128_MAXN + (t & 3) * 8 + (t >> 2)] = s_exp[s_linv[t * K + i]]; }
        if (p == 0) pl->n = N;
        return;
    }
    uint8_t *coef = a.coef + (size_t)f * K * K;
    if (N == 1) { // DecodeM1: XOR of everything that was received
        coef[p] = (a.strict && (int)s_rpos[0] < K - 1) ? 0 : 1;
        return;
    }
    for (int t = 0; t < N; ++t) {
        unsigned v;
        if (a.strict && (int)s_rpos[t] < K - N) {
            v = 0; // (a hole of strict mode: the row restores zeros)
        } else if (is_rec) {
            v = s_exp[s_linv[t * K + s_rank[p]]];
        } else {
            v = 0;
            for (int i = 0; i < N; ++i) v ^= s_exp[(int)s_linv[t * K + i] + (int)s_le[i * K + p]];
        }
        coef[(size_t)t * K + p] = (uint8_t)v;
    }
}

// out[f][dst[f][t]] = XOR_p coef[f][t][p] * in[f][p] with one matrix PER FRAME (the device planner's): the two
// half-waves of a wave work on different frames and fetch different multiplier tables (two LDS addresses per read
// instead of one broadcast); rows beyond a frame's count are skipped.  which = 0: payload rows, 1: the row that
// recovers block 0.
struct DecApplyArgs {
    const uint8_t *in;
    size_t in_frame_bytes;
    uint8_t *out;
    size_t out_frame_bytes;
    int out_pitch;
    const uint8_t *coef;       // [nframes][128][128]
    const int16_t *dst;        // [nframes][128]
    const int32_t *nrec;       // [nframes][2]
    const uint8_t *tab;
    int which;
    int nframes;
};

template <int RB> __global__ __launch_bounds__(GF_NT) void gf_decode_apply_kernel(DecApplyArgs a)
{
    constexpr int K = 128, ROWS_PER_WG = 4 * RB;
    __shared__ __attribute__((aligned(16))) unsigned tab[256 * 8];
    __shared__ __attribute__((aligned(16))) uint8_t coef[2][ROWS_PER_WG * K];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, l = lane & 31;
    const int row0 = blockIdx.y * ROWS_PER_WG;
    const int f0 = blockIdx.x * 2;
    int rows[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int fr = f0 + u;
        int n = 0;
        if (fr < a.nframes) n = a.which ? (a.nrec[(size_t)fr * 2 + 1] ? 1 : 0) : a.nrec[(size_t)fr * 2];
        rows[u] = n;
    }
    if (row0 >= rows[0] && row0 >= rows[1]) return; // (uniform: nothing to recover here)
    for (int i = tid; i < 256 * 8; i += GF_NT) tab[i] = reinterpret_cast<const unsigned *>(a.tab)[i];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint8_t *cg = a.coef + ((size_t)(f0 + u) * K + row0) * K;
        const int nr = rows[u] - row0;
        for (int i = tid; i < ROWS_PER_WG * K / 4; i += GF_NT) {
            const int r = (i * 4) / K;
            reinterpret_cast<unsigned *>(coef[u])[i] = (r < nr) ? reinterpret_cast<const unsigned *>(cg)[i] : 0u;
        }
    }
    __syncthreads();
    const int fr = f0 + h;
    const int myrows = rows[h];
    const int r0 = wave * RB;
    if (row0 + r0 >= rows[0] && row0 + r0 >= rows[1]) return;

    uint4_t acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = (uint4_t){0u, 0u, 0u, 0u};
    const bool live = fr < a.nframes && row0 + r0 < myrows;
    const uint8_t *fbase = a.in + (size_t)(fr < a.nframes ? fr : 0) * a.in_frame_bytes + 4;
    const uint8_t *cw = coef[h] + r0 * K;
    uint4_t xn = (uint4_t){0u, 0u, 0u, 0u};
    if (live) xn = load_slab(fbase, l);
    for (int j = 0; j < K; ++j) {
        const uint4_t x = xn;
        if (j + 1 < K && live) xn = load_slab(fbase + (size_t)(j + 1) * 512, l);
        const Sel s0 = make_sel(x.x), s1 = make_sel(x.y), s2 = make_sel(x.z), s3 = make_sel(x.w);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const unsigned m = cw[rb * K + j];
            const uint4_t t = *reinterpret_cast<const uint4_t *>(&tab[m * 8]);
            const unsigned tc = tab[m * 8 + 4];
            acc[rb].x ^= mulc(s0, t, tc);
            acc[rb].y ^= mulc(s1, t, tc);
            acc[rb].z ^= mulc(s2, t, tc);
            acc[rb].w ^= mulc(s3, t, tc);
        }
    }
    if (!live) return;
    const int16_t *dst = a.dst + (size_t)fr * K;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = row0 + r0 + rb;
        if (r >= myrows) break;
        const int db = dst[r];
        if (db < 0) continue;
        store_slab(a.out + (size_t)fr * a.out_frame_bytes + (size_t)db * a.out_pitch, l, acc[rb]);
    }
}


// ---- syndrome decoder for 128 originals (round 3).  The dense kernel above multiplies the received blocks with an N x 128
// matrix per frame (24 x 128 x 508 byte products at 24 erasures).  But what the received originals contribute to the received
// recovery rows is exactly what the ENCODER computes with the erased originals set to zero -- the XOR-convolution walk of
// gf_encode128_wg, 2.6 x cheaper per row than dense rows -- and the rest is small: syndrome_i = recovery_i ^ that, and the
// erased originals are Minv (N x N, closed form, from the planner) times the syndromes: 24 x 24 instead of 24 x 128 products.
// The kernel reads every received block once and writes it to its place in the payload on the way (the former scatter pass).
// One workgroup per (frame, half block) like the encoder; lanes = 4-byte columns.
struct Dec128Args {
    const uint8_t *rx;          // frames [nframes][128][512] in arrival order
    size_t rx_frame_bytes;
    const uint8_t *plan;        // [nframes] Dec128Plan
    const uint8_t *tab;         // 256 x 32 B multiplier tables
    const uint8_t *leaf_tables; // Karatsuba leaves of the encoder
    const uint8_t *fft_tables;  // constants of the additive-FFT encoder (gf_decode128_fft_kernel)
    uint8_t *payload_out;       // [nframes][127 x 508]
    size_t payload_frame_bytes;
    uint8_t *block0_out;        // optional [nframes][508]
    int nframes;
    int stagger, stagger_div;   // staggered start of the workgroups (Enc128Args::stagger)
    // fused plan (gf_decode128_fft_plan_kernel: the workgroup derives its frame's record itself, no gf_decode_plan_kernel launch)
    const uint8_t *indices;     // optional [nframes][128] block indices (device); NULL = the headers' blockIndex
    const uint8_t *explog;      // exp[512] + log[256] (uint16) of GF(256)
    int max_rows, strict;       // DecPlanArgs::max_rows (<= DEC128_MAXN here), ::strict
    unsigned *stats;            // DecPlanArgs::stats
    // no-copy mode (fused plan only; the Tx pipe in front of K5w's gather variant, InterpArgs::gmap): srcmap != NULL
    unsigned *srcmap;           // [nframes][128]: where original j of frame f lies (gf_decode128_fft.h: dec128_plan)
    uint8_t *restored;          // [nframes * restored_rows + 1] slots of 508 bytes: row t of frame f -> slot f * restored_rows + t; the last slot stays zero
    int restored_rows;
};
constexpr int DEC128_LDS_BYTES = 8 * KLEAVES * 20 + 256 * 32 + 33 * 64 * 4 + DEC128_MAXN * 64 * 4 + DEC128_PLAN_BYTES;

#ifndef DEC128_WPE
#define DEC128_WPE 4
#endif
// (OWN: the stand-alone kernel's form -- the LDS block is the function's own and the unit is the workgroup's index, as they were
// while this was the kernel itself; handed a pointer and an index the same code needs six more registers than it has)
template <bool OWN> __device__ __forceinline__ void gf_decode128_wg(const Dec128Args &a, int bx_in, unsigned char *lds_in)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds_own[OWN ? DEC128_LDS_BYTES : 16];
    unsigned char *ldsraw = OWN ? lds_own : lds_in;
    const int bx = OWN ? (int)blockIdx.x : bx_in;
    uint4_t *lt16 = reinterpret_cast<uint4_t *>(ldsraw);
    unsigned *lt4 = reinterpret_cast<unsigned *>(ldsraw + 8 * KLEAVES * 16);
    unsigned *tab = reinterpret_cast<unsigned *>(ldsraw + 8 * KLEAVES * 20);                       // all 256 constants: 8 dwords each
    unsigned (*ysum)[64] = reinterpret_cast<unsigned (*)[64]>(ldsraw + 8 * KLEAVES * 20 + 256 * 32); // convolution rows + parity
    unsigned (*syn)[64] = reinterpret_cast<unsigned (*)[64]>(ldsraw + 8 * KLEAVES * 20 + 256 * 32 + 33 * 64 * 4);
    Dec128Plan *pl = reinterpret_cast<Dec128Plan *>(ldsraw + 8 * KLEAVES * 20 + 256 * 32 + 33 * 64 * 4 + DEC128_MAXN * 64 * 4);
    const int tid = threadIdx.x;
    const int fr = bx >> 1;
    // (tables and plan records are 32- / 16-byte aligned: whole 16-byte loads)
    // (copy loops: hipcc compiles every iteration as load, wait, write -- left so in this fallback kernel: with all loads first the
    // FFT decoder's non-fused kernel, which holds this walk as its in-launch fallback, spills ten registers more)
    for (int i = tid; i < 8 * KLEAVES; i += GF_NT) {
        const uint4_t *src = reinterpret_cast<const uint4_t *>(a.leaf_tables) + (size_t)i * 2;
        lt16[i] = src[0];
        lt4[i] = reinterpret_cast<const unsigned *>(src + 1)[0];
    }
    for (int i = tid; i < 256 * 2; i += GF_NT) reinterpret_cast<uint4_t *>(tab)[i] = reinterpret_cast<const uint4_t *>(a.tab)[i];
    for (int i = tid; i < DEC128_PLAN_BYTES / 16; i += GF_NT)
        reinterpret_cast<uint4_t *>(pl)[i] = reinterpret_cast<const uint4_t *>(a.plan + (size_t)fr * DEC128_PLAN_BYTES)[i];
    for (int i = tid; i < 33 * 64; i += GF_NT) (&ysum[0][0])[i] = 0;
    __syncthreads();

    const int w = tid >> 6, lane = tid & 63;
    const int col = (bx & 1) * 64 + lane;
    const bool live = col < 127;
    const unsigned *rx = reinterpret_cast<const unsigned *>(a.rx + (size_t)fr * a.rx_frame_bytes) + 1 + (live ? col : 0);
    unsigned *pay = reinterpret_cast<unsigned *>(a.payload_out + (size_t)fr * a.payload_frame_bytes) + (live ? col : 0);
    unsigned *b0 = a.block0_out ? reinterpret_cast<unsigned *>(a.block0_out + (size_t)fr * 508) + (live ? col : 0) : nullptr;
    const int N = pl->n;
    const int npairs = N > 0 && !pl->m1 ? (pl->maxrow >> 5) + 1 : 1; // (M1 and plain copies: one walk for the parity / the copy)

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
#pragma unroll
            for (int i = 0; i < KN; ++i) {
                const int pos = pl->inv[KN * cb + i];
                v[i] = (pos >= 0 && live) ? SDRHIP_STREAM_LOAD(rx + (size_t)pos * 128) : 0u;
            }
            if (tp == 0 && live) {
                // the received originals go to their places (getSlotData's layout: blocks 1..127 back to back, block 0 apart)
#pragma unroll
                for (int i = 0; i < KN; ++i) {
                    const int j = KN * cb + i;
                    if (pl->inv[j] >= 0) {
                        if (j >= 1) pay[(size_t)(j - 1) * 127] = v[i];
                        else if (b0) b0[0] = v[i];
                    }
                }
            }
            if (N == 0) continue; // (copy only)
#pragma unroll
            for (int i = 0; i < KN; i += 2) p = x3(p, v[i], v[i + 1]);
            if (pl->m1) continue; // (parity only)
            const int t0 = (2 * tp) ^ cb, t1 = (2 * tp + 1) ^ cb;
            conv_block2(v, y0, y1, lds_addr(lt16 + t0 * KLEAVES), lds_addr(lt4 + t0 * KLEAVES), lds_addr(lt16 + t1 * KLEAVES), lds_addr(lt4 + t1 * KLEAVES));
        }
        if (N == 0) return; // (uniform)
        // the recovery rows this wave turns into syndromes (rows 32 tp + 8 w .. + 7, those that arrived): their loads go out here, in
        // front of the reduction and its barrier, not behind it (the phase was three dependent round trips: rowidx -> rpos -> HBM)
        int ri[8];
        unsigned rec[8];
        if (!pl->m1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 32 * tp + 8 * w + q;
                const int i = pl->rowidx[r & 127];
                ri[q] = (r < 128 && i != 255) ? i : -1;
                rec[q] = (ri[q] >= 0 && live) ? SDRHIP_STREAM_LOAD(rx + (size_t)pl->rpos[ri[q] & 31] * 128) : 0u;
            }
        }
        if (tp == 0) __hip_atomic_fetch_xor(&ysum[32][lane], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!pl->m1) {
#pragma unroll
            for (int i = 0; i < KN; ++i) {
                __hip_atomic_fetch_xor(&ysum[i][lane], y0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_xor(&ysum[KN + i][lane], y1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        // syndromes of the received recovery rows among rows 32 tp + 8 w .. + 7: recovery ^ (P ^ r * c_r)
        const unsigned P = ysum[32][lane];
        if (pl->m1) {
            if (w == 0) syn[0][lane] = live ? (P ^ SDRHIP_STREAM_LOAD(rx + (size_t)pl->rpos[0] * 128)) : 0u;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int rl = 8 * w + q, r = 32 * tp + rl;
                if (ri[q] >= 0) {
                    const uint4_t t = *reinterpret_cast<const uint4_t *>(&tab[r * 8]);
                    syn[ri[q]][lane] = rec[q] ^ P ^ kmul(ysum[rl][lane], t, tab[r * 8 + 4]);
                }
            }
        }
        __syncthreads();
        if (tp + 1 < npairs) {
#pragma unroll
            for (int q = 0; q < 8; ++q) ysum[8 * w + q][lane] = 0;
            __syncthreads();
        }
    }
    // erased originals = Minv x syndromes: wave w takes rows w, w + 4, ... (up to 8 of them); a syndrome dword is split into its
    // selector words once and multiplied by the constants of all the wave's rows
    unsigned acc[DEC128_MAXN / 4];
#pragma unroll
    for (int u = 0; u < DEC128_MAXN / 4; ++u) acc[u] = 0u;
    const int nmine = (N - w + 3) >> 2; // rows w + 4 u < N
    // four syndromes at a time, the next four (and the wave's constants for them: one 8-byte read per syndrome instead of a byte
    // read per product) already on their way from LDS: per syndrome the loop was two dependent LDS round trips (constants ->
    // table addresses -> tables), 24 times over = three quarters of this phase (tools/experiments_r04/dec_stamps.py)
    constexpr int CH = 4;
    unsigned sy[CH];
    uint2_t mc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        sy[k] = syn[k][lane];
        mc[k] = *reinterpret_cast<const uint2_t *>(&pl->minv[k * DEC128_MAXN + w * 8]);
    }
#pragma unroll 1
    for (int i0 = 0; i0 < N; i0 += CH) {
        unsigned syn_next[CH];
        uint2_t mc_next[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int i = (i0 + CH + k) & (DEC128_MAXN - 1); // (past the end: rows that exist and are not used)
            syn_next[k] = syn[i][lane];
            mc_next[k] = *reinterpret_cast<const uint2_t *>(&pl->minv[i * DEC128_MAXN + w * 8]);
        }
        // two syndromes at a time: their two products per row go into the accumulator with ONE 3-input XOR
#pragma unroll
        for (int k = 0; k < CH; k += 2) {
            if (i0 + k + 1 < N) {
                const Sel sl0 = make_sel(sy[k]), sl1 = make_sel(sy[k + 1]);
#pragma unroll
                for (int u = 0; u < DEC128_MAXN / 4; ++u) {
                    if (u < nmine) {
                        const unsigned m0 = ((u < 4 ? mc[k].x : mc[k].y) >> (8 * (u & 3))) & 0xffu;
                        const unsigned m1 = ((u < 4 ? mc[k + 1].x : mc[k + 1].y) >> (8 * (u & 3))) & 0xffu;
                        acc[u] = __builtin_amdgcn_bitop3_b32(acc[u], mulc(sl0, *reinterpret_cast<const uint4_t *>(&tab[m0 * 8]), tab[m0 * 8 + 4]),
                                                             mulc(sl1, *reinterpret_cast<const uint4_t *>(&tab[m1 * 8]), tab[m1 * 8 + 4]), 0x96);
                    }
                }
            } else if (i0 + k < N) {
                const Sel sl = make_sel(sy[k]);
#pragma unroll
                for (int u = 0; u < DEC128_MAXN / 4; ++u) {
                    if (u < nmine) {
                        const unsigned m = ((u < 4 ? mc[k].x : mc[k].y) >> (8 * (u & 3))) & 0xffu;
                        acc[u] ^= mulc(sl, *reinterpret_cast<const uint4_t *>(&tab[m * 8]), tab[m * 8 + 4]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) { sy[k] = syn_next[k]; mc[k] = mc_next[k]; }
    }
    if (live) {
#pragma unroll
        for (int u = 0; u < DEC128_MAXN / 4; ++u) {
            if (u < nmine) {
                const int yy = pl->ydst[w + 4 * u], y = yy & 0x7f;
                const unsigned val = (yy & 0x80) ? 0u : acc[u]; // (strict mode: a block the reference's copy-back would miss stays a hole)
                if (y >= 1) pay[(size_t)(y - 1) * 127] = val;
                else if (b0) b0[0] = val;
            }
        }
    }
}


__global__ __launch_bounds__(GF_NT, DEC128_WPE) void gf_decode128_kernel(Dec128Args a)
{
    gf_decode128_wg<true>(a, 0, nullptr);
}

// ... with the additive-FFT walk: a workgroup per frame; a frame that holds a recovery row beyond 31 takes the Karatsuba walk, its
// two units one after the other
#include "gf_decode128_fft.h"
__global__ __launch_bounds__(GF_NT, 4) void gf_decode128_fft_kernel(Dec128Args a)
{
    constexpr int LDSB = DEC128_FFT_LDS_BYTES > DEC128_LDS_BYTES ? DEC128_FFT_LDS_BYTES : DEC128_LDS_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char ldsraw[LDSB];
    const int fr = (int)blockIdx.x;
    const Dec128Plan *gp = reinterpret_cast<const Dec128Plan *>(a.plan + (size_t)fr * DEC128_PLAN_BYTES);
    if (gp->n > 0 && !gp->m1 && gp->maxrow >= FFT_MAX_ROWS) { // (workgroup-uniform)
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            gf_decode128_wg<false>(a, 2 * fr + u, ldsraw);
            __syncthreads();
        }
        return;
    }
    gf_decode128_fft_wg<false>(a, fr, ldsraw);
}
// ... with the frame's plan derived by the workgroup itself (dec_max_rows <= 32: no frame can need the Karatsuba walk or the dense kernel)
template <bool NOCOPY> __global__ __launch_bounds__(GF_NT, 4) void gf_decode128_fft_plan_kernel(Dec128Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char ldsraw[DEC128_FFT_FUSED_LDS_BYTES];
    gf_decode128_fft_wg<true, NOCOPY>(a, (int)blockIdx.x, ldsraw);
}

} // namespace

hipError_t launch_gf_apply(const GfArgs &a, hipStream_t stream)
{
    if (a.nframes <= 0 || a.rows <= 0) return hipSuccess;
    const int ngroups = a.frame_list ? a.ngroups : (a.nframes + GF_FRAMES_PER_GROUP - 1) / GF_FRAMES_PER_GROUP;
    if (ngroups <= 0) return hipSuccess;
    if (a.rows <= 16) {
        hipLaunchKernelGGL(gf_apply_kernel<4>, dim3(ngroups, 1), dim3(GF_NT), 0, stream, a);
    } else if (a.rows <= 24) {
        hipLaunchKernelGGL(gf_apply_kernel<6>, dim3(ngroups, 1), dim3(GF_NT), 0, stream, a);
    } else {
        hipLaunchKernelGGL(gf_apply_kernel<8>, dim3(ngroups, (a.rows + 31) / 32), dim3(GF_NT), 0, stream, a);
    }
    return hipGetLastError();
}

#ifdef FFT_STAMPS
extern "C" int sdrhip_debug_fft_stamps(unsigned long long *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fft_stamps), sizeof(g_fft_stamps)); }
#endif
hipError_t launch_gf_encode128(const Enc128Args &a, hipStream_t stream)
{
    if (a.nlist <= 0 || a.rows <= 0) return hipSuccess;
    if (a.use_fft && a.fft_tables && a.rows <= FFT_MAX_ROWS && a.half_units) hipLaunchKernelGGL(gf_encode128_fft_half_kernel, dim3(2 * a.nlist), dim3(128), 0, stream, a);
    else if (a.use_fft && a.fft_tables && a.rows <= FFT_MAX_ROWS) hipLaunchKernelGGL(gf_encode128_fft_kernel, dim3(a.nlist), dim3(GF_NT), 0, stream, a);
    else hipLaunchKernelGGL(gf_encode128_kernel, dim3(2 * a.nlist), dim3(GF_NT), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_gf_encode128_pack(const Enc128Args &a, const FrameArgs &f, int nstreams, hipStream_t stream)
{
    size_t blocks = (f.n - (f.skip_to - f.skip_from) + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const unsigned units = a.rows > 0 ? 2u * (unsigned)(a.nlist > 0 ? a.nlist : 0) : 0u;
    if (a.use_fft && a.fft_tables && a.rows > 0 && a.rows <= FFT_MAX_ROWS)
        hipLaunchKernelGGL(gf_encode128_fft_pack_kernel, dim3(units / 2u + (unsigned)blocks * (unsigned)nstreams), dim3(GF_NT), 0, stream, a, f, (unsigned)blocks);
    else
        hipLaunchKernelGGL(gf_encode128_pack_kernel, dim3(units + (unsigned)blocks * (unsigned)nstreams), dim3(GF_NT), 0, stream, a, f, (unsigned)blocks);
    return hipGetLastError();
}

hipError_t launch_block_scatter(const uint8_t *src, size_t src_frame_bytes, int src_pitch, int src_off, uint8_t *dst,
                                size_t dst_frame_bytes, int dst_pitch, int dst_off, const int16_t *map, int nblocks, int nframes,
                                hipStream_t stream)
{
    if (nframes <= 0) return hipSuccess;
    dim3 grid((nblocks + 7) / 8, nframes);
    hipLaunchKernelGGL(block_scatter_kernel, grid, dim3(256), 0, stream, src, src_frame_bytes, src_pitch, src_off, dst,
                       dst_frame_bytes, dst_pitch, dst_off, map, nblocks, nframes);
    return hipGetLastError();
}

hipError_t launch_fec_headers(const uint8_t *frames, size_t in_frame_bytes, uint8_t *rec, size_t out_frame_bytes, int nb_fec,
                              int first_index, int nframes, const int32_t *frame_list, int nlist, hipStream_t stream)
{
    if (!frame_list) nlist = nframes;
    if (nlist <= 0 || nb_fec <= 0) return hipSuccess;
    const int n = nlist * nb_fec;
    hipLaunchKernelGGL(fec_header_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, frames, in_frame_bytes, rec, out_frame_bytes,
                       nb_fec, first_index, nframes, frame_list, nlist);
    return hipGetLastError();
}

} // namespace sdrhip

namespace sdrhip {

hipError_t launch_fec_decode_device_plan(const DecodeBuffers &d, const uint8_t *rx, size_t rx_frame_bytes, const uint8_t *indices_dev,
                                         const uint8_t *explog, const uint8_t *tab, int nframes, uint8_t *payload_out,
                                         size_t payload_frame_bytes, uint8_t *block0_out, int max_rows, int strict, unsigned *stats, hipStream_t stream)
{
    if (nframes <= 0) return hipSuccess;
    if (max_rows < 1) max_rows = 1;
    if (max_rows > 128) max_rows = 128;
    if (d.plan2 && d.use_fft && d.fft_tables && d.fused_plan && max_rows <= DEC128_MAXN) {
        // one launch: every workgroup plans its own frame (gf_decode128_fft.h: dec128_plan_front / _back)
        Dec128Args k;
        k.rx = rx; k.rx_frame_bytes = rx_frame_bytes; k.plan = nullptr; k.tab = tab; k.leaf_tables = d.leaf_tables; k.fft_tables = d.fft_tables;
        k.payload_out = payload_out; k.payload_frame_bytes = payload_frame_bytes; k.block0_out = block0_out; k.nframes = nframes;
        k.stagger = d.stagger; k.stagger_div = d.stagger_div;
        k.indices = indices_dev; k.explog = explog; k.max_rows = max_rows; k.strict = strict; k.stats = stats;
        k.srcmap = d.srcmap; k.restored = d.restored; k.restored_rows = d.restored_rows;
        if (k.srcmap) hipLaunchKernelGGL(gf_decode128_fft_plan_kernel<true>, dim3(nframes), dim3(GF_NT), 0, stream, k);
        else hipLaunchKernelGGL(gf_decode128_fft_plan_kernel<false>, dim3(nframes), dim3(GF_NT), 0, stream, k);
        return hipGetLastError();
    }
    DecPlanArgs p;
    p.max_rows = max_rows; p.stats = stats; p.strict = strict;
    p.rx = rx; p.rx_frame_bytes = rx_frame_bytes; p.indices = indices_dev; p.explog = explog;
    p.coef = d.coef; p.pmap = d.pmap; p.zmap = d.zmap; p.pdst = d.pdst; p.zdst = d.zdst; p.nrec = d.nrec;
    p.payload_out = payload_out; p.payload_frame_bytes = payload_frame_bytes; p.block0_out = block0_out; p.nframes = nframes;
    p.plan2 = d.plan2;
    hipLaunchKernelGGL(gf_decode_plan_kernel, dim3(nframes), dim3(128), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (d.plan2) {
        // syndrome decoder: copies the received originals of EVERY frame and restores up to 32 erased ones; frames that need
        // more (nrec > 0 below) go on to the dense kernel
        Dec128Args k;
        k.rx = rx; k.rx_frame_bytes = rx_frame_bytes; k.plan = d.plan2; k.tab = tab; k.leaf_tables = d.leaf_tables; k.fft_tables = d.fft_tables;
        k.payload_out = payload_out; k.payload_frame_bytes = payload_frame_bytes; k.block0_out = block0_out; k.nframes = nframes;
        k.stagger = d.stagger; k.stagger_div = d.stagger_div;
        k.indices = nullptr; k.explog = nullptr; k.max_rows = max_rows; k.strict = strict; k.stats = stats;
        k.srcmap = nullptr; k.restored = nullptr; k.restored_rows = 0;
        if (d.use_fft && d.fft_tables) hipLaunchKernelGGL(gf_decode128_fft_kernel, dim3(nframes), dim3(GF_NT), 0, stream, k);
        else hipLaunchKernelGGL(gf_decode128_kernel, dim3(2 * nframes), dim3(GF_NT), 0, stream, k);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (max_rows <= DEC128_MAXN) return hipSuccess; // (no frame can need the dense kernel)
    } else {
        // received originals into place
        e = launch_block_scatter(rx, rx_frame_bytes, 512, 4, payload_out, payload_frame_bytes, 508, 0, d.pmap, 128, nframes, stream);
        if (e != hipSuccess) return e;
        if (block0_out) {
            e = launch_block_scatter(rx, rx_frame_bytes, 512, 4, block0_out, 508, 508, 0, d.zmap, 128, nframes, stream);
            if (e != hipSuccess) return e;
        }
    }
    // erased originals: rows 0 .. nrec[f] of every frame's matrix (workgroups beyond a frame's count leave at once)
    DecApplyArgs a;
    a.in = rx; a.in_frame_bytes = rx_frame_bytes; a.out = payload_out; a.out_frame_bytes = payload_frame_bytes; a.out_pitch = 508;
    a.coef = d.coef; a.dst = d.pdst; a.nrec = d.nrec; a.tab = tab; a.which = 0; a.nframes = nframes;
    const int groups = (nframes + 1) / 2;
    if (max_rows <= 16) hipLaunchKernelGGL(gf_decode_apply_kernel<4>, dim3(groups, 1), dim3(GF_NT), 0, stream, a);
    else hipLaunchKernelGGL(gf_decode_apply_kernel<6>, dim3(groups, (max_rows + 23) / 24), dim3(GF_NT), 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (block0_out) {
        a.out = block0_out; a.out_frame_bytes = 508; a.dst = d.zdst; a.which = 1;
        hipLaunchKernelGGL(gf_decode_apply_kernel<4>, dim3(groups, 1), dim3(GF_NT), 0, stream, a);
        e = hipGetLastError();
    }
    return e;
}

} // namespace sdrhip
