// sdrhip_internal.h -- shared between the host side (sdrhip.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// loads of data that is streamed through once (IQ input, frames): non-temporal.  tools/dma_probe.hip on MI355X: 7.1 TB/s with
// `nt` against 6.2 TB/s with the default cache policy, register loads and LDS-DMA alike (stores: no difference).
#ifndef SDRHIP_NT
#define SDRHIP_NT 1
#endif
#if SDRHIP_NT
#define SDRHIP_STREAM_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define SDRHIP_STREAM_LOAD(ptr) (*(ptr))
#endif

namespace sdrhip {

// Per-stream half-band decimator state: for each of the six filter instances
// (m_decimator2..64, Decimators.h:56-70) the last 64 inputs, split the way the kernels
// consume them: plane p = comp * 2 + parity (comp 0 = I, 1 = Q; parity 0 = even input
// = first sample of a myDecimate pair, 1 = odd), 32 int32 entries each, oldest first.
constexpr int DEC_STAGES = 6;
constexpr int DEC_HIST = 32;
constexpr int DEC_STATE_WORDS = DEC_STAGES * 4 * DEC_HIST; // int32 words per stream

// Per-stream interpolator state: for each of the six instances (m_interpolator2..64,
// Interpolators.h:47-52) the last 32 inputs per component (ring of order/2 <= 32), oldest first.
constexpr int INT_STAGES = 6;
constexpr int INT_HIST = 32;
constexpr int INT_STATE_WORDS = INT_STAGES * 2 * INT_HIST;

struct DecimArgs {
    const int16_t *in;   // stream s at in + 2 * s * in_stride
    int16_t *out;        // stream s at out + 2 * s * out_stride (or frame layout, see frame_*)
    size_t in_stride;    // samples
    size_t out_stride;   // samples
    size_t n_used;       // raw samples consumed per stream (multiple of 2^log2decim)
    const int32_t *state_cur; // [nstreams][DEC_STATE_WORDS]
    int32_t *state_next;
    int nstreams;
    int nsub_per_seg;    // sub-chunks per segment
    int nseg;            // grid.x
    int bias;            // 0 EO1, 1 DB
    int norm, trunk;     // final `<< norm >> trunk`
    // frame-layout epilogue (fused Rx pipe): when frame_mode != 0 the decimated sample with
    // per-stream running index g = out_index + frame_sample_base goes to the payload of super
    // block 1 + (g % 16129) / 127 of frame g / 16129 (UDPSinkFEC.cpp:134-155); `out` then is a
    // byte pointer to the stream's first frame slot and out_stride its stride in BYTES / 4.
    int frame_mode;
    int frame_blocks;        // super blocks per frame slot (128 + nb_fec)
    uint64_t frame_sample_base; // samples already sitting in the first (partial) frame slot
    // meta blocks of the frames this call starts (UDPSinkFEC.cpp:87-132, 150-152): frame slots meta_first ..
    // meta_first + meta_count - 1 of every stream get block 0 = {header, 24-byte MetaDataFEC, zero fill} and
    // the {frameIndex, blockIndex, 0} headers of blocks 1..127; frameIndex = meta_frame_count0 + i (mod 2^16)
    int meta_first, meta_count;
    unsigned meta_frame_count0;
    unsigned meta_w[6];
    uint64_t meta_idx0;  // decimated-sample index (counted from the call's first sample) at which frame meta_first starts
    unsigned meta_rate;  // sample rate of the frame stream in Hz (0: every frame carries the call's time stamp), see frame_meta_words()
    // matrix-core launch (decim_mfma.hip): per stream the VALU code runs the head [0, mf_head) and the tail
    // [mf_tail_start, n_used) in pieces of mf_tail_seg samples (mf_npieces = 1 + tail pieces workgroups), the
    // matrix-core waves run mf_wps groups of 8 spans of mf_span raw samples from mf_head on
    size_t mf_head, mf_span, mf_tail_start, mf_tail_seg;
    int mf_wps, mf_npieces;
    // the nstreams x mf_npieces VALU pieces go to mf_piece_wgs workgroups: the first mf_piece_early of them take mf_piece_share
    // pieces each (they start with the launch, on the CUs the matrix-core workgroups leave free), every other one a single piece
    int mf_piece_wgs, mf_piece_early, mf_piece_share;
    unsigned *mf_dump;   // >= 1 KiB of device memory that swallows the stores of the warm-up period
    int mf_ring;         // LDS-DMA ring depth of the decimate16 kernel in groups: 4 (147 KiB per workgroup), 3 (108 KiB: room for another kernel)
    int mf_prio;         // 1: the matrix-core waves raise their issue priority (they share their SIMDs with another kernel's waves)
};

// MetaDataFEC of the fi-th frame a call starts (UDPSinkFEC.cpp:90-115: the reference takes gettimeofday() when it opens a
// frame and CRCs the first 20 bytes).  A batched call opens all its frames "at once", so the stamp of a frame is the
// call's stamp (base[3] = tv_sec, base[4] = tv_usec: the time of the call's first sample) advanced by the sample clock:
// frame fi starts idx0 + fi * 16129 samples into the call, i.e. floor(idx * 10^6 / rate) microseconds later.
// CRC-32 (boost::crc_32_type = reflected 0xEDB88320) of the stamped record: the CRC is affine over GF(2), so
// crc(record) = crc(record with a zero stamp) ^ XOR over the set stamp bits k of CRC_BIT[k]; the first term comes from the host
// (base[5]), CRC_BIT[k] = the zero-init CRC of the 20-byte message that has only stamp bit k set (compile-time table).  Lane k
// of a wave looks at bit k, six DPP / swizzle steps XOR-reduce: ~25 instructions per frame instead of a 160-step bit-serial
// loop (which doubled the framing kernel's time).  Must be called by whole waves.  (The test-side framer applies the same
// rule: DESIGN.md K2.)
#if defined(__HIPCC__) && __cplusplus >= 201703L // (the kernels' translation units: C++17; the host files are C++11)
struct CrcBitTable { unsigned c[64]; };
constexpr CrcBitTable make_crc_bit_table()
{
    CrcBitTable T{};
    for (int k = 0; k < 64; ++k) {
        unsigned crc = 0u;
        for (int byte = 0; byte < 20; ++byte) {
            unsigned v = 0u;
            if (byte >= 12 && byte == 12 + k / 8) v = 1u << (k % 8);
            crc ^= v;
            for (int b = 0; b < 8; ++b) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
        }
        T.c[k] = crc;
    }
    return T;
}
__device__ const CrcBitTable CRC_BIT = make_crc_bit_table();

__device__ __forceinline__ void frame_meta_words(const unsigned (&base)[6], uint64_t idx0, unsigned rate, int fi, unsigned (&w)[6])
{
    unsigned sec = base[3], usec = base[4];
    if (rate) {
        const uint64_t idx = idx0 + (uint64_t)fi * 16129u;
        const uint64_t dus = idx * 1000000ull / rate;
        const uint64_t ds = dus / 1000000ull;
        usec += (unsigned)(dus - ds * 1000000ull);
        sec += (unsigned)ds;
        if (usec >= 1000000u) { usec -= 1000000u; sec += 1u; }
    }
    w[0] = base[0]; w[1] = base[1]; w[2] = base[2]; w[3] = sec; w[4] = usec;
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned word = lane < 32 ? sec : usec;
    unsigned x = ((word >> (lane & 31)) & 1u) ? CRC_BIT.c[lane] : 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x ^= (unsigned)__shfl_xor((int)x, m, 64);
    w[5] = base[5] ^ x;
}
// the same record by ONE thread (no cross-lane step: every lane of a wave can form the record of a different frame at once)
__device__ __forceinline__ void frame_meta_words_thread(const unsigned (&base)[6], uint64_t idx0, unsigned rate, int fi, unsigned (&w)[6])
{
    unsigned sec = base[3], usec = base[4];
    if (rate) {
        const uint64_t idx = idx0 + (uint64_t)fi * 16129u;
        const uint64_t dus = idx * 1000000ull / rate;
        const uint64_t ds = dus / 1000000ull;
        usec += (unsigned)(dus - ds * 1000000ull);
        sec += (unsigned)ds;
        if (usec >= 1000000u) { usec -= 1000000u; sec += 1u; }
    }
    w[0] = base[0]; w[1] = base[1]; w[2] = base[2]; w[3] = sec; w[4] = usec;
    unsigned x = 0u;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        x ^= (0u - ((sec >> k) & 1u)) & CRC_BIT.c[k];
        x ^= (0u - ((usec >> k) & 1u)) & CRC_BIT.c[32 + k];
    }
    w[5] = base[5] ^ x;
}
#endif

// returns hipSuccess or the launch error
hipError_t launch_decimate(int log2decim, int fcpos, bool pack16, const DecimArgs &a, hipStream_t stream);
// picks nsub_per_seg / nseg for a call (host helper living next to the kernel's geometry)
void plan_decimate(int log2decim, int fcpos, size_t n_used, int nstreams, int *nsub_per_seg, int *nseg);

// matrix-core variant of the centred cascades: fills the mf_* fields, false when the call is too short (or the
// mode unsupported); span_override != 0 forces the span length (tests)
bool plan_decimate_mfma(int log2decim, int fcpos, size_t n_used, int nstreams, size_t span_override, int n_cu, DecimArgs *a);
hipError_t launch_decimate_mfma(int log2decim, bool pack16, const DecimArgs &a, hipStream_t stream);
struct Enc128Args;
// the same decimator launch (register-ring variant) with encoder workgroups for `e` behind it in the grid (fused Rx step)
// roles: SDRHIP_FUSED_ROLE_WORDS zero-initialised device words owned by the context, tag: 1, 2, 3 ... per launch on them
constexpr int SDRHIP_FUSED_ROLE_WORDS = 4096 + 8;
hipError_t launch_rx_fused(int log2decim, bool pack16, const DecimArgs &a, const Enc128Args &e, unsigned *roles, unsigned tag, hipStream_t stream);

// filter-less paths: log2decim 0 (decimate1) and inf/sup 2, 4 (Decimators.cpp:22-91,127-170)
hipError_t launch_decimate_simple(int log2decim, int fcpos, const int16_t *in, size_t in_stride, int16_t *out,
                                  size_t out_stride, size_t n_in, int nstreams, int norm, int trunk,
                                  hipStream_t stream);

// TestSource bank (testsource_kernels.hip): one record per stream and call
struct TestSourceParams {
    unsigned phase0; // NCO phase of the call's first sample (2^32 = one turn)
    unsigned inc;    // phase increment per sample
    int amp;         // peak amplitude, Q15
};
hipError_t launch_testsource(const int *table, const TestSourceParams *par, int16_t *out, size_t out_stride, size_t n, int nstreams,
                             hipStream_t stream);

// K2 (frame_kernels.hip): stream-order samples -> super blocks of the frame area
struct FrameArgs {
    const unsigned *in;   // [nstreams][in_stride] IQ dwords
    unsigned *out;        // frame area: stream s at out + s * out_stride dwords, slot 0 = the frame being filled
    size_t in_stride, out_stride;
    size_t n;             // samples per stream in this call
    size_t skip_from, skip_to;  // samples [skip_from, skip_to) are not copied (the encoder moves them, Enc128Args::lin)
    uint64_t frame_sample_base; // samples already in slot 0
    int frame_blocks;     // super blocks per frame slot (128 + nb_fec)
    int meta_first, meta_count; // as DecimArgs::meta_*
    unsigned meta_frame_count0;
    unsigned meta_w[6];
    uint64_t meta_idx0;
    unsigned meta_rate;
};
hipError_t launch_frame_pack(const FrameArgs &a, int nstreams, hipStream_t stream);

struct InterpArgs {
    const int16_t *in;
    int16_t *out;
    size_t in_stride, out_stride; // samples
    size_t n_in;                  // input samples per stream
    const int32_t *state_cur;     // [nstreams][INT_STATE_WORDS]
    int32_t *state_next;
    int nstreams;
    int nsub_per_seg, nseg;
    // (the Tx pipe's decoded payload -- 127 x 508 bytes per frame, contiguous -- is already the linear sample layout: `in`.)
    // Gather mode (round 6, K5w only; gmap != NULL): the Tx pipe's decoder does NOT copy the received originals; stream s is
    // gframes frames of 127 blocks of 127 samples, block b (1..127) of frame f lies where gmap[(s * gframes + f) * 128 + b] says:
    // bit 31 clear: super block slot (payload at +4) of the received frames grx, 512 bytes apart; bit 31 set: 508-byte slot of the
    // restored blocks grest (the decoder's output; its last slot is all zeros: blocks that never came and cannot be restored)
    const unsigned *gmap;
    const uint8_t *grx, *grest;
    int gframes;
};
hipError_t launch_interpolate(int log2interp, const InterpArgs &a, hipStream_t stream);
void plan_interpolate(int log2interp, size_t n_in, int nstreams, int *nsub_per_seg, int *nseg);
// K5w (interp_wave.h): wave-private pipelines (workgroups of one or four independent waves), blocks of 128 inputs; log2interp 2..6
void plan_interpolate_wave(int log2interp, size_t n_in, int nstreams, int n_cu, size_t seg_override, int *nsub_per_seg, int *nseg);
hipError_t launch_interpolate_wave(int log2interp, const InterpArgs &a, hipStream_t stream);

// frames are processed in groups that share one coefficient matrix (one frame per half-wave)
constexpr int GF_FRAMES_PER_GROUP = 2;

// GF(256) matrix apply: out[f][r][:] = XOR_j coef[f or 0][r][j] * in[f][src(j)][:]
struct GfArgs {
    const uint8_t *in;       // frames: [nframes][in_blocks][in_pitch] bytes
    uint8_t *out;            // [nframes][rows][out_pitch]
    const uint8_t *coef;     // [ngroups or 1][rows][cols]
    const uint8_t *tab;      // 256 x 32 byte multiplier tables (device)
    size_t in_frame_bytes, out_frame_bytes;
    int in_pitch, out_pitch; // bytes between consecutive blocks
    int in_off, out_off;     // byte offset of the 508 protected bytes inside a block slot
    int rows, cols;
    int coef_per_frame;      // 1: coef indexed by frame, 0: shared
    const int16_t *row_dst;  // optional [ngroups or 1][rows] destination block index, -1 = skip (else r)
    const int16_t *col_src;  // optional [ngroups or 1][cols] source block index (else j)
    int nframes;
    // frames are processed in groups of GF_FRAMES_PER_GROUP that share one coefficient matrix (index =
    // group when coef_per_frame, else 0): frame_list[group * GF_FRAMES_PER_GROUP + slot] (or -1), NULL = identity
    const int32_t *frame_list;
    int ngroups;
    // optional indirection: matrix slot of each group (pattern cache); matrices / row_dst tables are then
    // matrix_rows rows apart (0 = tightly packed, `rows` apart)
    const int32_t *group_cm;
    int matrix_rows;
};
hipError_t launch_gf_apply(const GfArgs &a, hipStream_t stream);

// structured (Karatsuba) encoder for OriginalCount = 128: frames of 128 super blocks (pitch 512,
// payload at +4) -> `rows` recovery super blocks (pitch 512, payload at +4)
struct Enc128Args {
    const uint8_t *in;
    uint8_t *out;
    const uint8_t *tab;             // 256 x 32 byte multiplier tables (device)
    const uint8_t *leaf_tables;     // [8][81][32] multiplier tables of the Karatsuba leaves of G_0..G_7 (device)
    const uint8_t *fft_tables;      // [192][32] butterfly / fold / row constants of the additive-FFT encoder (gf_encode128_fft.h, device)
    int use_fft;                    // 1: rows <= 32 run the additive-FFT encoder (context option enc_path), 0: the Karatsuba walk
    size_t in_frame_bytes, out_frame_bytes;
    int rows;                       // recovery blocks, 1..128
    int nframes;                    // frames addressable through in/out
    const int32_t *frame_list;      // optional list of frame indices (-1 = skip), nlist entries; NULL = 0..nlist-1
    int nlist;
    int gen_done, gen_cap;          // gen_done > 0: entry i of the list IS (i / gen_done) * gen_cap + i % gen_done (the Rx pipe's
                                    // frames: `gen_done` finished slots of each stream, streams gen_cap slots apart), no array
    // Rx pipe behind a stream-order decimator: the payload of super blocks 1..127 of frame slot f >= lin_first of stream s
    // (frame index s * lin_cap + f) is taken from lin[s][f * 16129 - lin_pending ...] instead of the frame area, and
    // written into the frame area on the way (UDPSinkFEC::write's copy, UDPSinkFEC.cpp:134-155, fused into the encoder);
    // block 0 and the headers are in place already.  lin == NULL: plain encode.
    const unsigned *lin;
    size_t lin_stride;              // dwords between streams
    int lin_cap, lin_first, lin_pending;
    int lin_straddle;               // 1: frame slot 0 (open when the call began, lin_pending samples in it) is completed from
                                    // lin[0 ..] by the encoder itself, K2 leaves it alone (launch_gf_encode128_pack)
    // Rx pipe with K2 in the SAME launch (launch_gf_encode128_pack): the meta blocks and frame indices of the frame slots
    // meta_first .. meta_first + meta_count - 1 of every stream (gen_* addressing) are not in memory yet when the encoder reads
    // block 0: it derives them itself, exactly as K2 writes them (frame_meta_words).  meta_count = 0: everything is in memory.
    int meta_first, meta_count;
    unsigned meta_frame_count0;
    unsigned meta_w[6];
    uint64_t meta_idx0;
    unsigned meta_rate;
    // staggered start (round 6): the launch is ONE round of resident workgroups that all load first and compute afterwards -- the
    // memory phase and the VALU phase do not overlap.  Workgroup i sleeps (i / stagger_div) * stagger units of 1024 clocks before
    // its loads (stagger_div = the number of CUs: the i-th workgroup a CU receives), so that the co-resident workgroups of a CU are
    // in different phases.  0 = off.
    int stagger, stagger_div;
    int half_units;                 // 1: the FFT encoder runs in half-frame workgroups (gf_encode128_fft_half_kernel; context option enc_units)
};
// the sleep in front of a workgroup's loads (Enc128Args::stagger, DecodeBuffers::stagger)
#if defined(__HIPCC__)
// arrival counters per CU (key = XCC_ID, SE / SH / CU of HW_ID), never reset: the workgroups a CU receives one after the other get
// consecutive ranks whatever the dispatcher's dealing is (stagger_div <= -100)
static __device__ unsigned g_fec_cu_rank[4096];
__device__ __forceinline__ int fec_stagger_phase(int unit, int stagger_div)
{
    if (stagger_div <= -100) {
        __shared__ int s_phase;
        if (threadIdx.x == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned key = ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);
            s_phase = (int)(atomicAdd(&g_fec_cu_rank[key], 1u) % (unsigned)(-stagger_div - 100));
        }
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(s_phase);
    }
    // stagger_div > 0: phase = the resident round (unit / CUs); < 0: phase = unit mod -stagger_div (consecutive workgroups on one CU)
    return stagger_div > 0 ? unit / stagger_div : unit % -stagger_div;
}
__device__ __forceinline__ int fec_stagger_sleep(int unit, int stagger, int stagger_div)
{
    if (stagger <= 0 || stagger_div == 0) return 0;
    const int ph = fec_stagger_phase(unit, stagger_div); // (workgroup-uniform)
    const int n = ph * stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    return ph;
}
#endif
// smallest number of recovery blocks the structured 128-original encoder is used for (below: the generic matrix kernel)
constexpr int ENC128_MIN_ROWS = 13;
hipError_t launch_gf_encode128(const Enc128Args &a, hipStream_t stream);
// encoder + K2 (framing residue: open frames, meta blocks, headers) of the same call in one launch
hipError_t launch_gf_encode128_pack(const Enc128Args &a, const FrameArgs &f, int nstreams, hipStream_t stream);
hipError_t launch_block_scatter(const uint8_t *src, size_t src_frame_bytes, int src_pitch, int src_off, uint8_t *dst,
                                size_t dst_frame_bytes, int dst_pitch, int dst_off, const int16_t *map, int nblocks, int nframes,
                                hipStream_t stream);
hipError_t launch_fec_headers(const uint8_t *frames, size_t in_frame_bytes, uint8_t *rec, size_t out_frame_bytes, int nb_fec,
                              int first_index, int nframes, const int32_t *frame_list, int nlist, hipStream_t stream);


// device-side decode planning (gf_kernels.hip): work buffers of a batch of nframes
constexpr size_t DECODE_PLAN2_BYTES = 1488; // sizeof(Dec128Plan), gf_kernels.hip
struct DecodeBuffers {
    uint8_t *coef;            // [nframes][128][128]
    int16_t *pmap, *zmap;     // [nframes][128]
    int16_t *pdst, *zdst;     // [nframes][128]
    int32_t *nrec;            // [nframes][2]
    uint8_t *plan2;           // [nframes][DECODE_PLAN2_BYTES] records of the syndrome decoder, NULL = dense path only
    const uint8_t *leaf_tables; // Karatsuba leaf tables of the 128-original encoder (the syndrome decoder walks the same tree)
    const uint8_t *fft_tables;  // constants of the additive-FFT encoder (gf_decode128_fft.h); NULL or use_fft = 0: the Karatsuba walk
    int use_fft;
    int stagger, stagger_div;   // staggered start of the FFT decoder's workgroups (see Enc128Args::stagger)
    int fused_plan;             // 1: frames that can carry at most DEC128_MAXN recovery blocks are planned by the decoder's own workgroups (one launch)
    // no-copy mode (only honoured by the fused-plan launch: the caller checks fec_decode_gather_ok()): see Dec128Args::srcmap
    unsigned *srcmap;
    uint8_t *restored;
    int restored_rows;
    static size_t bytes(size_t nframes) { return nframes * (128 * 128 + 4 * 128 * sizeof(int16_t) + 2 * sizeof(int32_t) + DECODE_PLAN2_BYTES) + 64; }
};
// plan + scatter + apply, all on the stream, no host synchronisation; max_rows = upper bound of the recovery blocks a
// frame can have used (128 when unknown); a frame that carries more is left as received and counted in stats[0]
hipError_t launch_fec_decode_device_plan(const DecodeBuffers &d, const uint8_t *rx, size_t rx_frame_bytes, const uint8_t *indices_dev,
                                         const uint8_t *explog, const uint8_t *tab, int nframes, uint8_t *payload_out,
                                         size_t payload_frame_bytes, uint8_t *block0_out, int max_rows, int strict, unsigned *stats, hipStream_t stream);

} // namespace sdrhip
