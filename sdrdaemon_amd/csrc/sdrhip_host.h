// sdrhip_host.h -- host-side internals shared by the .cpp files of libsdrhip.so
#pragma once
#include "../../include/sdrhip.h"
#include "sdrhip_internal.h"

#include <cstdint>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace sdrhip {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                                           \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return ::sdrhip::fail(SDRHIP_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// growable device scratch buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n);
    void release();
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// growable pinned host staging buffer; reuse waits for the previous upload that read from it
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    int reserve(size_t n);                 // also waits for the pending upload
    void mark(hipStream_t s);              // call after enqueueing the copies that read from p
    void release();
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// host-pointer calls up to this many input bytes skip the copy engine: the input is memcpy'd into pinned host memory that the
// kernel reads over PCIe itself, small outputs are written the same way (one launch + one sync instead of H2D + launch + D2H + sync:
// 42 -> ~28 us for one 65 536-sample TestSource block, tools/bench_host_block.py)
constexpr size_t SDRHIP_ZEROCOPY_MAX = (size_t)384 << 10; // (measured: 256 KiB calls 42 -> 29 us, 1 MiB calls 59 -> 77 us: the copy engine wins from there)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

void ctx_retain(sdrhip_ctx *c);
void ctx_release(sdrhip_ctx *c);

// meta blocks the decimator kernel writes for the fused Rx pipe (DecimArgs::meta_*)
struct RxMeta {
    int first, count;
    unsigned frame_count0;
    unsigned w[6]; // MetaDataFEC, 24 bytes: w[3], w[4] = time stamp of the call's first sample; the kernels advance it per frame and add the CRC
    uint64_t idx0; // call-relative decimated-sample index of the first started frame's first sample
    unsigned rate; // sample rate the stamps advance with (Hz, 0 = none)
};

// sampleSize after decimateN (Decimators.cpp:43-44, 112-113 ...): grows by log2decim, capped at 16 bits
inline unsigned decimated_sample_size(unsigned log2decim, unsigned ss)
{
    if (log2decim == 0) return ss;
    const unsigned target = 16 - log2decim, trunk = ss < target ? 0 : ss - target;
    return ss + log2decim - trunk;
}

// device-pointer cores (no argument validation, no staging)
int decimate_device(sdrhip_decimators *d, int log2decim, int fcpos, unsigned *sampleSize, const int16_t *in, size_t n_in,
                    size_t in_stride, int16_t *out, size_t out_stride, size_t *n_out, int frame_mode, int frame_blocks,
                    uint64_t frame_sample_base, const RxMeta *meta = nullptr, const Enc128Args *fuse = nullptr, bool *fused = nullptr,
                    bool coresident = false); // coresident: another kernel's workgroups share the CUs (ring depth 3, raised wave priority)
// the context's second stream (created on first use; non-blocking, so that it never synchronises with a NULL caller stream)
int ctx_stream2(sdrhip_ctx *c, hipStream_t *out);
bool decimate_mfma_applies(const sdrhip_decimators *d, int log2decim, int fcpos, size_t n_in);
struct InterpGather { // InterpArgs::gmap / grx / grest / gframes
    const unsigned *map;
    const uint8_t *rx, *restored;
    int frames;
};
bool interpolate_gather_ok(const sdrhip_ctx *c, int log2interp); // K5w serves this ratio (interpolate4 .. 64, interp_path != valu)
int interpolate_device(sdrhip_interpolators *p, int log2interp, const int16_t *in, size_t n_in, size_t in_stride, int16_t *out,
                       size_t out_stride, size_t *n_out, const InterpGather *gather = nullptr);
// frames/recovery on the device; recovery slots may be interleaved with the frames
// (rec_frame_bytes = stride between the recovery areas of consecutive frames)
// frame_list_dev (optional, device): groups of GF_FRAMES_PER_GROUP frame indices (-1 = none), ngroups of them
// lin (optional, Rx pipe): the payload of the frame slots >= first of every stream is taken from the stream-order output
// of the decimator and copied into the frame area by the encoder itself (Enc128Args::lin); only honoured when the
// structured encoder runs (nb_fec >= ENC128_MIN_ROWS): the caller checks fec_encode_fuses_framing()
struct EncodeLin {
    const unsigned *lin;
    size_t stride;
    int cap, first, pending;
};
// smallest number of recovery blocks the structured encoder serves: 13 for the Karatsuba walk (ENC128_MIN_ROWS: below, the generic
// matrix kernel's rows x 128 products are fewer than one 32-row tile's), enc_min_rows (default 1) for the additive FFT, whose 592
// products do not depend on the row count (profiles/r05_enc_rows.txt)
int enc128_min_rows(const sdrhip_ctx *c);
inline bool fec_encode_fuses_framing(const sdrhip_ctx *c, int nb_fec) { return nb_fec >= enc128_min_rows(c); }
int fec_encode_device(sdrhip_ctx *ctx, const uint8_t *frames, size_t frame_bytes, size_t nframes, int nb_fec, uint8_t *rec,
                      size_t rec_frame_bytes, const int32_t *frame_list_dev = nullptr, int ngroups = 0, const EncodeLin *lin = nullptr);
// the structured 128-original encoder on prepared arguments (the Rx pipe's deferred encode)
int fec_encode128_launch(sdrhip_ctx *ctx, const Enc128Args &k, hipStream_t on = nullptr); // on: another stream than the context's
// rx on the device, indices on the host; payload_out / block0_out on the device
struct DecodeSide { // decode on another stream than the context's, with the caller's own work buffers
    hipStream_t stream;
    DevBuf *plan, *idx;
    PinnedBuf *pin;
};
// no-copy decode (round 6, the Tx pipe in front of K5w): the received originals stay in rx, the restored blocks go to `restored`
// (nframes * rows + 1 slots of 508 bytes, the last one all zeros), `srcmap` ([nframes][128]) tells the interpolator where every block
// lies (InterpArgs::gmap).  Only when fec_decode_gather_ok(): the fused-plan FFT decoder is the one kernel that writes the map.
struct DecodeGather {
    unsigned *srcmap;
    uint8_t *restored;
    int rows;
};
bool fec_decode_gather_ok(const sdrhip_ctx *c);
int fec_decode_device(sdrhip_ctx *ctx, const uint8_t *rx, size_t rx_frame_bytes, const uint8_t *indices, size_t nframes,
                      uint8_t *payload_out, size_t payload_frame_bytes, uint8_t *block0_out, const DecodeSide *side = nullptr,
                      const DecodeGather *gather = nullptr);

} // namespace sdrhip

namespace sdrhip {
enum { DECIM_PATH_AUTO = 0, DECIM_PATH_VALU = 1, DECIM_PATH_MFMA = 2 };
// Kernel-path knobs of a context.  Read ONCE from the environment when the context is created (SDRHIP_DECIM_PATH =
// valu | mfma | auto, SDRHIP_MFMA_SPAN, SDRHIP_MFMA_MIN, SDRHIP_INTERP_PATH = valu | wave, SDRHIP_INTERP_SPAN); afterwards
// only sdrhip_ctx_set_option() changes them (tests, tools), under the context's lock.
struct CtxOptions {
    int decim_path = DECIM_PATH_AUTO;
    size_t mfma_span = 0;                  // forced span length of the matrix-core decimator (0 = planner's choice)
    size_t mfma_min = (size_t)1 << 22;     // smallest call (samples over all streams, decimate4 / 8) the matrix cores take in auto mode
    int interp_wave = 1;                   // interpolate4 .. 64: K5w, the barrier-free wave-private pipeline (interp_wave.h); 0 = K5
    size_t interp_span = 0;
    int mfma_ring = 4;                     // LDS-DMA ring depth of the decimate16 matrix-core kernel (3 or 4 groups; overlap mode always runs 3)
    // pipelined Rx pipe, where the encoder of the previous call's frames runs: 0 = its own launch behind the decimator, 1 = inside the
    // decimator's launch (rx_fused_kernel), 3 = its own launch on the context's SECOND stream, beside the decimator ("overlap")
    int rx_fused = 1;
    // Rx pipe on the matrix-core decimator: 1 = its waves store straight into the frame layout (round 5: no stream-order buffer, no
    // framing copy in the encoder, no K2 launch), 0 = stream order + K2 + the encoder's fused copy (the round-3 arrangement)
    int rx_direct = 1;
    int rx_window = 0;                     // frame window of the Rx pipe in calls (1..8); 0 = the default: 2, pipelined pipes 4 (sdrhip_rx_process)
    int tx_overlap = 1;                    // pipelined Tx pipe: 1 = decode of this batch on the second stream beside the interpolator of the previous one, 0 = one stream
    int enc_fft = 1;                       // structured 128-original encoder, rows <= 32: additive FFT (1) or the Karatsuba XOR-convolution walk (0)
    int enc_half = 0;                      // FFT encoder in half-frame workgroups (enc_units = half): finer units for the CUs, 128 threads each
    int enc_min_rows = 1;                  // ... the FFT from this many recovery blocks on (the generic matrix kernel below)
    int fec_stagger_mod = 0;               // 0: phase = resident round (workgroup / CUs); 1..16: phase = workgroup mod m; 100 + m: arrival rank on the CU mod m (experiment)
    int fec_stagger = 0;                   // staggered start of the FFT encoder's / decoder's workgroups, units of 1024 clocks per resident round (0 = off)
    // Tx pipe without the decoder's copy (built in round 6): immediate mode on K5w with the fused-plan decoder, the
    // interpolator gathers the received originals through the decoder's position map.  OFF by default: the decoder gets 11 us
    // faster per 1024 frames (0.0822 -> 0.0710 ms), the interpolator 62 us SLOWER (0.2354 -> 0.2974 ms: 48 load instructions per
    // wave instead of 8 in front of its store stream, profiles/r06_tx_gather_ab.txt)
    int tx_gather = 0;
    int dec_fused_plan = 1;                // batched decode with dec_max_rows <= 32 on the FFT decoder: the plan is made inside the decoder's launch (0: gf_decode_plan_kernel in front)
    int dec_syndrome = 1;                  // batched CM256 decode: syndrome kernel (1) or the dense matrix kernel alone (0)
    int dec_strict = 0;                    // batched decode delivers only what the reference's copy-back loop delivers (SDRdaemonFECBuffer.cpp:204-211)
    int dec_max_rows = 128;                // upper bound of the recovery blocks a received frame can have used (the sender's fecblk)
};
// what the last decimate / rx call of a bank actually launched (sdrhip_decimators_last_plan)
struct DecimPlanInfo {
    int path = 0;       // 0 none yet, DECIM_PATH_VALU, DECIM_PATH_MFMA
    size_t span = 0, head = 0, tail_start = 0;
    int wps = 0, npieces = 0, nseg = 0;
};
} // namespace sdrhip

struct sdrhip_ctx {
    // Every public entry point that works on this context (directly or through one of its handles) holds this lock
    // for its duration: the staging buffers, the decode-plan cache and the timing log are per context.  Calls on
    // one context therefore serialise; threads that want to overlap use one context each.
    std::recursive_mutex mtx;
    int device = 0;
    int n_cu = 256;                          // hipDeviceProp_t::multiProcessorCount (the planners size their grids from it)
    sdrhip::CtxOptions opt;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // the library's own second stream (created on first use: overlap modes of the pipes)
    std::atomic<int> refs{0}; // handles created on this context (they keep it alive)
    std::atomic<bool> dying{false}; // sdrhip_ctx_destroy was called while handles were still alive
    sdrhip::DevBuf in, out, aux, aux3;       // staging for SDRHIP_MEM_HOST calls and FEC work areas
    uint8_t *gf_tab = nullptr;               // 256 x 32 B multiplier tables (device)
    uint8_t *enc_matrix = nullptr;           // 128 x 128 encode matrix, rows 128..255 (device)
    uint8_t *enc_leaves = nullptr;           // Karatsuba leaf tables of the structured k = 128 encoder (device)
    uint8_t *enc_fft = nullptr;              // constants of the additive-FFT form of the same encoder (device)
    unsigned *decim_dump = nullptr;          // sink of the matrix-core decimator's warm-up stores (DecimArgs::mf_dump)
    unsigned *fused_roles = nullptr;         // role table of the fused Rx launch (rx_fused_kernel), fused_tag = its launch counter
    unsigned fused_tag = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t *gf_explog = nullptr;             // exp[512] + log[256] (uint16) of GF(256) for the decode planner (device)
    sdrhip::DevBuf dec_plan;                  // per-frame decode plans of the current batch (DecodeBuffers)
    unsigned *dec_stats = nullptr;            // device counters of the batched decoder: [0] frames that broke the dec_max_rows promise
    sdrhip::PinnedBuf pin;                       // per-call upload staging (maps, frame lists)
    sdrhip::PinnedBuf zin, zout;                 // zero-copy staging of small host-pointer calls (the kernels read / write pinned host memory)
    // per-kernel-class timing with hipEvents on `stream` (sdrhip_ctx_kernel_timing)
    bool ktime_on = false;
    int ktime_stride = 1;                     // kernel-class timers bracket every ktime_stride-th launch of a class (option "ktime_stride")
    int ktime_stride_cls[4] = {0, 0, 0, 0};   // ... per class when > 0 (option "ktime_stride_class" = "<class>:<stride>"; "ktime_stride" resets them)
    unsigned ktime_seen[4] = {0, 0, 0, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev[4];
};

namespace sdrhip {
struct CtxLock {
    std::lock_guard<std::recursive_mutex> g;
    explicit CtxLock(sdrhip_ctx *c) : g(c->mtx) {}
};
// RAII: brackets the launches of one kernel class with events when timing is enabled
struct KTimer {
    sdrhip_ctx *c;
    int cls;
    hipEvent_t e1 = nullptr;
    hipStream_t st;
    KTimer(sdrhip_ctx *ctx, int kernel_class, hipStream_t on = nullptr) : c(ctx), cls(kernel_class), st(on ? on : ctx->stream)
    {
        if (!c->ktime_on) return;
        // (an event pair around a launch costs the stream ~2.5 us: timing every launch of a two-launch step took 3 % off the step)
        if (cls >= 0 && cls < 4) {
            const int stride = c->ktime_stride_cls[cls] > 0 ? c->ktime_stride_cls[cls] : c->ktime_stride;
            if (stride > 1 && (c->ktime_seen[cls]++ % (unsigned)stride) != 0) return;
        }
        hipEvent_t e0 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess) return;
        if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); e1 = nullptr; return; }
        (void)hipEventRecord(e0, st);
        c->kev[cls].push_back(std::make_pair(e0, e1));
    }
    ~KTimer()
    {
        if (e1) (void)hipEventRecord(e1, st);
    }
};
} // namespace sdrhip
