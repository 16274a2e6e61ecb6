/*
 * sdrhip.h -- C ABI of libsdrhip.so: the MI355X (gfx950) engine behind sdrdaemon's
 * data-parallel hot path (integer half-band decimators / interpolators and the CM256
 * Cauchy-MDS GF(256) block-erasure code).
 *
 * This is the drop-in boundary.  Every entry point states the reference interface it
 * replaces (file:line relative to the f4exb/sdrdaemon tree).  Plain pointers and sizes
 * only; no C++ or torch types.  All functions return 0 on success or a negative
 * SDRHIP_E* code; sdrhip_last_error() gives the message of the calling thread's last
 * failure.
 *
 * Threading: every entry point takes its context's (recursive) lock, so calls on the handles
 * of one context serialise and may come from any thread; handles of different contexts run
 * concurrently (objects that live on different threads in the reference own a context each).
 * Memory: every data pointer is either host memory (SDRHIP_MEM_HOST: the library
 * stages it through pinned buffers and copies back, synchronously) or device memory on
 * the context's GPU (SDRHIP_MEM_DEVICE: 16-byte aligned, work is enqueued on the
 * context's HIP stream and NOT synchronised -- call sdrhip_ctx_synchronize()).
 * IQ samples are interleaved little-endian {int16 re, int16 im} = IQSample
 * (SDRDaemon.h:52-70); sample counts are in IQ samples, not int16 words.
 */
#ifndef SDRHIP_H
#define SDRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRHIP_OK 0
#define SDRHIP_EINVAL (-1)   /* bad argument */
#define SDRHIP_ENOMEM (-2)   /* host or device allocation failed */
#define SDRHIP_EDEVICE (-3)  /* HIP runtime error (no GPU, launch failure ...) */
#define SDRHIP_EALIGN (-4)   /* device pointer / stride not 16-byte aligned */
#define SDRHIP_EDECODE (-5)  /* cm256 decode: duplicate original index / singular system */
#define SDRHIP_EBUSY (-6)    /* asynchronous entry: nothing to collect yet / every batch of the ring is in flight */

#define SDRHIP_MEM_HOST 0
#define SDRHIP_MEM_DEVICE 1

/* Downsampler::fcPos_t, Downsampler.h:29-33 */
#define SDRHIP_FC_INF 0
#define SDRHIP_FC_SUP 1
#define SDRHIP_FC_CEN 2

/* IntHalfbandFilterEO1 (x86 USE_SSE4_1 builds, centre tap x << 13, EO1.h:136-142) versus
 * IntHalfbandFilterDB (all other builds, (x + 1) << 13, DB.h:102-103): Decimators.h:24-28 */
#define SDRHIP_HB_EO1 0
#define SDRHIP_HB_DB 1

/* frame geometry, UDPSinkFEC.h:56-59,109 */
#define SDRHIP_UDPSIZE 512
#define SDRHIP_NB_ORIGINAL 128
#define SDRHIP_BLOCK_BYTES 508
#define SDRHIP_SAMPLES_PER_BLOCK 127
#define SDRHIP_SAMPLES_PER_FRAME 16129

const char *sdrhip_last_error(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int sdrhip_device_count(void);

/* ------------------------------------------------------------------ context -- */
/* One per GPU and host thread of control.  hip_stream: a hipStream_t to enqueue on
 * (e.g. torch.cuda.current_stream().cuda_stream), or NULL for the device's null stream. */
typedef struct sdrhip_ctx sdrhip_ctx;
int sdrhip_ctx_create(int device, void *hip_stream, sdrhip_ctx **out);
/* Handles created on a context keep it alive: destroying the context first only marks it, the last
 * handle to be destroyed frees it. */
void sdrhip_ctx_destroy(sdrhip_ctx *ctx);
int sdrhip_ctx_synchronize(sdrhip_ctx *ctx);
/* Kernel-path knobs for tests and tools (production code never needs them).  The defaults are read from the environment
 * ONCE, when the context is created (SDRHIP_DECIM_PATH, SDRHIP_MFMA_SPAN, SDRHIP_MFMA_MIN, SDRHIP_INTERP_PATH,
 * SDRHIP_INTERP_SPAN, SDRHIP_RX_FUSED, SDRHIP_RX_DIRECT, SDRHIP_RX_WINDOW, SDRHIP_ENC_PATH, SDRHIP_ENC_MIN_ROWS, SDRHIP_MFMA_RING,
 * SDRHIP_TX_OVERLAP, SDRHIP_DEC_PATH, SDRHIP_DEC_PLAN, SDRHIP_FEC_STAGGER[_MOD]); keys:
 * "decim_path" = auto | valu | mfma, "mfma_span" / "mfma_min" / "interp_span" = decimal sample counts, "interp_path" = auto | wave | valu
 * (wave = K5w, the default from interpolate4 up; valu = K5), "rx_fused" = 0 | 1 | 2 | overlap (pipelined Rx: where the deferred encode
 * runs), "rx_direct" = 1 | 0 (Rx pipe on the matrix-core decimator: frame-layout stores, the default, or stream order + framing pass),
 * "enc_path" = fft | karatsuba (CM256 128 + R encoder and the batched decoder's walk: additive FFT for R <= 32, the default, or the
 * Karatsuba XOR-convolution walk), "enc_min_rows" = 1..32 (fewest recovery blocks the FFT encoder serves; below: the generic matrix
 * kernel), "enc_units" = frame | half (the FFT encoder's workgroup: a frame, the default, or one column half of a frame -- an experiment, no faster), "mfma_ring" = 4 | 3 | 2 (2: the two-waves-per-SIMD experiment, slower), "tx_overlap" = 1 | 0 (pipelined Tx: decode on the second stream), "dec_path" = syndrome | dense, "dec_plan" =
 * fused | kernel (batched decode with dec_max_rows <= 32 on the FFT decoder: each frame's plan is made by the decoder's own
 * workgroup, the default, or by the planning kernel in a launch of its own), "rx_window" = 0 | 1..8 (frame window of the Rx pipe in
 * calls; 0 = the default: 2, pipelined pipes 4), "fec_stagger" / "fec_stagger_mod" (experiment: staggered start of the FFT encoder's /
 * decoder's workgroups, default off; mod 0: phase = resident round, 1..16: workgroup mod m, 100 + m: the workgroup's arrival rank on its CU mod m), "ktime_stride" = 1..1024 / "ktime_stride_class" = "<class>:<stride>" (the kernel-class timers
 * of sdrhip_ctx_kernel_timing bracket every n-th launch, of all classes / of one).  Every setting computes the same bytes.  One knob is a promise, not a path: "dec_max_rows" = 1..128 (default 128), the most recovery
 * blocks a received frame can carry (the sender's fecblk, known from the meta block; a collector that counted the recovery blocks
 * of a batch -- adapters/UDPSourceFEC.h -- passes that count); <= 32 makes the batched decode ONE launch (the plan inside the
 * decoder, no fallback kernel).  The promise is checked on the device: a frame that carries MORE recovery blocks than
 * dec_max_rows is left as received (like an undecodable frame: missing originals read zero) and counted, see
 * sdrhip_ctx_get_counter("dec_rows_exceeded").  One knob selects behaviour: "dec_strict" = 0 | 1 (default 0).  The reference copies
 * back only the descriptors [128 - recoveryCount, 128) after cm256_decode (SDRdaemonFECBuffer.cpp:204-211: it relies on the
 * recovery blocks arriving last), so a block restored into a recovery block that arrived BEFORE some original is never copied
 * and stays a hole; by default the batched decoder delivers every restored block (a superset), with dec_strict = 1 exactly the
 * reference's frames, holes included. */
int sdrhip_ctx_set_option(sdrhip_ctx *ctx, const char *key, const char *value);
/* Event counters of the context, kept on the device (reading one synchronises the context's stream).  Keys:
 * "dec_rows_exceeded" = frames, since the context was created, that the batched decoder (sdrhip_fec_decode_frames,
 * sdrhip_tx_process) left unrepaired because they carried more recovery blocks than the dec_max_rows option allows. */
int sdrhip_ctx_get_counter(sdrhip_ctx *ctx, const char *key, uint64_t *value);
/* Average duration in milliseconds of the kernels launched between timing_begin and
 * timing_end on the context's stream, measured with hipEvents on that stream (what
 * bench.py's roofline object reports). */
int sdrhip_ctx_timing_begin(sdrhip_ctx *ctx);
int sdrhip_ctx_timing_end(sdrhip_ctx *ctx, float *elapsed_ms);
/* Per-kernel-class timing: while enabled, every launch of the class is bracketed by
 * hipEvents on the context's stream.  _read synchronises, returns the summed duration and
 * the number of launches since the last read, and clears the log.  An event pair costs the
 * stream ~2.5 us: sdrhip_ctx_set_option("ktime_stride", "N") brackets only every N-th launch of
 * a class (default 1; bench.py samples every 4th step of its timed region).  Classes: */
#define SDRHIP_K_DECIMATE 0    /* half-band decimator cascade kernel */
#define SDRHIP_K_INTERPOLATE 1 /* half-band interpolator cascade kernel */
#define SDRHIP_K_FEC_ENCODE 2  /* GF(256) matrix apply, encoder rows */
#define SDRHIP_K_FEC_DECODE 3  /* GF(256) matrix apply, decode matrices */
int sdrhip_ctx_kernel_timing(sdrhip_ctx *ctx, int enable);
int sdrhip_ctx_kernel_timing_read(sdrhip_ctx *ctx, int kernel_class, double *total_ms, unsigned *launches);

/* --------------------------------------------------------------- decimators -- */
/* A bank of `nstreams` independent `Decimators` objects (Decimators.h:32-71): per stream
 * the six half-band filter states m_decimator2..64 persist across calls, exactly like the
 * reference members.  hb_variant selects EO1 / DB rounding. */
typedef struct sdrhip_decimators sdrhip_decimators;
int sdrhip_decimators_create(sdrhip_ctx *ctx, int nstreams, int hb_variant, sdrhip_decimators **out);
void sdrhip_decimators_destroy(sdrhip_decimators *d);
int sdrhip_decimators_reset(sdrhip_decimators *d); /* back to the constructor's zero state */
/* What the bank's last cascade launch was (diagnostics: bench.py labels its roofline kernel with it, the parity tests
 * assert that they ran the benchmarked geometry).  path: 0 = the last call launched no cascade kernel (none yet, an empty call,
 * decimate1 and the filter-less decimate2 / 4_inf / _sup), 1 = VALU kernel (nseg segments per
 * stream), 2 = matrix-core kernel (per stream: VALU head [0, head), wps waves x 8 spans of `span` samples, VALU tail from
 * tail_start in npieces - 1 pieces). */
typedef struct sdrhip_decim_plan {
    int path, wps, npieces, nseg;
    size_t span, head, tail_start;
} sdrhip_decim_plan;
int sdrhip_decimators_last_plan(const sdrhip_decimators *d, sdrhip_decim_plan *out);

/* One Decimators::decimate<2^log2decim>_{inf,sup,cen}(sampleSize, in, out) call
 * (Decimators.h:35-53; dispatch of Downsampler::process, Downsampler.cpp:74-162) on each
 * stream of the bank.  log2decim 0..6 (0 = Downsampler's copy + decimate1 rescale,
 * Decimators.cpp:22-35), fcpos SDRHIP_FC_*.  Stream s reads n_in samples at
 * iq_in + 2*s*in_stride and writes *n_out = n_in >> log2decim samples (what the reference
 * resizes `out` to) at iq_out + 2*s*out_stride (strides in samples; ignored for one
 * stream).  *sampleSize (effective bits, 8..16) is updated as the reference's by-reference
 * argument.  As in the reference, floor(n_in / N) * N samples are consumed and the
 * remainder never enters the filter history. */
int sdrhip_decimate(sdrhip_decimators *d, int log2decim, int fcpos, unsigned *sampleSize, const int16_t *iq_in,
                    size_t n_in, size_t in_stride, int16_t *iq_out, size_t out_stride, size_t *n_out, int mem);

/* ------------------------------------------------------------ interpolators -- */
/* Bank of `Interpolators` objects (Interpolators.h:35-61): HB64, HB32, 4 x HB16 states. */
typedef struct sdrhip_interpolators sdrhip_interpolators;
int sdrhip_interpolators_create(sdrhip_ctx *ctx, int nstreams, sdrhip_interpolators **out);
void sdrhip_interpolators_destroy(sdrhip_interpolators *p);
int sdrhip_interpolators_reset(sdrhip_interpolators *p);
/* One Interpolators::interpolate<2^log2interp>_cen(in, out) call per stream
 * (Interpolators.h:38-43; Upsampler::process, Upsampler.cpp:52-84; log2interp 0 copies).
 * *n_out = n_in << log2interp.  log2interp = 6 reproduces the reference's
 * interpolate64_cen as it is (32 interpolated + 32 zero samples per input,
 * Interpolators.cpp:363-606). */
int sdrhip_interpolate(sdrhip_interpolators *p, int log2interp, const int16_t *iq_in, size_t n_in,
                       size_t in_stride, int16_t *iq_out, size_t out_stride, size_t *n_out, int mem);

/* -------------------------------------------------------------------- CM256 -- */
/* CM256::cm256_encoder_params / CM256::cm256_block as used at UDPSinkFEC.cpp:195-246 and
 * SDRdaemonFECBuffer.cpp:148-197 (layout-compatible with cm256cc's structs). */
typedef struct {
    int OriginalCount;
    int RecoveryCount;
    int BlockBytes;
} sdrhip_cm256_params;
typedef struct {
    void *Block;
    unsigned char Index;
} sdrhip_cm256_block;

/* CM256::cm256_encode(params, originals, recoveryBlocks) (UDPSinkFEC.cpp:246): host
 * pointers; originals taken positionally; recovery block r is Cauchy row OriginalCount + r. */
int sdrhip_cm256_encode(sdrhip_ctx *ctx, sdrhip_cm256_params params, const sdrhip_cm256_block *originals,
                        void *recoveryBlocks);
/* CM256::cm256_decode(params, blocks) (SDRdaemonFECBuffer.cpp:197): host pointers, in-place
 * contract of the library: recovered originals overwrite the recovery blocks' buffers and
 * their Index becomes the recovered original's index.  RecoveryCount == 1 takes the
 * library's XOR shortcut (see DESIGN.md "mirrored quirks"). */
int sdrhip_cm256_decode(sdrhip_ctx *ctx, sdrhip_cm256_params params, sdrhip_cm256_block *blocks);

/* Batched form of the encode section of UDPSinkFEC::transmitUDP (UDPSinkFEC.cpp:228-256):
 * frames = nframes x 128 super blocks of 512 bytes (header + 508 protected bytes);
 * recovery_out = nframes x nb_fec super blocks with header {frameIndex, 128 + r, 0}. */
int sdrhip_fec_encode_frames(sdrhip_ctx *ctx, const uint8_t *frames, size_t nframes, int nb_fec,
                             uint8_t *recovery_out, int mem);
/* Batched form of the decode section of SDRdaemonFECBuffer::writeAndRead
 * (SDRdaemonFECBuffer.cpp:143-213) + getSlotData (:72-75): rx = nframes x 128 super
 * blocks, the first 128 datagrams of each frame in arrival order (originals and recovery
 * mixed; the reference relies on recovery blocks arriving last, :210); payload_out =
 * nframes x 127 x 508 bytes (blocks 1..127 in place, i.e. 16129 IQ samples per frame);
 * block0_out (may be NULL) = nframes x 508 bytes (the meta block).  The block indices are
 * header.blockIndex of the super blocks (:147), read on the device; `indices` (may be NULL;
 * nframes x 128 bytes, HOST memory in either mode) overrides them.  Planning (which
 * originals are missing, the inverse of the Cauchy block, the recovery matrix of every
 * frame) runs on the GPU: a batch may hold any number of distinct loss patterns and the
 * device-memory form never synchronises with the host. */
int sdrhip_fec_decode_frames(sdrhip_ctx *ctx, const uint8_t *rx, const uint8_t *indices, size_t nframes,
                             uint8_t *payload_out, uint8_t *block0_out, int mem);

/* ----------------------------------------------------------- TestSource bank -- */
/* A bank of `nstreams` TestSource devices (include/TestSource.h:29-116, sdmnbase/TestSource.cpp) producing their
 * 16-bit IQ samples straight into device memory: the input side of BASELINE configs 2-5 without an H2D copy.
 * Configuration = the reference's key=value string (TestSource.cpp:59-215: srate, freq, dfp, dfn, power, blklen,
 * fcpos, decim; same range checks, same error strings through sdrhip_last_error(), same quirks -- see
 * sdrhip_testsource.cpp).  The sample arithmetic is NOT the reference's float phasor (not reproducible:
 * -ffast-math, wrap bug :411-415) but an integer-exact NCO defined in testsource_kernels.hip and restated in
 * oracle/sdr_oracle.c.  No real-time pacing (the reference sleeps one block time per block, :418). */
typedef struct sdrhip_testsource sdrhip_testsource;
int sdrhip_testsource_create(sdrhip_ctx *ctx, int nstreams, sdrhip_testsource **out);
void sdrhip_testsource_destroy(sdrhip_testsource *ts);
/* TestSource::configure(parsekv::pairs_type&): kv = "key=value,key=value" (',' or '&' separated, parsekv.h:40-43);
 * stream = -1 configures every stream. */
int sdrhip_testsource_configure(sdrhip_testsource *ts, int stream, const char *kv);
/* get_sample_rate() / get_frequency() (TestSource.cpp:261-270), block length, forwarded decim / fcpos; any pointer may be NULL */
int sdrhip_testsource_get(const sdrhip_testsource *ts, int stream, uint32_t *sample_rate, uint32_t *frequency, int *block_length,
                          int *log2decim, int *fcpos);
/* the next n samples of every stream (stream s at iq_out + 2*s*out_stride), phase continuous across calls */
int sdrhip_testsource_read(sdrhip_testsource *ts, int16_t *iq_out, size_t n, size_t out_stride, int mem);

/* ------------------------------------------------------------ fused Rx pipe -- */
/* Bank of Rx chains: Downsampler::process (Downsampler.cpp:74-162) -> UDPSinkFEC::write
 * framing (UDPSinkFEC.cpp:79-191) -> encode section of transmitUDP (:228-256), i.e. what
 * sdrdaemonrx's main loop + writer + tx threads compute between source_buffer.pull() and
 * sendto() (sdrdaemonrx.cpp:579-663). */
typedef struct {
    int log2decim;                 /* decim=  0..6 */
    int fcpos;                     /* fcpos=  0..2 */
    int hb_variant;                /* SDRHIP_HB_EO1 / SDRHIP_HB_DB */
    unsigned sample_bits;          /* DeviceSource::get_sample_bits(), 8..16 */
    int nb_fec;                    /* fecblk= 0..128 */
    uint32_t center_frequency_khz; /* UDPSink::setCenterFrequency (kHz on the wire, UDPSink.h:93) */
    uint32_t sample_rate;          /* rate AFTER decimation, UDPSink::setSampleRate */
} sdrhip_rx_config;
typedef struct sdrhip_rx sdrhip_rx;
int sdrhip_rx_create(sdrhip_ctx *ctx, int nstreams, const sdrhip_rx_config *cfg, sdrhip_rx **out);
void sdrhip_rx_destroy(sdrhip_rx *rx);
/* Live reconfiguration between two sdrhip_rx_process calls, the way sdrdaemonrx applies a control
 * message (Downsampler::configure, Downsampler.cpp:32-67: decim / fcpos; UDPSink::setNbBlocksFEC,
 * setCenterFrequency, setSampleRate, sdrdaemonrx.cpp:300-340).  As in the reference the filter
 * states carry over (the six half-band instances are shared by every decimateN entry point), the
 * frame being filled keeps the meta block it was started with and is encoded with the fecblk value
 * in force when it completes (UDPSinkFEC.cpp:160-165).  hb_variant cannot change. */
int sdrhip_rx_reconfigure(sdrhip_rx *rx, const sdrhip_rx_config *cfg);
/* Feeds n_in device-rate samples per stream.  Completed frames of stream s are written to
 * frames_out + s*frame_stride_bytes as (128 + nb_fec) super blocks of 512 bytes each,
 * frame after frame; *n_frames (per stream, identical for all streams) is the number of
 * frames completed by this call.  tv_sec/tv_usec = the time of the call's FIRST sample; the
 * meta block of every frame STARTED by this call carries that time advanced by the sample
 * clock to the frame's first sample: + floor(p * 10^6 / sample_rate) microseconds for a frame
 * that starts p decimated samples into the call (sample_rate = the configured rate of the
 * frame stream; 0 = no advance), and the CRC-32 over it (the reference calls gettimeofday
 * when it opens a frame, UDPSinkFEC.cpp:90-115; a batched call opens many at once).
 * frames_out must hold sdrhip_rx_max_frames(rx, n_in) frames per stream. */
int sdrhip_rx_process(sdrhip_rx *rx, const int16_t *iq_in, size_t n_in, size_t in_stride, uint32_t tv_sec,
                      uint32_t tv_usec, uint8_t *frames_out, size_t frame_stride_bytes, size_t *n_frames,
                      int mem);
size_t sdrhip_rx_max_frames(const sdrhip_rx *rx, size_t n_in);
/* Pipelined mode (off by default).  The reference's sink is asynchronous as well: UDPSinkFEC::write returns at once and the
 * transmit thread encodes and sends a frame later (UDPSinkFEC.cpp:193-211).  With on != 0 a sdrhip_rx_process call DELIVERS
 * (frames_out, *n_frames, sdrhip_rx_frames_view) the frames that the PREVIOUS call completed; their recovery blocks are computed
 * by encoder workgroups that ride in this call's decimator launch (one launch instead of two, the encoder fills the issue
 * slots the decimator's waves leave empty).  Same bytes, one call later.  sdrhip_rx_flush encodes and delivers the frames the
 * last call completed (end of stream, before switching the mode off, and before a sdrhip_rx_reconfigure that changes fecblk:
 * the waiting frames carry the old frame size; reconfigure refuses otherwise).  frames_out of a pipelined call must hold
 * sdrhip_rx_max_frames() frames per stream.
 * What it is for: the reference's delivery semantics, and an A / B partner.  It is NOT the fast path on an MI355X: both kernels run
 * at the board's power cap, co-resident they take the sum of their times (DESIGN.md "Whole pipes": 0.296-0.299 ms per step of the
 * headline bank against 0.260-0.276 in the default, immediate mode); a pipelined pipe also keeps the stream-order arrangement
 * (context option "rx_direct" applies to immediate pipes). */
int sdrhip_rx_set_pipelined(sdrhip_rx *rx, int on);
int sdrhip_rx_flush(sdrhip_rx *rx, uint8_t *frames_out, size_t frame_stride_bytes, size_t *n_frames, int mem);
/* the decimator launch of the last sdrhip_rx_process call (see sdrhip_decimators_last_plan) */
int sdrhip_rx_last_plan(const sdrhip_rx *rx, sdrhip_decim_plan *out);
/* Zero-copy alternative for device-side consumers (what transmitUDP does when it sends straight
 * from m_txBlocks, UDPSinkFEC.cpp:259-282): call sdrhip_rx_process with frames_out = NULL and
 * mem = SDRHIP_MEM_DEVICE, then read the n_frames finished frames of stream s at
 * base + s * stream_stride_bytes (device memory, frame after frame).  The view stays valid until
 * the next sdrhip_rx_process / sdrhip_rx_destroy on this handle. */
int sdrhip_rx_frames_view(const sdrhip_rx *rx, const uint8_t **base, size_t *stream_stride_bytes, size_t *n_frames);

/* Asynchronous host-pointer entry.  The reference's Rx chain is asynchronous end to end (source thread -> source_buffer ->
 * Downsampler::process -> output_buffer -> writer -> transmit thread, sdrdaemonrx.cpp:555-663): the frames of a block leave the
 * process long after it was pulled.  sdrhip_rx_submit takes one block of host samples per stream like sdrhip_rx_process
 * (SDRHIP_MEM_HOST) and returns at once: the block is appended to a pinned staging buffer -- or used IN PLACE when it lies in
 * sdrhip_host_alloc memory, which the caller then leaves untouched until the batch is collected -- and every `blocks` blocks go
 * out as one upload + launch + download on the context's stream.  sdrhip_rx_collect returns the finished frames of the OLDEST
 * batch ((128 + nb_fec) super blocks per frame, stream s at frames_out + s * frame_stride_bytes; frames_out has room for
 * max_frames frames per stream: a batch that holds more stays uncollected, *n_frames says how many, the call returns
 * SDRHIP_EINVAL; sdrhip_rx_max_frames() of the batch's samples bounds them -- in pipelined mode a batch delivers the frames the
 * PREVIOUS batch completed).  SDRHIP_OK always means: ONE batch was collected (*n_frames of it, possibly 0: blocks shorter than a
 * frame); SDRHIP_EBUSY: none was -- nothing submitted, or (wait = 0) the oldest batch is still in flight or being filled; wait = 1
 * blocks, outside the context lock (another thread may go on submitting), and launches a partly filled batch as it is (end of
 * stream).  At most `depth` batches are in flight; sdrhip_rx_submit
 * returns SDRHIP_EBUSY when the ring is full.  Defaults (no sdrhip_rx_set_async call): depth 4, one block per batch.  tv_sec /
 * tv_usec of a batch = those of its first block (frames are stamped by the sample clock from there, see sdrhip_rx_process).
 * Do not mix sdrhip_rx_process calls into a submit / collect sequence while batches are in flight. */
int sdrhip_rx_set_async(sdrhip_rx *rx, int depth, int blocks);
int sdrhip_rx_submit(sdrhip_rx *rx, const int16_t *iq_in, size_t n_in, size_t in_stride, uint32_t tv_sec, uint32_t tv_usec);
int sdrhip_rx_collect(sdrhip_rx *rx, uint8_t *frames_out, size_t frame_stride_bytes, size_t max_frames, size_t *n_frames, int wait);
/* Pinned host memory for the source side (the buffers a DeviceSource pushes): blocks submitted from it skip the staging copy. */
void *sdrhip_host_alloc(sdrhip_ctx *ctx, size_t bytes);
void sdrhip_host_free(sdrhip_ctx *ctx, void *p);

/* ------------------------------------------------------------ fused Tx pipe -- */
/* Bank of Tx chains: SDRdaemonFECBuffer decode (SDRdaemonFECBuffer.cpp:143-213) ->
 * getSlotData (:72-75) -> Upsampler::process (Upsampler.cpp:52-84), i.e. what
 * sdrdaemontx's main loop computes between recvfrom() and sink_buffer.push()
 * (sdrdaemontx.cpp:449-498).  rx = per stream nframes x 128 received super blocks (arrival
 * order); iq_out receives nframes * 16129 << log2interp samples per stream. */
typedef struct sdrhip_tx sdrhip_tx;
int sdrhip_tx_create(sdrhip_ctx *ctx, int nstreams, int log2interp, sdrhip_tx **out);
void sdrhip_tx_destroy(sdrhip_tx *tx);
/* Upsampler::configure's `interp` key (Upsampler.cpp:31-50) between two sdrhip_tx_process calls, the way
 * sdrdaemontx applies a control message (sdrdaemontx.cpp:381); the interpolator histories carry over. */
int sdrhip_tx_reconfigure(sdrhip_tx *tx, int log2interp);
int sdrhip_tx_process(sdrhip_tx *tx, const uint8_t *rx, const uint8_t *indices, size_t nframes,
                      size_t rx_stride_bytes, int16_t *iq_out, size_t out_stride, size_t *n_out, int mem);
/* Pipelined mode (off by default).  The reference's Tx chain lives with one frame of latency already: SDRdaemonFECBuffer hands a
 * frame out when the NEXT frame's first block arrives (SDRdaemonFECBuffer.cpp:133-139), and a reader thread keeps receiving while
 * the main loop interpolates (sdrdaemontx.cpp:449-498).  With on != 0 a sdrhip_tx_process call decodes ITS batch on the
 * context's second stream (planner + syndrome decoder: VALU / LDS-latency work) while the first stream interpolates the batch
 * the PREVIOUS call decoded (store-bound), and DELIVERS that previous batch: iq_out / out_stride / *n_out describe the previous
 * batch's samples (sdrhip_tx_pending_samples() per stream before the call; 0 after the first call or a flush -- iq_out may then
 * be NULL).  Same samples, one call later; a waiting batch keeps the interpolation factor it was handed in with.
 * sdrhip_tx_flush delivers the batch the last call decoded (end of stream, before switching the mode off).  A DEVICE rx buffer
 * of a pipelined call is read by the second stream after the call returns: leave it untouched until the next
 * sdrhip_tx_process / sdrhip_tx_flush on this handle has returned.  (Context option "tx_overlap" = 0: the same one-call-late
 * delivery with both kernels on the first stream, the A / B partner.) */
int sdrhip_tx_set_pipelined(sdrhip_tx *tx, int on);
int sdrhip_tx_flush(sdrhip_tx *tx, int16_t *iq_out, size_t out_stride, size_t *n_out, int mem);
size_t sdrhip_tx_pending_samples(const sdrhip_tx *tx);
/* Asynchronous host-pointer entry, the Tx twin of sdrhip_rx_submit / sdrhip_rx_collect.  sdrdaemontx receives on a reader thread
 * while the main loop interpolates (sdrdaemontx.cpp:449-498) and its collector releases a frame one frame late
 * (SDRdaemonFECBuffer.cpp:133-139).  sdrhip_tx_submit takes ONE batch of received frames from host memory (rx, indices,
 * nframes, rx_stride_bytes as in sdrhip_tx_process; staged through pinned memory, or used in place when it lies in
 * sdrhip_host_alloc memory, which the caller then leaves untouched until the batch is collected), enqueues upload + decode +
 * interpolate + download on the context's stream and returns at once.  sdrhip_tx_collect returns the OLDEST batch: its
 * nframes * 16129 << log2interp samples per stream (stream s at iq_out + 2 * s * out_stride; iq_out has room for max_samples per
 * stream: a bigger batch stays uncollected, *n_out says how many, the call returns SDRHIP_EINVAL) and, when block0_out is not
 * NULL, the frames' meta blocks (super block 0: nstreams * nframes x 508 bytes, stream-major) -- with log2interp = 0 the two
 * together are exactly what SDRdaemonFECBuffer hands out per frame (getSlotData + the meta block, .cpp:72-110).  SDRHIP_OK: one
 * batch collected (*n_out samples per stream, *n_frames frames); SDRHIP_EBUSY: none -- nothing submitted, or (wait = 0) the oldest
 * batch is still in flight; wait = 1 blocks outside the context lock.  At most `depth` batches are in flight (default 4);
 * sdrhip_tx_submit returns SDRHIP_EBUSY when the ring is full.  The factor in force at submit time applies to the batch. */
int sdrhip_tx_set_async(sdrhip_tx *tx, int depth);
int sdrhip_tx_submit(sdrhip_tx *tx, const uint8_t *rx, const uint8_t *indices, size_t nframes, size_t rx_stride_bytes);
int sdrhip_tx_collect(sdrhip_tx *tx, int16_t *iq_out, size_t out_stride, size_t max_samples, uint8_t *block0_out, size_t *n_out,
                      size_t *n_frames, int wait);

#ifdef __cplusplus
}
#endif
#endif /* SDRHIP_H */
