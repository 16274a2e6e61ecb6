/*
 * sdr_oracle.h -- CPU restatement ("oracle") of sdrdaemon's DSP/FEC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the timed CPU baseline.  The product path
 * (sdrdaemon_amd/, libsdrhip.so) never links or calls this code.
 *
 * Plain C99, written from the arithmetic specification in SURVEY.md section 7.1;
 * every function cites the reference file:line whose behaviour it restates
 * (paths relative to the reference tree).  No reference source text is copied.
 *
 * Pinning status:
 *   - DSP half (decimators / interpolators): PINNED.  oracle/Makefile builds the
 *     real reference sources (Decimators.cpp, Interpolators.cpp,
 *     HBFilterTraits.cpp, both the USE_SSE4_1 = EO1 and the plain = DB flavour)
 *     into oracle/_ref/ and tests/test_oracle_vs_ref.py diffs this restatement
 *     against them; golden vectors produced by the reference are committed
 *     under tests/golden/.
 *   - FEC half (CM256 / GF(256)): PARITY UNPINNED.  The arithmetic lives in the
 *     third-party library f4exb/cm256cc (C++ fork of catid/cm256, version not
 *     pinned by the reference: no submodule, no tag) which is absent from the
 *     reference tree and from this machine.  The restatement follows the
 *     published upstream algorithm (GF(2^8) polynomial 0x14D, generator 2,
 *     Cauchy element (y_j ^ x_0) / (x_i ^ y_j), x_0 = OriginalCount) and is
 *     anchored on the reference call sites (UDPSinkFEC.cpp:228-256,
 *     SDRdaemonFECBuffer.cpp:148-213) plus algebraic known-answer tests.
 */
#ifndef SDR_ORACLE_H
#define SDR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- DSP ---- */

enum { ORC_FC_INF = 0, ORC_FC_SUP = 1, ORC_FC_CEN = 2 }; /* Downsampler.h:29-33 fcPos_t */

/* Restates class Decimators (Decimators.h:32-71): six HB64 stage states that
 * persist across calls.  bias = 0 restates IntHalfbandFilterEO1 (USE_SSE4_1
 * builds), bias = 1 restates IntHalfbandFilterDB (all other builds). */
typedef struct orc_decimators orc_decimators;
orc_decimators *orc_decimators_new(int bias);
void orc_decimators_free(orc_decimators *d);
void orc_decimators_reset(orc_decimators *d);

/* One Decimators::decimate<2^log2>_{inf,sup,cen} call (log2 in 0..6; log2 = 0 is
 * decimate1 and copies in -> out).  Returns the size the reference resizes
 * `out` to (n_in >> log2); iq_out must hold that many samples.  *sampleSize is
 * updated exactly like the reference's by-reference argument. */
size_t orc_decimate(orc_decimators *d, int log2decim, int fcpos, unsigned *sampleSize,
                    const int16_t *iq_in, size_t n_in, int16_t *iq_out);

/* Restates class Interpolators (Interpolators.h:35-61): HB64, HB32, 4 x HB16. */
typedef struct orc_interpolators orc_interpolators;
orc_interpolators *orc_interpolators_new(void);
void orc_interpolators_free(orc_interpolators *p);
void orc_interpolators_reset(orc_interpolators *p);
/* One Interpolators::interpolate<2^log2>_cen call (log2 0..6; 0 copies, as
 * Upsampler::process does, Upsampler.cpp:54-57).  Returns n_in << log2. */
size_t orc_interpolate(orc_interpolators *p, int log2interp,
                       const int16_t *iq_in, size_t n_in, int16_t *iq_out);

/* ------------------------------------------------------------ framing ---- */

#define ORC_UDPSIZE 512           /* UDPSinkFEC.h:56 */
#define ORC_NB_ORIGINAL 128       /* UDPSinkFEC.h:57 */
#define ORC_BLOCK_BYTES 508       /* sizeof(ProtectedBlock), UDPSinkFEC.h:109-114 */
#define ORC_SAMPLES_PER_BLOCK 127 /* UDPSinkFEC.h:109 */
#define ORC_SAMPLES_PER_FRAME (127 * 127)

uint32_t orc_crc32(const void *data, size_t n); /* boost::crc_32_type == zlib CRC-32 */

/* MetaDataFEC, UDPSinkFEC.h:77-100 (packed, little endian, 24 bytes). */
#pragma pack(push, 1)
typedef struct {
    uint32_t center_frequency_khz;
    uint32_t sample_rate;
    uint8_t sample_bytes;
    uint8_t sample_bits;
    uint8_t nb_original_blocks;
    uint8_t nb_fec_blocks;
    uint32_t tv_sec;
    uint32_t tv_usec;
    uint32_t crc32;
} orc_meta;
#pragma pack(pop)

/* Restates the framing half of UDPSinkFEC::write (UDPSinkFEC.cpp:79-191): a
 * sample stream is cut into frames of 128 super blocks of 512 bytes; block 0
 * carries the meta data, blocks 1..127 carry 127 samples each.  The time stamp
 * (gettimeofday in the reference, :91) is injected by the caller. */
typedef struct {
    uint8_t cur[ORC_UDPSIZE]; /* m_superBlock */
    uint8_t slot[ORC_NB_ORIGINAL * ORC_UDPSIZE]; /* m_txBlocks[m_txBlocksIndex][0..127] */
    int tx_block_index;       /* m_txBlockIndex */
    int sample_index;         /* m_sampleIndex */
    uint16_t frame_count;     /* m_frameCount */
    uint32_t center_frequency_khz, sample_rate;
    uint8_t sample_bytes, sample_bits, nb_fec_blocks;
    uint32_t tv_sec, tv_usec;
    /* 0: every frame a write() opens carries (tv_sec, tv_usec) as it is (the test plays gettimeofday, UDPSinkFEC.cpp:91);
     * 1: (tv_sec, tv_usec) is the time of the write() call's FIRST sample and a frame opened p samples into the call is
     * stamped floor(p * 10^6 / sample_rate) microseconds later (the rule of the batched product, where one call opens many
     * frames: sdrhip_internal.h frame_meta_words) */
    int stamp_from_samples;
} orc_framer;
void orc_framer_init(orc_framer *f);
/* Feeds n samples; every completed frame is appended to frames_out as
 * 128 x 512 bytes.  Returns the number of frames completed by this call
 * (frames_out must have room for (pending + n) / 16129 + 1 frames). */
size_t orc_framer_write(orc_framer *f, const int16_t *iq, size_t n, uint8_t *frames_out);

/* Restates the encode section of UDPSinkFEC::transmitUDP (UDPSinkFEC.cpp:228-256)
 * for one frame: frame = 128 super blocks (512 B each); writes nb_fec recovery
 * super blocks (header {frameIndex, 128 + r, 0} + 508 recovery bytes). */
int orc_frame_encode(const uint8_t *frame, int nb_fec, uint8_t *recovery_superblocks);

/* ---------------------------------------------------------- GF / CM256 ---- */

uint8_t orc_gf_mul(uint8_t a, uint8_t b);
uint8_t orc_gf_div(uint8_t a, uint8_t b);
uint8_t orc_gf_exp(int i);
int orc_gf_log(uint8_t a);
/* Cauchy element of upstream GetMatrixElement(x_i, x_0, y_j). */
uint8_t orc_cm256_matrix_element(uint8_t x_i, uint8_t x_0, uint8_t y_j);
/* dst[i] ^= c * src[i] ; the timed inner loop (upstream gf256_muladd_mem). */
void orc_gf_muladd_mem(uint8_t *dst, uint8_t c, const uint8_t *src, size_t n);

typedef struct { /* CM256::cm256_encoder_params at the call sites */
    int OriginalCount;
    int RecoveryCount;
    int BlockBytes;
} orc_cm256_params;
typedef struct { /* CM256::cm256_block */
    void *Block;
    uint8_t Index;
} orc_cm256_block;

/* cm256_encode: originals taken positionally; recovery r = row 128 + r.
 * 0 on success, negative on bad arguments (same codes as upstream). */
int orc_cm256_encode(orc_cm256_params p, const orc_cm256_block *originals, void *recoveryBlocks);
/* cm256_decode: in-place contract of upstream (recovered data overwrites the
 * recovery block buffers, whose Index becomes the erased original's index).
 * Mirrors the upstream RecoveryCount == 1 XOR shortcut. */
int orc_cm256_decode(orc_cm256_params p, orc_cm256_block *blocks);

/* Restates SDRdaemonFECBuffer (SDRdaemonFECBuffer.cpp:72-250): one decoder
 * slot fed datagram by datagram. */
typedef struct {
    uint8_t frame[ORC_NB_ORIGINAL][ORC_BLOCK_BYTES];    /* m_decoderSlot.m_frame */
    uint8_t recovery[ORC_NB_ORIGINAL][ORC_BLOCK_BYTES]; /* m_recoveryBlocks */
    orc_cm256_block desc[ORC_NB_ORIGINAL];
    int block_count, recovery_count, decoded, meta_retrieved;
    int frame_head;
    int cur_nb_blocks, cur_nb_recovery, min_nb_blocks, max_nb_recovery;
} orc_fecbuffer;
void orc_fecbuffer_init(orc_fecbuffer *b);
/* SDRdaemonFECBuffer::writeAndRead: returns 1 when `data` received the previous
 * frame's 127 x 508 bytes (the first such emission is all zeros here; the
 * reference emits uninitialised memory). */
int orc_fecbuffer_write_and_read(orc_fecbuffer *b, const uint8_t *superblock, uint8_t *data,
                                 size_t *data_length);

#ifdef __cplusplus
}
#endif
/* TestSource bank: the integer NCO the product defines (see sdr_oracle.c) */
int orc_nco_cos_q30(unsigned idx12);
unsigned orc_nco_phase_inc(int64_t df, int64_t srate);
int orc_nco_amp_q15(int db);
unsigned orc_testsource_generate(unsigned phase0, unsigned inc, int amp_q15, size_t n, int16_t *iq_out);

#endif
