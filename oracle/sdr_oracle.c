/*
 * sdr_oracle.c -- CPU restatement of sdrdaemon's DSP/FEC hot path.
 * TEST INFRASTRUCTURE ONLY -- see sdr_oracle.h for the rules and the pinning
 * status (DSP: pinned against the compiled reference; FEC: PARITY UNPINNED).
 *
 * All integer arithmetic is done on uint32_t (wrap-around mod 2^32, which is
 * what the reference's int32 code compiles to and what its SSE4.1 kernel,
 * IntHalfbandFilterEO1i.h:55-69, does by construction) and re-interpreted as
 * int32 for the arithmetic right shifts.
 */
#include "sdr_oracle.h"

#include <stdlib.h>
#include <string.h>

#if defined(__SSSE3__)
#include <tmmintrin.h>
#endif

/* ------------------------------------------------------------------ taps -- */
/* (int32_t)(literal * (1 << 14)) of HBFilterTraits.cpp:25-31 (order 16),
 * :62-72 (order 32), :210-228 (order 64); hbShift = 14 (HBFilterTraits.h). */
static const int32_t C64[16] = {-7, 11, -20, 32, -49, 71, -101, 140,
                                -190, 256, -345, 469, -656, 978, -1698, 5201};
static const int32_t C32[8] = {-30, 63, -135, 261, -469, 830, -1605, 5176};
static const int32_t C16[4] = {-85, 380, -1246, 5041};
#define HB_SHIFT_M1 13 /* hbShift - 1, IntHalfbandFilterEO1.h:136-146 */

static inline int32_t asr(uint32_t v, unsigned s) { return ((int32_t)v) >> s; }

/* ------------------------------------------------------- decimator stage -- */
/* One IntHalfbandFilter{EO1,DB}<64> instance used as a decimator.  The ring
 * buffers of the reference (EO1.h:68-92, DB.h:74-77) hold exactly the last
 * 62 inputs before the current pair; that is the whole state. */
typedef struct {
    int32_t hist[2][62];
} dec_stage;

/* myDecimate over n_out pairs (EO1.h:34-42 + doFIR :100-147; DB.h:32-49 +
 * doFIR :79-107).  x[c] holds 2*n_out inputs per component, y[c] receives
 * n_out outputs.  Output k sits at n = 2k+1:
 *   acc = sum_i c[i]*(s[n-2i] + s[n-62+2i]) + ((s[n-31] + bias) << 13); o = acc >> 13 */
static void dec_stage_run(dec_stage *st, int bias, size_t n_out, int32_t *const x[2],
                          int32_t *const y[2])
{
    if (n_out == 0) return;
    size_t n_in = 2 * n_out;
    int32_t *ext = (int32_t *)malloc((62 + n_in) * sizeof(int32_t));
    for (int c = 0; c < 2; ++c) {
        memcpy(ext, st->hist[c], 62 * sizeof(int32_t));
        memcpy(ext + 62, x[c], n_in * sizeof(int32_t));
        for (size_t k = 0; k < n_out; ++k) {
            const int32_t *s = ext + 62 + 2 * k + 1; /* s[0] = newest sample of the pair */
            uint32_t acc = 0;
            for (int i = 0; i < 16; ++i)
                acc += ((uint32_t)s[-2 * i] + (uint32_t)s[-62 + 2 * i]) * (uint32_t)C64[i];
            acc += ((uint32_t)s[-31] + (uint32_t)bias) << HB_SHIFT_M1;
            y[c][k] = asr(acc, HB_SHIFT_M1);
        }
        memcpy(st->hist[c], ext + n_in, 62 * sizeof(int32_t));
    }
    free(ext);
}

struct orc_decimators {
    dec_stage st[6]; /* m_decimator2 .. m_decimator64, Decimators.h:56-70 */
    int bias;
};

orc_decimators *orc_decimators_new(int bias)
{
    orc_decimators *d = (orc_decimators *)calloc(1, sizeof(*d)); /* ctor zero-fills, EO1.h:171-188 */
    if (d) d->bias = bias ? 1 : 0;
    return d;
}
void orc_decimators_free(orc_decimators *d) { free(d); }
void orc_decimators_reset(orc_decimators *d) { memset(d->st, 0, sizeof(d->st)); }

static inline int16_t final_shift(int32_t v, unsigned norm, unsigned trunk)
{
    /* `x << norm_shift >> trunk_shift` on int32, then implicit FixReal conversion
     * (Decimators.cpp:112-113, SDRDaemon.h:59-60): plain truncation. */
    return (int16_t)(asr((uint32_t)v << norm, trunk));
}

size_t orc_decimate(orc_decimators *d, int log2decim, int fcpos, unsigned *sampleSize,
                    const int16_t *in, size_t n_in, int16_t *out)
{
    const unsigned L = (unsigned)log2decim;
    if (L == 0) {
        /* Downsampler::process m_decim == 0 (Downsampler.cpp:76-80) = copy + decimate1
         * (Decimators.cpp:22-35): left shift to 16 bits, sampleSize unchanged. */
        unsigned ss = *sampleSize;
        if (ss < 16) {
            unsigned norm = 16 - ss;
            for (size_t i = 0; i < 2 * n_in; ++i) out[i] = (int16_t)((uint32_t)(int32_t)in[i] << norm);
        } else if (out != in) {
            memcpy(out, in, n_in * 4);
        }
        return n_in;
    }
    const unsigned N = 1u << L;
    const unsigned target = 16 - L; /* 15, 14, 13 ... Decimators.cpp:43-44, 132-133, 222-223 */
    const unsigned ss = *sampleSize;
    const unsigned trunk = ss < target ? 0 : ss - target;
    const unsigned norm = ss < target ? target - ss : 0;
    const size_t n_resize = n_in >> L; /* out.resize(len/N) */
    *sampleSize = ss + L - trunk;      /* sampleSize += (L - trunk_shift) */
    if (n_resize == 0) return 0;       /* the reference's unsigned `len - (N-1)` would wrap: never called so */
    memset(out, 0, n_resize * 4);      /* elements a fresh vector would hold if never written */

    if (fcpos != ORC_FC_CEN && L <= 2) {
        /* static, filter-less variants: Decimators.cpp:38-91 (2), :127-170 (4) */
        size_t o = 0;
        for (size_t pos = 0; pos + 3 < n_in; pos += 4) {
            const int16_t *s = in + 2 * pos;
            int32_t I0 = s[0], Q0 = s[1], I1 = s[2], Q1 = s[3], I2 = s[4], Q2 = s[5], I3 = s[6], Q3 = s[7];
            if (L == 1) {
                int32_t xa, ya, xb, yb;
                if (fcpos == ORC_FC_INF) {
                    xa = I0 - Q1; ya = Q0 + I1; xb = Q3 - I2; yb = -Q2 - I3;
                } else {
                    xa = Q0 - I1; ya = -I0 - Q1; xb = I3 - Q2; yb = I2 + Q3;
                }
                out[2 * o] = final_shift(xa, norm, trunk); out[2 * o + 1] = final_shift(ya, norm, trunk); ++o;
                out[2 * o] = final_shift(xb, norm, trunk); out[2 * o + 1] = final_shift(yb, norm, trunk); ++o;
            } else {
                int32_t x, y;
                if (fcpos == ORC_FC_INF) {
                    x = I0 - Q1 + Q3 - I2; y = Q0 - Q2 + I1 - I3;
                } else {
                    x = Q0 - I1 - Q2 + I3; y = -I0 - Q1 + I2 + Q3;
                }
                out[2 * o] = final_shift(x, norm, trunk); out[2 * o + 1] = final_shift(y, norm, trunk); ++o;
            }
        }
        return n_resize;
    }

    /* filtered variants: every loop consumes floor(len/N)*N samples
     * (`pos < len - (N-1)`), the tail never enters the filter history. */
    size_t n_used = n_resize * N;
    size_t n0;           /* samples entering the first half-band stage */
    unsigned nstages;    /* half-band stages in the chain */
    int32_t *a[2], *b[2];
    if (fcpos == ORC_FC_CEN) {
        /* decimateN_cen: Decimators.cpp:94-120, 173-213, 270-334, 403-516, 595-805, 902-1305 */
        n0 = n_used; nstages = L;
    } else {
        /* decimateN_inf/_sup, N >= 8: fs/4 rotate + sum of 4 first (no filter), then L-2 stages
         * m_decimator2, m_decimator4, ...: Decimators.cpp:216-267, 337-400, 519-592, 808-899 */
        n0 = n_used / 4; nstages = L - 2;
    }
    for (int c = 0; c < 2; ++c) {
        a[c] = (int32_t *)malloc((n0 ? n0 : 1) * sizeof(int32_t));
        b[c] = (int32_t *)malloc((n0 ? n0 : 1) * sizeof(int32_t));
    }
    if (fcpos == ORC_FC_CEN) {
        for (size_t i = 0; i < n0; ++i) { a[0][i] = in[2 * i]; a[1][i] = in[2 * i + 1]; }
    } else {
        for (size_t g = 0; g < n0; ++g) {
            const int16_t *s = in + 8 * g;
            int32_t I0 = s[0], Q0 = s[1], I1 = s[2], Q1 = s[3], I2 = s[4], Q2 = s[5], I3 = s[6], Q3 = s[7];
            if (fcpos == ORC_FC_INF) {
                a[0][g] = I0 - Q1 + Q3 - I2; a[1][g] = Q0 - Q2 + I1 - I3; /* :351-352 */
            } else {
                a[0][g] = Q0 - I1 - Q2 + I3; a[1][g] = -I0 - Q1 + I2 + Q3; /* :384-385 */
            }
        }
    }
    size_t n = n0;
    for (unsigned s = 0; s < nstages; ++s) {
        n /= 2;
        dec_stage_run(&d->st[s], d->bias, n, a, b);
        int32_t *t;
        t = a[0]; a[0] = b[0]; b[0] = t;
        t = a[1]; a[1] = b[1]; b[1] = t;
    }
    for (size_t i = 0; i < n; ++i) {
        out[2 * i] = final_shift(a[0][i], norm, trunk);
        out[2 * i + 1] = final_shift(a[1][i], norm, trunk);
    }
    for (int c = 0; c < 2; ++c) { free(a[c]); free(b[c]); }
    return n_resize;
}

/* ---------------------------------------------------- interpolator stage -- */
/* IntHalfbandFilter{EO1,DB}<O>::myInterpolate (EO1.h:44-65 + :149-168;
 * DB.h:51-72 + :109-128, identical arithmetic).  Ring of O/2 inputs:
 *   v[2m]   = u[m - O/4]
 *   v[2m+1] = (sum_{i<O/4} c[i]*(u[m-(O/2-1)+i] + u[m-i])) >> 13                 */
typedef struct {
    int32_t hist[2][32]; /* last O/2 inputs (O/2 <= 32) */
    int order;
} int_stage;

static void int_stage_run(int_stage *st, size_t n_in, int32_t *const u[2], int32_t *const v[2])
{
    if (n_in == 0) return;
    const int O = st->order, S = O / 2, K = O / 4;
    const int32_t *c = O == 64 ? C64 : (O == 32 ? C32 : C16);
    int32_t *ext = (int32_t *)malloc((S + n_in) * sizeof(int32_t));
    for (int comp = 0; comp < 2; ++comp) {
        memcpy(ext, st->hist[comp], S * sizeof(int32_t));
        memcpy(ext + S, u[comp], n_in * sizeof(int32_t));
        for (size_t m = 0; m < n_in; ++m) {
            const int32_t *p = ext + S + m; /* p[0] = u[m] */
            uint32_t acc = 0;
            for (int i = 0; i < K; ++i)
                acc += ((uint32_t)p[-(S - 1) + i] + (uint32_t)p[-i]) * (uint32_t)c[i];
            v[comp][2 * m] = p[-K];
            v[comp][2 * m + 1] = asr(acc, HB_SHIFT_M1);
        }
        memcpy(st->hist[comp], ext + n_in, S * sizeof(int32_t));
    }
    free(ext);
}

struct orc_interpolators {
    int_stage st[6]; /* m_interpolator2 (64), 4 (32), 8..64 (16): Interpolators.h:47-52 */
};

void orc_interpolators_reset(orc_interpolators *p)
{
    static const int orders[6] = {64, 32, 16, 16, 16, 16};
    memset(p, 0, sizeof(*p));
    for (int i = 0; i < 6; ++i) p->st[i].order = orders[i];
}
orc_interpolators *orc_interpolators_new(void)
{
    orc_interpolators *p = (orc_interpolators *)malloc(sizeof(*p));
    if (p) orc_interpolators_reset(p);
    return p;
}
void orc_interpolators_free(orc_interpolators *p) { free(p); }

size_t orc_interpolate(orc_interpolators *p, int log2interp, const int16_t *in, size_t n_in,
                       int16_t *out)
{
    const unsigned L = (unsigned)log2interp;
    if (L == 0) { /* Upsampler.cpp:54-57 */
        if (out != in) memcpy(out, in, n_in * 4);
        return n_in;
    }
    /* interpolateN_cen (Interpolators.cpp:23, 47, 80, 130, 213, 363): every input runs
     * through stage 1, its two outputs through stage 2, ... int32 in between,
     * int16 truncation at the very end, no shifts. */
    size_t n_out = n_in << L;
    int32_t *a[2], *b[2];
    for (int c = 0; c < 2; ++c) {
        a[c] = (int32_t *)malloc((n_out ? n_out : 1) * sizeof(int32_t));
        b[c] = (int32_t *)malloc((n_out ? n_out : 1) * sizeof(int32_t));
    }
    for (size_t i = 0; i < n_in; ++i) { a[0][i] = in[2 * i]; a[1][i] = in[2 * i + 1]; }
    /* Reference quirk, mirrored: interpolate64_cen (Interpolators.cpp:363-606) only runs the
     * five stages m_interpolator2..32 (intbuf[0..63]) and emits intbuf[64..127], which it
     * zeroes once (:370) and never writes: per input sample 32 interpolated outputs followed
     * by 32 zero samples.  m_interpolator64 is never used. */
    const unsigned nstages = L == 6 ? 5 : L;
    size_t n = n_in;
    for (unsigned s = 0; s < nstages; ++s) {
        int_stage_run(&p->st[s], n, a, b);
        n *= 2;
        int32_t *t;
        t = a[0]; a[0] = b[0]; b[0] = t;
        t = a[1]; a[1] = b[1]; b[1] = t;
    }
    if (L == 6) {
        memset(out, 0, n_out * 4);
        for (size_t m = 0; m < n_in; ++m)
            for (size_t j = 0; j < 32; ++j) {
                out[2 * (64 * m + j)] = (int16_t)a[0][32 * m + j];
                out[2 * (64 * m + j) + 1] = (int16_t)a[1][32 * m + j];
            }
    } else {
        for (size_t i = 0; i < n_out; ++i) {
            out[2 * i] = (int16_t)a[0][i];
            out[2 * i + 1] = (int16_t)a[1][i];
        }
    }
    for (int c = 0; c < 2; ++c) { free(a[c]); free(b[c]); }
    return n_out;
}

/* ------------------------------------------------------------------ CRC -- */
uint32_t orc_crc32(const void *data, size_t n)
{
    /* boost::crc_32_type (UDPSinkFEC.cpp:106-109): reflected 0x04C11DB7, init/xorout ~0 */
    static uint32_t table[256];
    static int init = 0;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = 1;
    }
    const uint8_t *p = (const uint8_t *)data;
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

/* -------------------------------------------------------------- framing -- */
void orc_framer_init(orc_framer *f)
{
    memset(f, 0, sizeof(*f)); /* ctor: UDPSinkFEC.cpp:29-44 */
}

size_t orc_framer_write(orc_framer *f, const int16_t *iq, size_t n, uint8_t *frames_out)
{
    size_t frames = 0, pos = 0;
    while (pos < n) {
        size_t remaining = n - pos;
        if (f->tx_block_index == 0) { /* UDPSinkFEC.cpp:87-132 */
            orc_meta m;
            m.center_frequency_khz = f->center_frequency_khz;
            m.sample_rate = f->sample_rate;
            m.sample_bytes = f->sample_bytes;
            m.sample_bits = f->sample_bits;
            m.nb_original_blocks = ORC_NB_ORIGINAL;
            m.nb_fec_blocks = f->nb_fec_blocks;
            m.tv_sec = f->tv_sec;
            m.tv_usec = f->tv_usec;
            if (f->stamp_from_samples && f->sample_rate) {
                uint64_t dus = (uint64_t)pos * 1000000ull / f->sample_rate;
                m.tv_usec += (uint32_t)(dus % 1000000ull);
                m.tv_sec += (uint32_t)(dus / 1000000ull);
                if (m.tv_usec >= 1000000u) { m.tv_usec -= 1000000u; m.tv_sec += 1; }
            }
            m.crc32 = orc_crc32(&m, 20);
            memset(f->cur, 0, ORC_UDPSIZE);
            f->cur[0] = (uint8_t)(f->frame_count & 0xFF);
            f->cur[1] = (uint8_t)(f->frame_count >> 8);
            f->cur[2] = 0;
            memcpy(f->cur + 4, &m, sizeof(m));
            memcpy(f->slot, f->cur, ORC_UDPSIZE);
            f->tx_block_index = 1;
        }
        if ((size_t)f->sample_index + remaining < ORC_SAMPLES_PER_BLOCK) { /* :134-141 */
            memcpy(f->cur + 4 + 4 * f->sample_index, iq + 2 * pos, remaining * 4);
            f->sample_index += (int)remaining;
            pos = n;
        } else { /* :142-190 */
            size_t take = ORC_SAMPLES_PER_BLOCK - f->sample_index;
            memcpy(f->cur + 4 + 4 * f->sample_index, iq + 2 * pos, take * 4);
            pos += take;
            f->sample_index = 0;
            f->cur[0] = (uint8_t)(f->frame_count & 0xFF);
            f->cur[1] = (uint8_t)(f->frame_count >> 8);
            f->cur[2] = (uint8_t)f->tx_block_index;
            memcpy(f->slot + (size_t)f->tx_block_index * ORC_UDPSIZE, f->cur, ORC_UDPSIZE);
            if (f->tx_block_index == ORC_NB_ORIGINAL - 1) {
                memcpy(frames_out + frames * sizeof(f->slot), f->slot, sizeof(f->slot));
                ++frames;
                f->tx_block_index = 0;
                f->frame_count++;
            } else {
                f->tx_block_index++;
            }
        }
    }
    return frames;
}

/* --------------------------------------------------------------- GF(256) -- */
/* Upstream gf256.cpp (catid/cm256 = f4exb/cm256cc): GF256_GEN_POLY[3] = 0xa6
 * -> polynomial (0xa6 << 1) | 1 = 0x14D, generator 2. */
#define GF_POLY 0x14D
static uint8_t GF_EXP[512 * 2 + 1];
static uint16_t GF_LOG[256];
static uint8_t GF_MUL[256][256];
static int gf_ready = 0;

static void gf_init(void)
{
    if (gf_ready) return;
    GF_LOG[0] = 512;
    GF_EXP[0] = 1;
    for (unsigned j = 1; j < 255; ++j) {
        unsigned next = (unsigned)GF_EXP[j - 1] * 2;
        if (next >= 256) next ^= GF_POLY;
        GF_EXP[j] = (uint8_t)next;
        GF_LOG[GF_EXP[j]] = (uint16_t)j;
    }
    GF_EXP[255] = GF_EXP[0];
    GF_LOG[GF_EXP[255]] = 255;
    for (unsigned j = 256; j < 2 * 255; ++j) GF_EXP[j] = GF_EXP[j % 255];
    GF_EXP[2 * 255] = 1;
    for (unsigned j = 2 * 255 + 1; j < sizeof(GF_EXP); ++j) GF_EXP[j] = 0;
    for (unsigned a = 0; a < 256; ++a)
        for (unsigned b = 0; b < 256; ++b) GF_MUL[a][b] = (a && b) ? GF_EXP[GF_LOG[a] + GF_LOG[b]] : 0;
    gf_ready = 1;
}

uint8_t orc_gf_mul(uint8_t a, uint8_t b) { gf_init(); return GF_MUL[a][b]; }
uint8_t orc_gf_div(uint8_t a, uint8_t b)
{
    gf_init(); /* upstream gf256_div: EXP[LOG[a] + 255 - LOG[b]], LOG[0] = 512 -> 0 */
    if (a == 0) return 0;
    return GF_EXP[GF_LOG[a] + 255 - GF_LOG[b]];
}
uint8_t orc_gf_exp(int i) { gf_init(); return GF_EXP[i]; }
int orc_gf_log(uint8_t a) { gf_init(); return GF_LOG[a]; }

uint8_t orc_cm256_matrix_element(uint8_t x_i, uint8_t x_0, uint8_t y_j)
{
    /* upstream GetMatrixElement: div(add(y_j, x_0), add(x_i, y_j)) */
    return orc_gf_div((uint8_t)(y_j ^ x_0), (uint8_t)(x_i ^ y_j));
}

void orc_gf_muladd_mem(uint8_t *dst, uint8_t c, const uint8_t *src, size_t n)
{
    gf_init();
    if (c == 0) return;
    size_t i = 0;
#if defined(__SSSE3__)
    /* the upstream technique: two 16-entry nibble tables applied with pshufb */
    uint8_t lo[16], hi[16];
    for (int k = 0; k < 16; ++k) { lo[k] = GF_MUL[c][k]; hi[k] = GF_MUL[c][k << 4]; }
    const __m128i tlo = _mm_loadu_si128((const __m128i *)lo);
    const __m128i thi = _mm_loadu_si128((const __m128i *)hi);
    const __m128i mask = _mm_set1_epi8(0x0f);
    for (; i + 16 <= n; i += 16) {
        __m128i x = _mm_loadu_si128((const __m128i *)(src + i));
        __m128i l = _mm_and_si128(x, mask);
        __m128i h = _mm_and_si128(_mm_srli_epi64(x, 4), mask);
        __m128i p = _mm_xor_si128(_mm_shuffle_epi8(tlo, l), _mm_shuffle_epi8(thi, h));
        __m128i d = _mm_loadu_si128((const __m128i *)(dst + i));
        _mm_storeu_si128((__m128i *)(dst + i), _mm_xor_si128(d, p));
    }
#endif
    const uint8_t *row = GF_MUL[c];
    for (; i < n; ++i) dst[i] ^= row[src[i]];
}

/* ---------------------------------------------------------------- CM256 -- */
int orc_cm256_encode(orc_cm256_params p, const orc_cm256_block *originals, void *recoveryBlocks)
{
    if (p.OriginalCount <= 0 || p.RecoveryCount <= 0 || p.BlockBytes <= 0) return -1;
    if (p.OriginalCount + p.RecoveryCount > 256) return -2;
    if (!originals || !recoveryBlocks) return -3;
    gf_init();
    uint8_t *rec = (uint8_t *)recoveryBlocks;
    const uint8_t x_0 = (uint8_t)p.OriginalCount;
    for (int r = 0; r < p.RecoveryCount; ++r, rec += p.BlockBytes) {
        if (p.OriginalCount == 1) { memcpy(rec, originals[0].Block, (size_t)p.BlockBytes); continue; }
        memset(rec, 0, (size_t)p.BlockBytes);
        const uint8_t x_i = (uint8_t)(p.OriginalCount + r);
        for (int j = 0; j < p.OriginalCount; ++j) {
            /* row 0 (x_i == x_0) is all ones = plain XOR parity */
            uint8_t m = orc_cm256_matrix_element(x_i, x_0, (uint8_t)j);
            orc_gf_muladd_mem(rec, m, (const uint8_t *)originals[j].Block, (size_t)p.BlockBytes);
        }
    }
    return 0;
}

int orc_cm256_decode(orc_cm256_params p, orc_cm256_block *blocks)
{
    if (p.OriginalCount <= 0 || p.RecoveryCount <= 0 || p.BlockBytes <= 0) return -1;
    if (p.OriginalCount + p.RecoveryCount > 256) return -2;
    if (!blocks) return -3;
    if (p.OriginalCount == 1) { blocks[0].Index = 0; return 0; }
    gf_init();
    const int k = p.OriginalCount;
    const size_t bb = (size_t)p.BlockBytes;
    /* CM256Decoder::Initialize: split the first k descriptors, list the erasures ascending */
    orc_cm256_block *orig[256], *rec[256];
    uint8_t present[256];
    uint8_t erased[256];
    int n_orig = 0, n_rec = 0;
    memset(present, 0, sizeof(present));
    for (int i = 0; i < k; ++i) {
        int row = blocks[i].Index;
        if (row < k) {
            if (present[row]) return -5; /* duplicate original index */
            present[row] = 1;
            orig[n_orig++] = &blocks[i];
        } else {
            rec[n_rec++] = &blocks[i];
        }
    }
    if (n_rec <= 0) return 0; /* nothing erased */
    for (int i = 0, cnt = 0; i < 256 && cnt < n_rec; ++i)
        if (!present[i]) erased[cnt++] = (uint8_t)i;

    if (p.RecoveryCount == 1) {
        /* upstream DecodeM1: XOR every received original into recovery block 0, whatever
         * its row -- correct only for row k; mirrored on purpose. */
        uint8_t *o = (uint8_t *)rec[0]->Block;
        for (int i = 0; i < n_orig; ++i) {
            const uint8_t *s = (const uint8_t *)orig[i]->Block;
            for (size_t b = 0; b < bb; ++b) o[b] ^= s[b];
        }
        rec[0]->Index = erased[0];
        return 0;
    }

    const int N = n_rec;
    const uint8_t x_0 = (uint8_t)k;
    /* eliminate the received originals from the recovery rows (upstream Decode, first loop) */
    for (int oi = 0; oi < n_orig; ++oi) {
        const uint8_t *src = (const uint8_t *)orig[oi]->Block;
        for (int ri = 0; ri < N; ++ri) {
            uint8_t m = orc_cm256_matrix_element(rec[ri]->Index, x_0, orig[oi]->Index);
            orc_gf_muladd_mem((uint8_t *)rec[ri]->Block, m, src, bb);
        }
    }
    /* solve the N x N Cauchy system  A * X = B,  A[ri][e] = elem(x_ri, x_0, erased[e]).
     * Upstream uses an O(N^2) LDU factorisation; the code is MDS so the solution is unique
     * and plain Gauss-Jordan gives the same bytes. */
    uint8_t *A = (uint8_t *)malloc((size_t)N * N);
    for (int ri = 0; ri < N; ++ri)
        for (int e = 0; e < N; ++e) A[ri * N + e] = orc_cm256_matrix_element(rec[ri]->Index, x_0, erased[e]);
    uint8_t **B = (uint8_t **)malloc((size_t)N * sizeof(uint8_t *));
    for (int ri = 0; ri < N; ++ri) B[ri] = (uint8_t *)rec[ri]->Block;
    uint8_t *tmp = (uint8_t *)malloc(bb);
    int rc = 0;
    for (int col = 0; col < N; ++col) {
        int piv = -1;
        for (int r = col; r < N; ++r)
            if (A[r * N + col]) { piv = r; break; }
        if (piv < 0) { rc = -6; break; }
        if (piv != col) {
            for (int c = 0; c < N; ++c) { uint8_t t = A[piv * N + c]; A[piv * N + c] = A[col * N + c]; A[col * N + c] = t; }
            /* swap block CONTENTS so that row `col` keeps living in rec[col]'s buffer */
            memcpy(tmp, B[piv], bb); memcpy(B[piv], B[col], bb); memcpy(B[col], tmp, bb);
        }
        uint8_t inv = orc_gf_div(1, A[col * N + col]);
        for (int c = 0; c < N; ++c) A[col * N + c] = GF_MUL[inv][A[col * N + c]];
        for (size_t b = 0; b < bb; ++b) B[col][b] = GF_MUL[inv][B[col][b]];
        for (int r = 0; r < N; ++r) {
            if (r == col) continue;
            uint8_t f = A[r * N + col];
            if (!f) continue;
            for (int c = 0; c < N; ++c) A[r * N + c] ^= GF_MUL[f][A[col * N + c]];
            orc_gf_muladd_mem(B[r], f, B[col], bb);
        }
    }
    free(tmp); free(B); free(A);
    if (rc) return rc;
    for (int i = 0; i < N; ++i) rec[i]->Index = erased[i]; /* recovery i now holds erased[i] */
    return 0;
}

int orc_frame_encode(const uint8_t *frame, int nb_fec, uint8_t *recovery_superblocks)
{
    /* UDPSinkFEC.cpp:228-256: descriptors over the 508-byte protected blocks of the 128 super
     * blocks, encode, then super blocks 128.. get header {frameIndex, i} + recovery bytes. */
    if (nb_fec <= 0) return 0;
    orc_cm256_params p = {ORC_NB_ORIGINAL, nb_fec, ORC_BLOCK_BYTES};
    orc_cm256_block desc[ORC_NB_ORIGINAL];
    for (int i = 0; i < ORC_NB_ORIGINAL; ++i) {
        desc[i].Block = (void *)(frame + (size_t)i * ORC_UDPSIZE + 4);
        desc[i].Index = (uint8_t)i;
    }
    uint8_t *fec = (uint8_t *)malloc((size_t)nb_fec * ORC_BLOCK_BYTES);
    int rc = orc_cm256_encode(p, desc, fec);
    if (rc == 0) {
        for (int r = 0; r < nb_fec; ++r) {
            uint8_t *sb = recovery_superblocks + (size_t)r * ORC_UDPSIZE;
            sb[0] = frame[0]; sb[1] = frame[1]; /* frameIndex */
            sb[2] = (uint8_t)(ORC_NB_ORIGINAL + r);
            sb[3] = 0;
            memcpy(sb + 4, fec + (size_t)r * ORC_BLOCK_BYTES, ORC_BLOCK_BYTES);
        }
    }
    free(fec);
    return rc;
}

/* ----------------------------------------------------- SDRdaemonFECBuffer -- */
static void fecbuffer_init_slot(orc_fecbuffer *b)
{
    /* initDecodeSlot, SDRdaemonFECBuffer.cpp:95-110 */
    b->cur_nb_blocks = b->block_count;
    b->cur_nb_recovery = b->recovery_count;
    if (b->cur_nb_blocks < b->min_nb_blocks) b->min_nb_blocks = b->cur_nb_blocks;
    if (b->cur_nb_recovery > b->max_nb_recovery) b->max_nb_recovery = b->cur_nb_recovery;
    b->block_count = 0;
    b->recovery_count = 0;
    b->decoded = 0;
    b->meta_retrieved = 0;
    memset(b->frame, 0, sizeof(b->frame));
}

void orc_fecbuffer_init(orc_fecbuffer *b)
{
    memset(b, 0, sizeof(*b));
    b->frame_head = -1;     /* SDRdaemonFECBuffer.cpp:36 */
    b->min_nb_blocks = 256; /* :39 */
}

int orc_fecbuffer_write_and_read(orc_fecbuffer *b, const uint8_t *sb, uint8_t *data, size_t *data_length)
{
    int available = 0;
    *data_length = 0;
    int frame_index = sb[0] | (sb[1] << 8);
    if (b->frame_head != frame_index) { /* :133-139 */
        *data_length = (ORC_NB_ORIGINAL - 1) * ORC_BLOCK_BYTES; /* getSlotData :72-75 */
        memcpy(data, b->frame[1], *data_length);
        available = 1;
        fecbuffer_init_slot(b);
        b->frame_head = frame_index;
    }
    if (b->block_count < ORC_NB_ORIGINAL) { /* :143-166 */
        int bc = b->block_count, rc = b->recovery_count;
        int block_index = sb[2];
        b->desc[bc].Index = (uint8_t)block_index;
        if (block_index == 0) b->meta_retrieved = 1;
        if (block_index < ORC_NB_ORIGINAL) {
            memcpy(b->frame[block_index], sb + 4, ORC_BLOCK_BYTES);
            b->desc[bc].Block = b->frame[block_index];
        } else {
            memcpy(b->recovery[rc], sb + 4, ORC_BLOCK_BYTES);
            b->desc[bc].Block = b->recovery[rc];
            b->recovery_count++;
        }
    }
    b->block_count++;
    if (b->block_count == ORC_NB_ORIGINAL) { /* :170-247 */
        b->decoded = 1;
        if (b->recovery_count > 0) {
            orc_cm256_params p = {ORC_NB_ORIGINAL, b->recovery_count, ORC_BLOCK_BYTES};
            if (orc_cm256_decode(p, b->desc) == 0) {
                for (int ir = 0; ir < b->recovery_count; ++ir) { /* :208-213 */
                    int ri = ORC_NB_ORIGINAL - b->recovery_count + ir;
                    int block_index = b->desc[ri].Index;
                    memcpy(b->frame[block_index], b->desc[ri].Block, ORC_BLOCK_BYTES);
                }
            }
        }
    }
    return available;
}

/* ------------------------------------------------------------------------------------------------
 * TestSource bank.  The reference's generator (TestSource.cpp:395-422: float phasor, -ffast-math, a
 * wrap bug at :411-415) is not reproducible, so the product defines an integer-exact NCO of its own; this is the
 * independent statement of that definition the product is checked against:
 *   phase(n) = phase0 + n * inc mod 2^32;  I = trunc(A * C[phase >> 20] / 2^30), Q = the same a quarter turn back;
 *   C[i] = 2^30 cos(2 pi i / 4096) from a 31-step CORDIC in integers; inc = round(2^32 df / srate);
 *   A(dB) in Q15 from dB steps of 10^(-1/20) in Q30.  Written table-free (the CORDIC runs per sample). */
static const int orc_nco_atan[31] = {536870912, 316933406, 167458907, 85004756, 42667331, 21354465, 10679838, 5340245, 2670163, 1335087,
                                     667544, 333772, 166886, 83443, 41722, 20861, 10430, 5215, 2608, 1304, 652, 326, 163, 81, 41, 20, 10, 5,
                                     3, 1, 1};

int orc_nco_cos_q30(unsigned idx12)
{
    int64_t z = (int64_t)(idx12 & 4095u) << 20;
    int neg = 0;
    if (z >= 0x80000000LL) z -= 0x100000000LL;
    if (z > 0x40000000LL) { z -= 0x80000000LL; neg = 1; }
    else if (z < -0x40000000LL) { z += 0x80000000LL; neg = 1; }
    int64_t x = 652032874, y = 0;
    for (int k = 0; k < 31; ++k) {
        const int64_t xs = x >> k, ys = y >> k;
        if (z >= 0) { x -= ys; y += xs; z -= orc_nco_atan[k]; }
        else { x += ys; y -= xs; z += orc_nco_atan[k]; }
    }
    return (int)(neg ? -x : x);
}

unsigned orc_nco_phase_inc(int64_t df, int64_t srate)
{
    const int64_t num = df * 4294967296LL;
    const int64_t q = num >= 0 ? (num + srate / 2) / srate : -((-num + srate / 2) / srate);
    return (unsigned)(uint64_t)q;
}

int orc_nco_amp_q15(int db)
{
    int64_t a = 1LL << 30;
    for (int i = 0; i < db && a > 0; ++i) a = (a * 956973408LL + (1LL << 29)) >> 30;
    return (int)((a + (1LL << 14)) >> 15);
}

static int16_t orc_nco_scale(int amp, int c)
{
    const int64_t v = (int64_t)amp * c;
    int64_t r = v >= 0 ? v >> 30 : -((-v) >> 30);
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return (int16_t)r;
}

/* n IQ samples from phase0; returns the phase of the sample after the last one */
unsigned orc_testsource_generate(unsigned phase0, unsigned inc, int amp_q15, size_t n, int16_t *iq_out)
{
    unsigned ph = phase0;
    for (size_t k = 0; k < n; ++k, ph += inc) {
        const unsigned idx = ph >> 20;
        iq_out[2 * k] = orc_nco_scale(amp_q15, orc_nco_cos_q30(idx));
        iq_out[2 * k + 1] = orc_nco_scale(amp_q15, orc_nco_cos_q30((idx - 1024u) & 4095u));
    }
    return ph;
}
