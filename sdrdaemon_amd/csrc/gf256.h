// gf256.h -- host-side GF(2^8) arithmetic and CM256 matrix planning (small, per-frame
// O(N^2..N^3) byte work; the per-byte block arithmetic runs on the GPU, gf_kernels.hip).
//
// Field and matrix follow the published cm256 algorithm (catid/cm256 = f4exb/cm256cc,
// the library the reference links: cm256cc/CMakeLists.txt:12-34): polynomial 0x14D
// (GF256_GEN_POLY[3] = 0xa6 -> (0xa6 << 1) | 1), generator 2, Cauchy element
// GetMatrixElement(x_i, x_0, y_j) = (y_j ^ x_0) / (x_i ^ y_j) with x_0 = OriginalCount,
// x_i = OriginalCount + r, y_j = j.  The library itself is absent from the reference tree:
// wire compatibility rests on these two named constants ("parity unpinned").
#pragma once
#include <stdint.h>

namespace sdrhip {

constexpr unsigned GF_POLYNOMIAL = 0x14D;

struct GF256 {
    uint8_t exp[1024];
    uint16_t log[256];
    uint8_t mul[256][256];
    uint8_t inv[256];
    GF256();
    uint8_t div(uint8_t a, uint8_t b) const { return a ? exp[log[a] + 255 - log[b]] : 0; }
    uint8_t matrix_element(uint8_t x_i, uint8_t x_0, uint8_t y_j) const { return div((uint8_t)(y_j ^ x_0), (uint8_t)(x_i ^ y_j)); }
};
const GF256 &gf();

// 256 x 32 byte table used by the kernels: for multiplier m
//   [0..7]   m * i          (i = 0..7, low three bits of the data byte)
//   [8..15]  m * (i << 3)   (middle three bits)
//   [16..19] m * (i << 6)   (top two bits), [20..31] zero
int gf_build_tables(uint8_t *tab);

// rows x k encode matrix of cm256_encode (row r <-> recovery block index k + r)
void cm256_encode_matrix(int k, int rows, uint8_t *m /* rows * k */);

// Karatsuba leaf constants of the structured k = 128 encoder (gf_kernels.hip): for b = 0..7 the
// 81 leaves, in depth-first (lo, hi, lo ^ hi) order, of the 16-point kernel G_b[u] = 1 / (128 ^ (16 b + u)),
// each as its 32-byte multiplier table (layout of gf_build_tables).
void cm256_karatsuba_leaf_tables(uint8_t *out /* 8 * 81 * 32 */);

// Constants of the additive-FFT form of the same encoder (gf_encode128_fft.h), each as its 32-byte multiplier table (layout of
// gf_build_tables), CM256_FFT_TABLES entries:
//   [63 h + (64 - (64 >> k)) + j]  normalised subspace polynomial s^_k(64 h ^ (j << (k + 1))): butterfly constant of block j of stage
//                                  k = 0..5 of the size-64 inverse transform on the coset 64 h + V6 (h = 0, 1)
//   [126], [127]                   s^_5(128), s^_6(128): the fold of the 128 novel-basis coefficients onto the coset 128 + V5
//   [128 + (32 - (32 >> k)) + j]   s^_k(128 ^ (j << (k + 1))): stage k = 0..4 of the size-32 transform on 128 + V5
//   [160 + r]                      r * c / q (c = product of the nonzero elements of V7, q = s_7(128)): the scale of recovery row r < 32
constexpr int CM256_FFT_TABLES = 192;
void cm256_fft_tables(uint8_t *out /* CM256_FFT_TABLES * 32 */);

// Decode plan for one frame, restating CM256Decoder::Initialize + Decode/DecodeM1:
// `indices` are the Index fields of the k descriptors in array order.  On success
//   n_rec         number of recovery descriptors (= erasures repaired)
//   rec_pos[i]    array position of the i-th recovery descriptor
//   erased[i]     original index recovered INTO that descriptor (ascending)
//   coef          n_rec x k matrix over the k received blocks in array order such that
//                 recovered_i = XOR_p coef[i][p] * block[p]
// Returns 0, or -5 on a duplicate original index / singular system (upstream: Initialize false).
int cm256_decode_plan(int k, int recovery_count_param, const uint8_t *indices, int *n_rec, uint8_t *rec_pos,
                      uint8_t *erased, uint8_t *coef /* up to k * k */);

} // namespace sdrhip
