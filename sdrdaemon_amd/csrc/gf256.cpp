// gf256.cpp -- see gf256.h
#include "gf256.h"

#include <cstring>
#include <vector>

namespace sdrhip {

GF256::GF256()
{
    // upstream gf256_explog_init: exp[0] = 1, exp[j] = xtime(exp[j-1]), log[0] = 512
    memset(exp, 0, sizeof(exp));
    log[0] = 512;
    exp[0] = 1;
    for (unsigned j = 1; j < 255; ++j) {
        unsigned next = (unsigned)exp[j - 1] * 2;
        if (next >= 256) next ^= GF_POLYNOMIAL;
        exp[j] = (uint8_t)next;
        log[exp[j]] = (uint16_t)j;
    }
    exp[255] = exp[0];
    log[exp[255]] = 255;
    for (unsigned j = 256; j < 2 * 255; ++j) exp[j] = exp[j % 255];
    exp[2 * 255] = 1;
    for (unsigned a = 0; a < 256; ++a)
        for (unsigned b = 0; b < 256; ++b) mul[a][b] = (a && b) ? exp[log[a] + log[b]] : 0;
    inv[0] = 0;
    for (unsigned a = 1; a < 256; ++a) inv[a] = div(1, (uint8_t)a);
}

const GF256 &gf()
{
    static const GF256 g;
    return g;
}

int gf_build_tables(uint8_t *tab)
{
    const GF256 &g = gf();
    memset(tab, 0, 256 * 32);
    for (int m = 0; m < 256; ++m) {
        uint8_t *t = tab + m * 32;
        for (int i = 0; i < 8; ++i) t[i] = g.mul[m][i];
        for (int i = 0; i < 8; ++i) t[8 + i] = g.mul[m][i << 3];
        for (int i = 0; i < 4; ++i) t[16 + i] = g.mul[m][i << 6];
    }
    return 0;
}

void cm256_encode_matrix(int k, int rows, uint8_t *m)
{
    const GF256 &g = gf();
    const uint8_t x_0 = (uint8_t)k;
    for (int r = 0; r < rows; ++r)
        for (int j = 0; j < k; ++j) m[r * k + j] = g.matrix_element((uint8_t)(k + r), x_0, (uint8_t)j);
}

static void kleaves(const uint8_t *g, int n, std::vector<uint8_t> &out)
{
    if (n == 1) { out.push_back(g[0]); return; }
    const int h = n / 2;
    kleaves(g, h, out);
    kleaves(g + h, h, out);
    uint8_t s[16];
    for (int i = 0; i < h; ++i) s[i] = (uint8_t)(g[i] ^ g[h + i]);
    kleaves(s, h, out);
}

void cm256_karatsuba_leaf_tables(uint8_t *out)
{
    const GF256 &f = gf();
    std::vector<uint8_t> all(256 * 32);
    gf_build_tables(all.data());
    for (int b = 0; b < 8; ++b) {
        uint8_t g[16];
        for (int u = 0; u < 16; ++u) g[u] = f.inv[128 ^ (16 * b + u)];
        std::vector<uint8_t> lv;
        kleaves(g, 16, lv);
        for (size_t i = 0; i < lv.size(); ++i) memcpy(out + ((size_t)b * 81 + i) * 32, &all[(size_t)lv[i] * 32], 32);
    }
}

// CM256's 128-original encode as an additive FFT (Lin, Chung, Han: "Novel polynomial basis and its application to Reed-Solomon
// erasure codes", 2014).  With x_0 = 128, x_r = 128 ^ r, y_j = j the recovery block is P ^ r * S(x_r), S(x) = sum_j d_j / (x ^ j)
// (gf_encode128_body.h).  V_k = {0 .. 2^k - 1} is a GF(2)-subspace; s_k = its subspace polynomial (s_0 = x, s_{k+1}(x) = s_k(x)
// (s_k(x) ^ s_k(2^k)); linearised, so constant on cosets of V_k), s^_k = s_k / s_k(2^k).  S = N / s_7 with N the polynomial of
// degree < 128 that takes the values c d_j on V7 (c = s_7'(0) = the product of the nonzero elements of V7), and s_7 = q on the
// whole coset 128 + V7.  In the basis X_i = prod_{bit k of i} s^_k a transform of size 2^m on a coset is m stages of butterflies
// (a ^= const * b, b ^= a) whose constant is s^_k of the block's coset representative.
void cm256_fft_tables(uint8_t *out)
{
    const GF256 &f = gf();
    std::vector<uint8_t> all(256 * 32);
    gf_build_tables(all.data());
    uint8_t s[8][256], shat[8][256];
    for (int x = 0; x < 256; ++x) s[0][x] = (uint8_t)x;
    for (int k = 0; k < 7; ++k)
        for (int x = 0; x < 256; ++x) s[k + 1][x] = f.mul[s[k][x]][s[k][x] ^ s[k][1 << k]];
    for (int k = 0; k < 8; ++k)
        for (int x = 0; x < 256; ++x) shat[k][x] = f.mul[s[k][x]][f.inv[s[k][1 << k]]];
    uint8_t cst[CM256_FFT_TABLES];
    memset(cst, 0, sizeof(cst));
    for (int h = 0; h < 2; ++h)
        for (int k = 0; k < 6; ++k)
            for (int j = 0; j < (32 >> k); ++j) cst[63 * h + (64 - (64 >> k)) + j] = shat[k][(64 * h) ^ (j << (k + 1))];
    cst[126] = shat[5][128];
    cst[127] = shat[6][128];
    for (int k = 0; k < 5; ++k)
        for (int j = 0; j < (16 >> k); ++j) cst[128 + (32 - (32 >> k)) + j] = shat[k][128 ^ (j << (k + 1))];
    uint8_t c = 1;
    for (int v = 1; v < 128; ++v) c = f.mul[c][v];
    const uint8_t q = s[7][128];
    for (int r = 0; r < 32; ++r) cst[160 + r] = f.mul[f.mul[r][c]][f.inv[q]];
    for (int i = 0; i < CM256_FFT_TABLES; ++i) memcpy(out + (size_t)i * 32, &all[(size_t)cst[i] * 32], 32);
}

int cm256_decode_plan(int k, int recovery_count_param, const uint8_t *indices, int *n_rec_out, uint8_t *rec_pos,
                      uint8_t *erased, uint8_t *coef)
{
    const GF256 &g = gf();
    uint8_t present[256];
    memset(present, 0, sizeof(present));
    int n_rec = 0;
    std::vector<int> orig_pos;
    orig_pos.reserve(k);
    for (int p = 0; p < k; ++p) {
        int row = indices[p];
        if (row < k) {
            if (present[row]) return -5;
            present[row] = 1;
            orig_pos.push_back(p);
        } else {
            rec_pos[n_rec++] = (uint8_t)p;
        }
    }
    *n_rec_out = n_rec;
    if (n_rec == 0) return 0;
    for (int i = 0, cnt = 0; i < 256 && cnt < n_rec; ++i)
        if (!present[i]) erased[cnt++] = (uint8_t)i;

    if (recovery_count_param == 1) {
        // upstream DecodeM1: XOR of every received original into Recovery[0], whatever its row
        *n_rec_out = 1;
        memset(coef, 0, (size_t)k);
        for (size_t t = 0; t < orig_pos.size(); ++t) coef[orig_pos[t]] = 1;
        coef[rec_pos[0]] = 1;
        return 0;
    }

    const int N = n_rec;
    const uint8_t x_0 = (uint8_t)k;
    // A[i][t] = element(x_i, x_0, erased[t]); invert by Gauss-Jordan on [A | I]
    std::vector<uint8_t> A((size_t)N * N), Inv((size_t)N * N, 0);
    for (int i = 0; i < N; ++i) {
        for (int t = 0; t < N; ++t) A[(size_t)i * N + t] = g.matrix_element(indices[rec_pos[i]], x_0, erased[t]);
        Inv[(size_t)i * N + i] = 1;
    }
    for (int col = 0; col < N; ++col) {
        int piv = -1;
        for (int r = col; r < N; ++r)
            if (A[(size_t)r * N + col]) { piv = r; break; }
        if (piv < 0) return -5;
        if (piv != col)
            for (int c = 0; c < N; ++c) {
                uint8_t t = A[(size_t)piv * N + c]; A[(size_t)piv * N + c] = A[(size_t)col * N + c]; A[(size_t)col * N + c] = t;
                t = Inv[(size_t)piv * N + c]; Inv[(size_t)piv * N + c] = Inv[(size_t)col * N + c]; Inv[(size_t)col * N + c] = t;
            }
        const uint8_t *mi = g.mul[g.inv[A[(size_t)col * N + col]]];
        for (int c = 0; c < N; ++c) { A[(size_t)col * N + c] = mi[A[(size_t)col * N + c]]; Inv[(size_t)col * N + c] = mi[Inv[(size_t)col * N + c]]; }
        for (int r = 0; r < N; ++r) {
            if (r == col) continue;
            uint8_t f = A[(size_t)r * N + col];
            if (!f) continue;
            const uint8_t *mf = g.mul[f];
            for (int c = 0; c < N; ++c) { A[(size_t)r * N + c] ^= mf[A[(size_t)col * N + c]]; Inv[(size_t)r * N + c] ^= mf[Inv[(size_t)col * N + c]]; }
        }
    }
    // X = Inv * B, B_i = rec_i ^ sum_{received originals p} element(x_i, idx[p]) * block[p]
    memset(coef, 0, (size_t)N * k);
    std::vector<uint8_t> E((size_t)N * orig_pos.size());
    for (int i = 0; i < N; ++i)
        for (size_t t = 0; t < orig_pos.size(); ++t)
            E[(size_t)i * orig_pos.size() + t] = g.matrix_element(indices[rec_pos[i]], x_0, indices[orig_pos[t]]);
    for (int t = 0; t < N; ++t) {
        uint8_t *row = coef + (size_t)t * k;
        for (int i = 0; i < N; ++i) {
            const uint8_t a = Inv[(size_t)t * N + i];
            row[rec_pos[i]] = a;
            if (!a) continue;
            const uint8_t *ma = g.mul[a];
            const uint8_t *e = &E[(size_t)i * orig_pos.size()];
            for (size_t q = 0; q < orig_pos.size(); ++q) row[orig_pos[q]] ^= ma[e[q]];
        }
    }
    return 0;
}

} // namespace sdrhip
