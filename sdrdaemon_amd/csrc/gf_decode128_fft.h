// gf_decode128_fft.h -- the syndrome decoder for 128 originals (gf_decode128_wg, gf_kernels.hip) with the additive-FFT encoder walk
// (gf_encode128_fft.h) in place of the Karatsuba walk (round 5).  Include inside namespace sdrhip { namespace { ... } } behind
// Dec128Plan / Dec128Args, gf_encode128_fft.h and mulc / make_sel.
//
// What the received originals contribute to the received recovery rows is what the ENCODER computes with the erased originals set
// to zero (SDRdaemonFECBuffer.cpp:197's cm256_decode seen from the other side): syndrome_i = recovery_i ^ that, erased originals =
// Minv (N x N, from the planner) x syndromes.  The encoder part is the size-128 inverse transform + fold + size-32 transform of
// gf_encode128_fft.h: 592 constant multiplications per 4-byte column instead of 1296, for recovery rows 0..31 (a frame that holds a
// higher row -- a sender with more than 32 FEC blocks -- takes the Karatsuba walk: gf_decode128_fft_unit).
// A WORKGROUP is one frame, its four waves = (column half ch, block half hf) like the encoder's; Minv x syndromes: the two waves
// of a column half share the N rows (rows t with t mod 4 in {hf, hf + 2}).
#pragma once

// LDS: FFT tables | all 256 multiplier tables (Minv's constants) | the frame's plan record | per column half [34][64] dwords of
// exchange (fft_rows16 + two parity rows); the syndromes ([32][64] per column half) take the exchange rows' place behind a barrier
constexpr int DEC128_FFT_LDS_BYTES = FFT_TAB_BYTES + 256 * 32 + DEC128_PLAN_BYTES + 2 * FFT_XCH_DWORDS * 4;
static_assert(DEC128_PLAN_BYTES % 16 == 0 && DEC128_MAXN == 32, "plan record layout");

template <int HF, bool FUSED, bool NOCOPY> __device__ __forceinline__ void gf_decode128_fft_wave(const Dec128Args &a, int fr, unsigned char *ldsraw, int ch);

// The received frames are read exactly once: their loads are non-temporal (buffer cache policy bit 1 = nt on gfx950).  With the default
// policy every line of the burst allocates in the L2 / Infinity Cache and evicts a dirty line of whatever ran before (in the Tx pipe the
// interpolator's 1 GB of output): tools/fec_burst_probe.hip -- the workgroups' 67 MB behind a 1-GiB store stream: 28.1 us, nt 13.7 us;
// with the copy stores 44.9 / 30.2 us (nt STORES: no gain, 33.4).
#ifndef DEC_LOAD_AUX
#define DEC_LOAD_AUX 2
#endif

// ---- the frame's plan made by the decoder's own workgroup (round 6: FUSED).  gf_decode_plan_kernel (gf_kernels.hip) used to run as a
// launch of its own in front of this kernel: 16 us per 1024 frames, serial, most of it one workgroup's latency chain.  When no frame
// can carry more than DEC128_MAXN recovery blocks (the caller's dec_max_rows promise <= 32: the sender's fecblk) everything a frame's
// record holds is derived HERE, in LDS, by the workgroup that decodes the frame: the classification (where every original lies, which
// recovery rows came, which originals they restore, N, cm256's DecodeM1 case, the repeated-block and dec_max_rows checks)
// and the closed-form inverse of the Cauchy block (four logarithm sums per row and column -- LDS atomics over the N x N pairs -- and one
// table look-up per element: gf_decode_plan_kernel's formulas), all of it in front of the data loads: they need the position map, and
// anything placed behind their issue waits for them (64 loads per lane in flight: a barrier or a spill reload there drains them first:
// measured, 87 against 77 us).  Same records, same semantics (SDRdaemonFECBuffer.cpp:143-213 through cm256_decode); a recovery block that
// comes twice is found by counting (the separate kernel finds it as a singular system: x_i = x_k), an incomplete frame's missing
// blocks are zeroed by the copy loop itself (Dec128Plan::pad = 1) instead of a fill pass in front.
struct Dec128Scratch {
    uint8_t exp[512];          // (exp, then log: the 1 KiB of gf_explog as it is)
    uint16_t log[256];
    int cnt[256];              // how often block index b arrived
    uint8_t x[128], y[128], rpos[128], idx[128];
    int lpx[DEC128_MAXN], lqx[DEC128_MAXN], lpy[DEC128_MAXN], lqy[DEC128_MAXN];
    unsigned long long mask[2][3];
};
static_assert(sizeof(Dec128Scratch) % 16 == 0, "scratch size");
static_assert(offsetof(Dec128Scratch, lqx) == offsetof(Dec128Scratch, lpx) + 4 * DEC128_MAXN && offsetof(Dec128Scratch, lpy) == offsetof(Dec128Scratch, lpx) + 8 * DEC128_MAXN &&
              offsetof(Dec128Scratch, lqy) == offsetof(Dec128Scratch, lpx) + 12 * DEC128_MAXN && offsetof(Dec128Scratch, log) == 512,
              "dec128_plan clears the four logarithm sums as one array and loads exp + log as one 1 KiB block");
constexpr int DEC128_FFT_FUSED_LDS_BYTES = DEC128_FFT_LDS_BYTES + (int)sizeof(Dec128Scratch);

__device__ __forceinline__ void dec128_plan(const Dec128Args &a, int fr, Dec128Plan *pl, Dec128Scratch *s, unsigned char *ldsraw)
{
    constexpr int K = 128;
    const int tid = threadIdx.x;
    // the frame's block indices first (the one load the whole chain waits for: 128 header bytes, 512 bytes apart), the tables behind them
    int b = 0;
#ifndef DEC_HDR_NT
#define DEC_HDR_NT 1 // (the header bytes non-temporal too, although their lines are read again by the data loads: this barrier 8.1 -> 4.4 us, decode -2.8 us)
#endif
    if (tid < K) {
        if (a.indices) b = a.indices[(size_t)fr * K + tid];
        else if (DEC_HDR_NT) b = __builtin_amdgcn_raw_buffer_load_b8(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.rx) + (size_t)fr * a.rx_frame_bytes, 0, 0x7fffffff, 0x00020000), (unsigned)tid * 512u + 2u, 0, 2);
        else b = a.rx[(size_t)fr * a.rx_frame_bytes + (size_t)tid * 512 + 2];
    }
    {
        // ALL of the workgroup's table loads are issued before the first is written to LDS: written as copy loops every iteration was
        // load, wait, write -- five global round trips one after the other, 8 us in front of everything (tools/experiments_r06/
        // dec_timeline.py, stamp set 1: this barrier stood at 8.4 us)
        static_assert(GF_NT == 256 && 2 * FFT_NTAB == 384, "two loads per thread and table");
        uint4_t *lt = reinterpret_cast<uint4_t *>(ldsraw);
        uint4_t *tab = reinterpret_cast<uint4_t *>(ldsraw + FFT_TAB_BYTES);
        const uint4_t *src = reinterpret_cast<const uint4_t *>(a.fft_tables), *mt = reinterpret_cast<const uint4_t *>(a.tab);
        const uint4_t z = {0u, 0u, 0u, 0u};
        const uint4_t f0 = src[tid], f1 = tid < 128 ? src[256 + tid] : z;
        const uint4_t m0 = mt[tid], m1 = mt[256 + tid];
        const uint4_t el = tid < 64 ? reinterpret_cast<const uint4_t *>(a.explog)[tid] : z;
        lt[tid] = f0;
        if (tid < 128) lt[256 + tid] = f1;
        tab[tid] = m0; tab[256 + tid] = m1;
        if (tid < 64) reinterpret_cast<uint4_t *>(s->exp)[tid] = el;
    }
    s->cnt[tid] = 0;
    if (tid < 4 * DEC128_MAXN) s->lpx[tid] = 0; // (lpx, lqx, lpy, lqy: consecutive)
    __syncthreads();
    PLAN_STAMP(1);
    if (tid < K) atomicAdd(&s->cnt[b], 1);
    __syncthreads();
    PLAN_STAMP(2);
    const int wv = tid >> 6, ln = tid & 63;
    const bool is_rec = b >= K;
    unsigned long long br = 0ull, bm = 0ull;
    if (tid < K) { // (waves 0 and 1, whole)
        br = __ballot(is_rec);
        bm = __ballot(s->cnt[tid] == 0);
        const unsigned long long bd = __ballot(s->cnt[tid] > 1 || s->cnt[K + tid] > 1); // an original, or a recovery block, more than once
        if (ln == 0) { s->mask[wv][0] = br; s->mask[wv][1] = bm; s->mask[wv][2] = bd; }
    }
    __syncthreads();
    const int nrec = __popcll(s->mask[0][0]) + __popcll(s->mask[1][0]), nmiss = __popcll(s->mask[0][1]) + __popcll(s->mask[1][1]);
    const bool dup = (s->mask[0][2] | s->mask[1][2]) != 0ull;
    const int N = nrec;
    bool ok = N > 0 && !dup;
    if (N > a.max_rows) { // more recovery blocks than the caller promised: left as received and COUNTED (gf_decode_plan_kernel)
        ok = false;
        if (tid == 0) atomicAdd(a.stats, 1u);
    }
    int rrank = 0, mrank_of_tid = K;
    if (tid < K) {
        const unsigned long long below = (1ull << ln) - 1ull;
        rrank = (wv ? __popcll(s->mask[0][0]) : 0) + __popcll(br & below);
        const int mrank = (wv ? __popcll(s->mask[0][1]) : 0) + __popcll(bm & below);
        if (s->cnt[tid] == 0) mrank_of_tid = mrank; // original `tid` is the mrank-th missing one (ascending): row mrank of the inverse restores it
        if (is_rec) { s->x[rrank] = (uint8_t)b; s->rpos[rrank] = (uint8_t)tid; }
        if (s->cnt[tid] == 0 && mrank < nrec) s->y[mrank] = (uint8_t)tid; // erased originals, ascending
        pl->inv[tid] = (int16_t)-1;
        pl->rowidx[tid] = 255;
    }
    if (tid == 0) { pl->n = ok ? N : 0; pl->m1 = (ok && N == 1) ? 1 : 0; pl->maxrow = 0; pl->pad = (!ok && nmiss > 0) ? 1 : 0; if (ok && N == 1) pl->minv[0] = 1; }
    __syncthreads();
    PLAN_STAMP(3);
    if (tid < K) {
        if (b < K) pl->inv[b] = (int16_t)tid; // (a repeated original: any copy)
        else if (ok) { pl->rowidx[b - K] = (uint8_t)rrank; pl->rpos[rrank] = (uint8_t)tid; atomicMax(&pl->maxrow, b - K); }
        if (ok && tid < N) {
            // strict mode: the reference copies back only the descriptors [128 - recoveryCount, 128) (SDRdaemonFECBuffer.cpp:204-211)
            const bool hole = a.strict && (int)s->rpos[tid] < K - N;
            pl->ydst[tid] = (uint8_t)(s->y[tid] | (hole ? 0x80 : 0));
        }
    }
    if (a.srcmap) {
        // no-copy mode (the Tx pipe on K5w): nothing is copied; the interpolator finds original `tid` of this frame through the map --
        // where it arrived (super block slot of the received frames), the slot this launch restores it into (row mrank of the
        // inverse -> slot mrank of the frame's restored blocks), or the all-zero slot behind the last frame's (it never came and
        // cannot be restored: initDecodeSlot's zero fill, SDRdaemonFECBuffer.cpp:109)
        __syncthreads(); // (pl->inv is complete)
        if (tid >= 1 && tid < K) {
            const int pos = pl->inv[tid];
            unsigned code;
            if (pos >= 0) code = (unsigned)fr * 128u + (unsigned)pos;
            else if (ok && mrank_of_tid < N) code = 0x80000000u | ((unsigned)fr * (unsigned)a.restored_rows + (unsigned)mrank_of_tid);
            else code = 0x80000000u | ((unsigned)a.nframes * (unsigned)a.restored_rows);
            a.srcmap[(size_t)fr * 128u + tid] = code;
        }
    }
    PLAN_STAMP(4);
    __syncthreads();
}

// ... second part, for frames with N >= 2 (pl->n; workgroup-uniform): the N x N inverse.  Called by all four waves BEHIND the issue of their
// 64 data loads (round 6): nothing in here touches what the loads return, the barriers wait for LDS only, so these ~2.4 us of LDS
// latency chains run while the received blocks are on their way instead of in front of them.
__device__ __forceinline__ void dec128_plan_inverse(Dec128Plan *pl, Dec128Scratch *s, int N)
{
    constexpr int K = 128;
    const int tid = threadIdx.x;
    // Minv[t][i] = PX_i PY_t / ((x_i ^ y_t) QX_i QY_t (y_t ^ 128)), PX_i = prod_k (x_i ^ y_k), PY_t = prod_k (x_k ^ y_t), QX_i = prod_{k != i}
    // (x_i ^ x_k), QY_t = prod_{k != t} (y_t ^ y_k) (gf_decode_plan_kernel's closed form of the Cauchy inverse): the four logarithm
    // sums per index as LDS atomics over the N x N pairs -- a pair per thread instead of a serial N-step loop on N threads
    // (every sum is filed under v = the thread's lane mod 32 -- two lanes per address in a wave's atomic; filed under u, as the first
    // version had three of the four, all 32 lanes of a half wave met in one address: ~600 clocks per atomic, 4.3 us for this loop.
    // PX_i: the pair (u, v) contributes log(x_v ^ y_u) to i = v; QX and QY are symmetric in their pair.)  A thread's four pairs share
    // v: their terms are summed in registers, one atomic per sum.
    {
        const int v = tid & (DEC128_MAXN - 1);
        int spx = 0, spy = 0, sqx = 0, sqy = 0;
        if (v < N) {
            const int xv = s->x[v], yv = s->y[v];
#pragma unroll
            for (int u = tid >> 5; u < DEC128_MAXN; u += GF_NT / DEC128_MAXN) {
                if (u < N) {
                    const int xu = s->x[u], yu = s->y[u];
                    spy += (int)s->log[xu ^ yv];
                    spx += (int)s->log[xv ^ yu];
                    if (u != v) { sqx += (int)s->log[xu ^ xv]; sqy += (int)s->log[yu ^ yv]; }
                }
            }
            atomicAdd(&s->lpx[v], spx);
            atomicAdd(&s->lpy[v], spy);
            atomicAdd(&s->lqx[v], sqx);
            atomicAdd(&s->lqy[v], sqy);
        }
    }
    __syncthreads();
    PLAN_STAMP(5);
#pragma unroll
    for (int e = tid; e < DEC128_MAXN * DEC128_MAXN; e += GF_NT) {
        const int t = e >> 5, i = e & (DEC128_MAXN - 1);
        if (t < N && i < N) {
            const int yt = s->y[t], xi = s->x[i];
            const int num = s->lpx[i] + s->lpy[t];
            const int den = s->log[xi ^ yt] + s->lqx[i] + s->lqy[t] + s->log[yt ^ K];
            pl->minv[i * DEC128_MAXN + (t & 3) * 8 + (t >> 2)] = s->exp[(unsigned)(num + 255 * 80 - den) % 255u]; // (den <= 2 * 254 + 2 * 31 * 254 < 255 * 80)
        }
    }
    __syncthreads();
    PLAN_STAMP(6);
}

template <bool FUSED, bool NOCOPY = false> __device__ __forceinline__ void gf_decode128_fft_wg(const Dec128Args &a, int fr, unsigned char *ldsraw)
{
    Dec128Plan *pl = reinterpret_cast<Dec128Plan *>(ldsraw + FFT_TAB_BYTES + 256 * 32);
    FFT_STAMP(0);
    if constexpr (FUSED) {
        dec128_plan(a, fr, pl, reinterpret_cast<Dec128Scratch *>(ldsraw + DEC128_FFT_LDS_BYTES), ldsraw); // (tables + plan; ends with a barrier)
    } else {
        // (all loads first, then the LDS writes: dec128_plan)
        static_assert(GF_NT == 256 && 2 * FFT_NTAB == 384 && DEC128_PLAN_BYTES / 16 <= GF_NT, "loads per thread");
        uint4_t *lt = reinterpret_cast<uint4_t *>(ldsraw);
        uint4_t *tab = reinterpret_cast<uint4_t *>(ldsraw + FFT_TAB_BYTES); // all 256 constants: 8 dwords each
        const int tid = threadIdx.x;
        const uint4_t *src = reinterpret_cast<const uint4_t *>(a.fft_tables), *mt = reinterpret_cast<const uint4_t *>(a.tab);
        const uint4_t z = {0u, 0u, 0u, 0u};
        const uint4_t f0 = src[tid], f1 = tid < 128 ? src[256 + tid] : z;
        const uint4_t m0 = mt[tid], m1 = mt[256 + tid];
        const uint4_t pr = tid < DEC128_PLAN_BYTES / 16 ? reinterpret_cast<const uint4_t *>(a.plan + (size_t)fr * DEC128_PLAN_BYTES)[tid] : z;
        lt[tid] = f0;
        if (tid < 128) lt[256 + tid] = f1;
        tab[tid] = m0; tab[256 + tid] = m1;
        if (tid < DEC128_PLAN_BYTES / 16) reinterpret_cast<uint4_t *>(pl)[tid] = pr;
        __syncthreads();
    }
    const int stagger_phase = fec_stagger_sleep(fr, a.stagger, a.stagger_div);
    FFT_STAMP(1);
    INV_STAMP(1);
#ifdef FFT_STAMPS
    if (FFT_STAMP_SET == 0 && (threadIdx.x & 63) == 0 && blockIdx.x < 2048) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); g_fft_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 6] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xfu) << 32) | ((unsigned long long)stagger_phase << 40); }
#endif
    (void)stagger_phase;
    // (the block half is a template parameter of everything behind this point, like the encoder's: gf_encode128_fft_wave)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wv >> 1) gf_decode128_fft_wave<1, FUSED, NOCOPY>(a, fr, ldsraw, wv & 1);
    else gf_decode128_fft_wave<0, FUSED, NOCOPY>(a, fr, ldsraw, wv & 1);
}

template <int HF, bool FUSED, bool NOCOPY> __device__ __forceinline__ void gf_decode128_fft_wave(const Dec128Args &a, int fr, unsigned char *ldsraw, int ch)
{
    constexpr int hf = HF;
    unsigned *tab = reinterpret_cast<unsigned *>(ldsraw + FFT_TAB_BYTES); // all 256 constants: 8 dwords each
    Dec128Plan *pl = reinterpret_cast<Dec128Plan *>(ldsraw + FFT_TAB_BYTES + 256 * 32);
    unsigned *xall = reinterpret_cast<unsigned *>(ldsraw + FFT_TAB_BYTES + 256 * 32 + DEC128_PLAN_BYTES);
    const unsigned la = lds_addr(ldsraw);
    unsigned *const xch0 = xall + ch * FFT_XCH_DWORDS;
    const int N = __builtin_amdgcn_readfirstlane(pl->n), m1 = __builtin_amdgcn_readfirstlane(pl->m1);
    const int zm = FUSED ? __builtin_amdgcn_readfirstlane(pl->pad) : 0; // (fused plan: an incomplete frame's missing blocks are zeroed by the copy loop)
    // no-copy mode (fused plan only, Dec128Args::srcmap): the received originals stay where they are, the restored ones go to the
    // frame's slots of a.restored (row t -> slot t); block 0 (the meta block) still goes to block0_out when the caller wants it
    // (a template parameter: as a run-time flag the two variants' stores met in joins the register allocator answered with 74 spills)
    static_assert(!NOCOPY || FUSED, "no-copy mode needs the fused plan (it writes the map)");
    constexpr bool nocopy = NOCOPY;
    // descriptors: the frame as it was received (payload of the block at position 0 = byte 4), the payload area (block 1's samples
    // = byte 0).  Offsets with bit 31 set lie beyond their range: such loads return zero, such stores are dropped -- that is how
    // the erased originals read as zero and how lane 63 of the second column half (no column) stores nothing, without a branch.
    const __amdgpu_buffer_rsrc_t rrx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.rx) + (size_t)fr * a.rx_frame_bytes + 4, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rpay = __builtin_amdgcn_make_buffer_rsrc(nocopy ? a.restored + (size_t)fr * (size_t)a.restored_rows * 508u : a.payload_out + (size_t)fr * a.payload_frame_bytes,
                                                                          0, 0x7fffffff, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned d[64], e[16], rec[16];
    {
        const unsigned lane = fft_lane(), col = (unsigned)ch * 64u + lane;
        const bool live = col < 127u;
        const unsigned lc4 = 4u * (live ? col : 126u);
        const unsigned st4 = live ? lc4 : OOB;
        const int b0 = 64 * hf;
        if (NOCOPY && N == 0) { // (workgroup-uniform, in front of every barrier) nothing to restore, nothing to copy -- but the meta block
            if (hf == 0 && a.block0_out && live) {
                const int p0 = __builtin_amdgcn_readfirstlane((int)pl->inv[0]);
                reinterpret_cast<unsigned *>(a.block0_out + (size_t)fr * 508)[col] = p0 < 0 ? 0u : __builtin_amdgcn_raw_buffer_load_b32(rrx, lc4, (p0 & 127) * 512, 0);
            }
            return;
        }
        // the wave's 64 entries of the position map in ONE register (lane L: original b0 + L), handed out with v_readlane: read block
        // by block from LDS the compiler hoisted all 64 reads to the top -- 64 registers of temporaries under the 64 data loads,
        // whose first dozen then went to scratch straight from the load (a wait for each in the middle of the burst)
        const int invv = (int)pl->inv[b0 + (int)lane];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int pos = __builtin_amdgcn_readlane(invv, i); // position of original b0 + i in the received array, -1 = erased
            // (erased: an offset beyond the descriptor's range -- the load returns zero; a select on the loaded value here would stand
            // between the loads and the plan's second part and wait for the data)
            d[i] = __builtin_amdgcn_raw_buffer_load_b32(rrx, pos < 0 ? OOB : lc4, (pos < 0 ? 0 : pos & 127) * 512, DEC_LOAD_AUX);
        }
        if constexpr (FUSED) {
            if (N >= 2) dec128_plan_inverse(pl, reinterpret_cast<Dec128Scratch *>(ldsraw + DEC128_FFT_LDS_BYTES), N); // (uniform; barriers inside)
        }
        // the received originals go to their places (getSlotData's layout: blocks 1..127 back to back, block 0 apart) -- the first
        // 32 of the wave's blocks here, the other 32 between the two halves of the transform (fft_rows16's hook: they may still be
        // on their way while the first half runs)
        auto copy_out = [&](auto first, auto last) __attribute__((always_inline)) {
#pragma unroll
            for (int i = decltype(first)::value; i < decltype(last)::value; ++i) {
                const int j = b0 + i; // (uniform)
                const int pos = __builtin_amdgcn_readlane(invv, i);
                if constexpr (NOCOPY) { if (!(i == 0 && hf == 0)) continue; } // (no-copy mode: only the meta block travels)
                if (pos < 0 && !zm) continue; // (uniform; nothing is defined in here: no join of register values)
                // (pos < 0 with zm: the block never arrived and nothing will restore it -- d[i] is zero -- initDecodeSlot's zero fill, .cpp:109)
                if (i == 0 && hf == 0) {
                    if (a.block0_out && live) reinterpret_cast<unsigned *>(a.block0_out + (size_t)fr * 508)[col] = d[0];
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(d[i], rpay, st4, (j - 1) * 508, 0);
                }
            }
        };
        if (N == 0 || m1) { // (copy only / cm256's DecodeM1; workgroup-uniform, in front of the barriers; this path ENDS here: no join)
            copy_out(std::integral_constant<int, 0>{}, std::integral_constant<int, 64>{});
            if (N == 0) return;
            // cm256's DecodeM1: one recovery block, the erased original is the XOR of everything received (Minv = 1)
            unsigned par = 0u;
#pragma unroll
            for (int i = 0; i < 64; i += 2) par = x3(par, d[i], d[i + 1]);
            (xch0 + lane)[(32 + hf) * 64] = par;
            __syncthreads();
            if (hf == 0) {
                const unsigned P = (xch0 + lane)[32 * 64] ^ (xch0 + lane)[33 * 64];
                const int rp = __builtin_amdgcn_readfirstlane((int)pl->rpos[0]) & 127;
                const unsigned rec = __builtin_amdgcn_raw_buffer_load_b32(rrx, lc4, rp * 512, 0);
                const int yy = __builtin_amdgcn_readfirstlane((int)pl->ydst[0]), y = yy & 0x7f;
                const unsigned val = (yy & 0x80) ? 0u : (P ^ rec); // (strict mode: a block the reference's copy-back would miss stays a hole)
                if (y >= 1) __builtin_amdgcn_raw_buffer_store_b32(val, rpay, st4, nocopy ? 0 : (y - 1) * 508, 0);
                else if (a.block0_out && live) reinterpret_cast<unsigned *>(a.block0_out + (size_t)fr * 508)[col] = val;
            }
            return;
        }
        copy_out(std::integral_constant<int, 0>{}, std::integral_constant<int, 32>{});
        unsigned par = 0u;
#pragma unroll
        for (int i = 0; i < 32; i += 2) par = x3(par, d[i], d[i + 1]);
        fft_rows16<HF>(d, e, la, xch0, [&]() __attribute__((always_inline)) {
            copy_out(std::integral_constant<int, 32>{}, std::integral_constant<int, 64>{});
#pragma unroll
            for (int i = 32; i < 64; i += 2) par = x3(par, d[i], d[i + 1]);
            (xch0 + lane)[(32 + hf) * 64] = par;
        }, [&]() __attribute__((always_inline)) {
            // the received recovery rows among rows 16 hf .. 16 hf + 15, on their way while the exchange and the size-16 transform run
            const unsigned ld4 = live ? lc4 : OOB;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ri = __builtin_amdgcn_readfirstlane((int)pl->rowidx[16 * hf + i]); // 255 = not received
                const int rp = ri == 255 ? 0 : __builtin_amdgcn_readfirstlane((int)pl->rpos[ri & 31]) & 127; // (a row that did not arrive: block 0, not used)
                rec[i] = __builtin_amdgcn_raw_buffer_load_b32(rrx, ld4, rp * 512, DEC_LOAD_AUX);
            }
        });
    }

    unsigned *const syn = xch0; // [i][lane]
    {
        // syndromes of the received recovery rows among rows 16 hf .. 16 hf + 15: recovery ^ (P ^ (r c / q) * value_r)
        const unsigned lane = fft_lane(), col = (unsigned)ch * 64u + lane;
        const unsigned P = (xch0 + lane)[32 * 64] ^ (xch0 + lane)[33 * 64];
        __syncthreads(); // (both waves are through with the exchange rows: the syndromes take their place)
        {
            FftTabs R;
            const unsigned lk = la + (unsigned)hf * 512u;
            fft_issue<0, 160>(R, lk);
            fft_for<16>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, PP = i & 1;
                fft_wait<PP>(R);
                if constexpr (i + 1 < 16) fft_issue<PP ^ 1, 160 + i + 1>(R, lk);
                const int ri = __builtin_amdgcn_readfirstlane((int)pl->rowidx[16 * hf + i]);
                if (ri != 255) {
                    unsigned v = rec[i] ^ P;
                    fft_muladd<PP>(v, e[i], R);
                    (syn + lane)[(ri & 31) * 64] = col < 127u ? v : 0u;
                }
            });
        }
        __syncthreads();
    }
    FFT_STAMP(5);

    // erased originals = Minv x syndromes: this wave takes rows t = w + 4 u of its column half for w = hf and w = hf + 2 (up to 8 each);
    // a syndrome dword is split into its selector words once and multiplied by the constants of all the wave's rows.  Two
    // syndromes at a time, the next two (and the wave's constants for them: two 8-byte reads per syndrome) already on their way.
    const unsigned lane = fft_lane(), col = (unsigned)ch * 64u + lane;
    unsigned acc[2][DEC128_MAXN / 4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int u = 0; u < DEC128_MAXN / 4; ++u) acc[g][u] = 0u;
    const int nm0 = (N - hf + 3) >> 2, nm1 = (N - (hf + 2) + 3) >> 2; // rows w + 4 u < N
    unsigned sy[2];
    uint2_t mc[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        sy[k] = (syn + lane)[k * 64];
#pragma unroll
        for (int g = 0; g < 2; ++g) mc[k][g] = *reinterpret_cast<const uint2_t *>(&pl->minv[k * DEC128_MAXN + (hf + 2 * g) * 8]);
    }
#pragma unroll 1
    for (int i0 = 0; i0 < N; i0 += 2) {
        unsigned sy_next[2];
        uint2_t mc_next[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = (i0 + 2 + k) & (DEC128_MAXN - 1); // (past the end: rows that exist and are not used)
            sy_next[k] = (syn + lane)[i * 64];
#pragma unroll
            for (int g = 0; g < 2; ++g) mc_next[k][g] = *reinterpret_cast<const uint2_t *>(&pl->minv[i * DEC128_MAXN + (hf + 2 * g) * 8]);
        }
        const bool two = i0 + 1 < N; // (uniform)
        const Sel sl0 = make_sel(sy[0]), sl1 = make_sel(two ? sy[1] : 0u);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int u = 0; u < DEC128_MAXN / 4; ++u) {
                if (u < (g ? nm1 : nm0)) {
                    const unsigned c0 = ((u < 4 ? mc[0][g].x : mc[0][g].y) >> (8 * (u & 3))) & 0xffu;
                    const unsigned c1 = ((u < 4 ? mc[1][g].x : mc[1][g].y) >> (8 * (u & 3))) & 0xffu;
                    // (a second syndrome that does not exist has the selector words of zero: its product is zero whatever the constant)
                    acc[g][u] = __builtin_amdgcn_bitop3_b32(acc[g][u], mulc(sl0, *reinterpret_cast<const uint4_t *>(&tab[c0 * 8]), tab[c0 * 8 + 4]),
                                                            mulc(sl1, *reinterpret_cast<const uint4_t *>(&tab[c1 * 8]), tab[c1 * 8 + 4]), 0x96);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            sy[k] = sy_next[k];
#pragma unroll
            for (int g = 0; g < 2; ++g) mc[k][g] = mc_next[k][g];
        }
    }
    const unsigned st4 = col < 127u ? 4u * col : OOB;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int u = 0; u < DEC128_MAXN / 4; ++u) {
            if (u < (g ? nm1 : nm0)) {
                const int yy = __builtin_amdgcn_readfirstlane((int)pl->ydst[hf + 2 * g + 4 * u]), y = yy & 0x7f;
                const unsigned val = (yy & 0x80) ? 0u : acc[g][u]; // (strict mode: a block the reference's copy-back would miss stays a hole)
                if (y >= 1) __builtin_amdgcn_raw_buffer_store_b32(val, rpay, st4, nocopy ? (hf + 2 * g + 4 * u) * 508 : (y - 1) * 508, 0);
                else if (a.block0_out && col < 127u) reinterpret_cast<unsigned *>(a.block0_out + (size_t)fr * 508)[col] = val;
            }
        }
    }
    FFT_STAMP(7);
}
