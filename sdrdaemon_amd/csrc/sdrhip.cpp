// sdrhip.cpp -- host side of libsdrhip.so: the C ABI declared in include/sdrhip.h.
// C++11-style host code calling HIP; no torch, no oracle, no CPU fallback: without a
// usable GPU every compute entry point fails with SDRHIP_EDEVICE.
#include "sdrhip_host.h"
#include "gf256.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace sdrhip;

namespace {
thread_local std::string g_err;
}

namespace sdrhip {

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int DevBuf::reserve(size_t n)
{
    if (n <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 256;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return fail(SDRHIP_ENOMEM, "hipMalloc(%zu) failed", want); }
    cap = want;
    return 0;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

int PinnedBuf::reserve(size_t n)
{
    if (pending) { (void)hipEventSynchronize(ev); pending = false; }
    if (n <= cap) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 2 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return fail(SDRHIP_ENOMEM, "hipHostMalloc(%zu) failed", want); }
    cap = want;
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(SDRHIP_EDEVICE, "hipEventCreate");
    return 0;
}

void PinnedBuf::mark(hipStream_t s)
{
    if (ev && hipEventRecord(ev, s) == hipSuccess) pending = true;
}

void PinnedBuf::release()
{
    if (pending) (void)hipEventSynchronize(ev);
    if (p) (void)hipHostFree(p);
    if (ev) (void)hipEventDestroy(ev);
    p = nullptr; cap = 0; ev = nullptr; pending = false;
}

} // namespace sdrhip

extern "C" const char *sdrhip_last_error(void) { return g_err.c_str(); }

extern "C" int sdrhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace sdrhip {
static void ctx_free(sdrhip_ctx *c);
static void drop_kernel_events(sdrhip_ctx *c)
{
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    for (int k = 0; k < 4; ++k) {
        for (auto &pr : c->kev[k]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        c->kev[k].clear();
    }
}
}

extern "C" int sdrhip_ctx_create(int device, void *hip_stream, sdrhip_ctx **out)
{
    if (!out) return fail(SDRHIP_EINVAL, "ctx_create: out is NULL");
    *out = nullptr;
    int n = sdrhip_device_count();
    if (n <= 0) return fail(SDRHIP_EDEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(SDRHIP_EINVAL, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    sdrhip_ctx *c = new (std::nothrow) sdrhip_ctx();
    if (!c) return fail(SDRHIP_ENOMEM, "out of host memory");
    c->device = device;
    c->stream = static_cast<hipStream_t>(hip_stream);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
        // the environment is read here, once (getenv is not safe against a concurrent setenv): later changes go through
        // sdrhip_ctx_set_option()
        static const char *const keys[][2] = {{"SDRHIP_DECIM_PATH", "decim_path"}, {"SDRHIP_MFMA_SPAN", "mfma_span"}, {"SDRHIP_MFMA_MIN", "mfma_min"},
                                              {"SDRHIP_INTERP_PATH", "interp_path"}, {"SDRHIP_INTERP_SPAN", "interp_span"}, {"SDRHIP_RX_FUSED", "rx_fused"}, {"SDRHIP_RX_DIRECT", "rx_direct"},
                                              {"SDRHIP_DEC_PATH", "dec_path"}, {"SDRHIP_ENC_PATH", "enc_path"}, {"SDRHIP_ENC_MIN_ROWS", "enc_min_rows"}, {"SDRHIP_MFMA_RING", "mfma_ring"}, {"SDRHIP_TX_OVERLAP", "tx_overlap"}, {"SDRHIP_RX_WINDOW", "rx_window"}, {"SDRHIP_FEC_STAGGER", "fec_stagger"}, {"SDRHIP_FEC_STAGGER_MOD", "fec_stagger_mod"}, {"SDRHIP_DEC_PLAN", "dec_plan"}, {"SDRHIP_TX_GATHER", "tx_gather"}, {"SDRHIP_ENC_UNITS", "enc_units"}};
        for (size_t i = 0; i < sizeof(keys) / sizeof(keys[0]); ++i)
            if (const char *v = getenv(keys[i][0])) (void)sdrhip_ctx_set_option(c, keys[i][1], v);
    }
    std::vector<uint8_t> tab(256 * 32);
    gf_build_tables(tab.data());
    if (hipMalloc(reinterpret_cast<void **>(&c->gf_tab), tab.size()) != hipSuccess) { c->gf_tab = nullptr; ctx_free(c); return fail(SDRHIP_ENOMEM, "hipMalloc gf tables"); }
    if (hipMemcpy(c->gf_tab, tab.data(), tab.size(), hipMemcpyHostToDevice) != hipSuccess) { ctx_free(c); return fail(SDRHIP_EDEVICE, "upload gf tables"); }
    std::vector<uint8_t> em(128 * 128);
    cm256_encode_matrix(128, 128, em.data());
    if (hipMalloc(reinterpret_cast<void **>(&c->enc_matrix), em.size()) != hipSuccess ||
        hipMemcpy(c->enc_matrix, em.data(), em.size(), hipMemcpyHostToDevice) != hipSuccess) { ctx_free(c); return fail(SDRHIP_ENOMEM, "upload encode matrix"); }
    std::vector<uint8_t> kl(8 * 81 * 32);
    cm256_karatsuba_leaf_tables(kl.data());
    if (hipMalloc(reinterpret_cast<void **>(&c->enc_leaves), kl.size()) != hipSuccess ||
        hipMemcpy(c->enc_leaves, kl.data(), kl.size(), hipMemcpyHostToDevice) != hipSuccess) { ctx_free(c); return fail(SDRHIP_ENOMEM, "upload encoder constants"); }
    {
        std::vector<uint8_t> ft((size_t)CM256_FFT_TABLES * 32);
        cm256_fft_tables(ft.data());
        if (hipMalloc(reinterpret_cast<void **>(&c->enc_fft), ft.size()) != hipSuccess ||
            hipMemcpy(c->enc_fft, ft.data(), ft.size(), hipMemcpyHostToDevice) != hipSuccess) { ctx_free(c); return fail(SDRHIP_ENOMEM, "upload encoder FFT constants"); }
    }
    {
        const GF256 &g = gf();
        uint8_t el[1024];
        memcpy(el, g.exp, 512);
        memcpy(el + 512, g.log, 512);
        if (hipMalloc(reinterpret_cast<void **>(&c->gf_explog), sizeof(el)) != hipSuccess ||
            hipMemcpy(c->gf_explog, el, sizeof(el), hipMemcpyHostToDevice) != hipSuccess) { ctx_free(c); return fail(SDRHIP_ENOMEM, "upload GF(256) exp/log tables"); }
    }
    if (hipMalloc(reinterpret_cast<void **>(&c->dec_stats), 64) != hipSuccess ||
        hipMemset(c->dec_stats, 0, 64) != hipSuccess) { ctx_free(c); return fail(SDRHIP_ENOMEM, "hipMalloc decoder counters"); }
    if (hipMalloc(reinterpret_cast<void **>(&c->decim_dump), 4096) != hipSuccess) { c->decim_dump = nullptr; ctx_free(c); return fail(SDRHIP_ENOMEM, "hipMalloc decimator scratch"); }
    if (hipMalloc(reinterpret_cast<void **>(&c->fused_roles), SDRHIP_FUSED_ROLE_WORDS * 4) != hipSuccess ||
        hipMemset(c->fused_roles, 0, SDRHIP_FUSED_ROLE_WORDS * 4) != hipSuccess) { ctx_free(c); return fail(SDRHIP_ENOMEM, "hipMalloc role table"); }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { ctx_free(c); return fail(SDRHIP_EDEVICE, "hipEventCreate"); }
    *out = c;
    return SDRHIP_OK;
}

namespace sdrhip {
int ctx_stream2(sdrhip_ctx *c, hipStream_t *out)
{
    if (!c->stream2) {
        // lowest priority: when both streams have workgroups to place, the first stream's (the long kernel) go first
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
        if (hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo) != hipSuccess) {
            c->stream2 = nullptr;
            return fail(SDRHIP_EDEVICE, "hipStreamCreate (second stream)");
        }
    }
    *out = c->stream2;
    return SDRHIP_OK;
}
void ctx_retain(sdrhip_ctx *c) { c->refs.fetch_add(1); }
void ctx_release(sdrhip_ctx *c)
{
    if (c->refs.fetch_sub(1) == 1 && c->dying) ctx_free(c);
}
} // namespace sdrhip

// Handles created on a context keep it alive: destroying the context first only marks it; the
// last handle to go frees it (destruction order is then irrelevant, e.g. under a garbage collector).
extern "C" void sdrhip_ctx_destroy(sdrhip_ctx *c)
{
    if (!c || c->dying.exchange(true)) return;
    if (c->refs.load() == 0) ctx_free(c);
}

// Kernel-path knobs (tests and tools; the defaults come from the environment at creation).  Unknown keys / values: EINVAL.
extern "C" int sdrhip_ctx_set_option(sdrhip_ctx *c, const char *key, const char *value)
{
    if (!c || !key || !value) return fail(SDRHIP_EINVAL, "ctx_set_option: NULL argument");
    sdrhip::CtxLock lock_(c);
    const std::string k(key), v(value);
    char *end = nullptr;
    const unsigned long long num = strtoull(value, &end, 10);
    const bool isnum = *value && end && !*end;
    if (k == "decim_path") {
        if (v == "auto") c->opt.decim_path = DECIM_PATH_AUTO;
        else if (v == "valu") c->opt.decim_path = DECIM_PATH_VALU;
        else if (v == "mfma") c->opt.decim_path = DECIM_PATH_MFMA;
        else return fail(SDRHIP_EINVAL, "ctx_set_option: decim_path must be auto, valu or mfma");
    } else if (k == "interp_path") {
        if (v == "auto") c->opt.interp_wave = CtxOptions().interp_wave;
        else if (v == "valu") c->opt.interp_wave = 0;
        else if (v == "wave") c->opt.interp_wave = 1;
        else return fail(SDRHIP_EINVAL, "ctx_set_option: interp_path must be auto, valu or wave");
    } else if (k == "mfma_span" && isnum) c->opt.mfma_span = (size_t)num;
    else if (k == "mfma_min" && isnum) c->opt.mfma_min = (size_t)num;
    else if (k == "interp_span" && isnum) c->opt.interp_span = (size_t)num;
    else if (k == "rx_fused" && isnum && num <= 3) c->opt.rx_fused = (int)num;
    else if (k == "rx_fused" && v == "overlap") c->opt.rx_fused = 3;
    else if (k == "rx_direct" && isnum && num <= 1) c->opt.rx_direct = (int)num;
    else if (k == "rx_window" && isnum && num <= 8) c->opt.rx_window = (int)num;
    else if (k == "mfma_ring" && isnum && num >= 2 && num <= 4) c->opt.mfma_ring = (int)num;
    else if (k == "tx_overlap" && isnum && num <= 1) c->opt.tx_overlap = (int)num;
    else if (k == "fec_stagger" && isnum && num <= 64) c->opt.fec_stagger = (int)num;
    else if (k == "fec_stagger_mod" && isnum && num <= 116) c->opt.fec_stagger_mod = (int)num;
    else if (k == "tx_gather" && isnum && num <= 1) c->opt.tx_gather = (int)num;
    else if (k == "dec_plan") {
        if (v == "fused") c->opt.dec_fused_plan = 1;
        else if (v == "kernel") c->opt.dec_fused_plan = 0;
        else return fail(SDRHIP_EINVAL, "ctx_set_option: dec_plan must be fused or kernel");
    }
    else if (k == "enc_min_rows" && isnum && num >= 1 && num <= 32) c->opt.enc_min_rows = (int)num;
    else if (k == "enc_units" && (v == "frame" || v == "half")) c->opt.enc_half = v == "half";
    else if (k == "enc_path") {
        if (v == "fft") c->opt.enc_fft = 1;
        else if (v == "karatsuba") c->opt.enc_fft = 0;
        else return fail(SDRHIP_EINVAL, "ctx_set_option: enc_path must be fft or karatsuba");
    } else if (k == "dec_path") {
        if (v == "syndrome") c->opt.dec_syndrome = 1;
        else if (v == "dense") c->opt.dec_syndrome = 0;
        else return fail(SDRHIP_EINVAL, "ctx_set_option: dec_path must be syndrome or dense");
    } else if (k == "dec_max_rows" && isnum && num >= 1 && num <= 128) c->opt.dec_max_rows = (int)num;
    else if (k == "dec_strict" && isnum && num <= 1) c->opt.dec_strict = (int)num;
    else if (k == "ktime_stride" && isnum && num >= 1 && num <= 1024) { c->ktime_stride = (int)num; for (int i = 0; i < 4; ++i) c->ktime_stride_cls[i] = 0; }
    else if (k == "ktime_stride_class") { // "<class>:<stride>": this kernel class only (e.g. the roofline kernel on every launch, the others on every 4th)
        int cls = -1, st = 0;
        if (sscanf(value, "%d:%d", &cls, &st) != 2 || cls < 0 || cls > 3 || st < 1 || st > 1024) return fail(SDRHIP_EINVAL, "ctx_set_option: ktime_stride_class takes <class 0..3>:<stride 1..1024>");
        c->ktime_stride_cls[cls] = st;
    }
    else return fail(SDRHIP_EINVAL, "ctx_set_option: unknown key or malformed value: %s=%s", key, value);
    return SDRHIP_OK;
}

// Event counters kept on the device (read = one stream synchronisation + a 4-byte copy).
extern "C" int sdrhip_ctx_get_counter(sdrhip_ctx *c, const char *key, uint64_t *value)
{
    if (!c || !key || !value) return fail(SDRHIP_EINVAL, "ctx_get_counter: NULL argument");
    sdrhip::CtxLock lock_(c);
    if (std::string(key) != "dec_rows_exceeded") return fail(SDRHIP_EINVAL, "ctx_get_counter: unknown key: %s", key);
    HIP_TRY(hipSetDevice(c->device));
    unsigned v = 0;
    HIP_TRY(hipMemcpyAsync(&v, c->dec_stats, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *value = v;
    return SDRHIP_OK;
}

int sdrhip::enc128_min_rows(const sdrhip_ctx *c) { return c->opt.enc_fft ? c->opt.enc_min_rows : ENC128_MIN_ROWS; }

static void sdrhip::ctx_free(sdrhip_ctx *c)
{
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    drop_kernel_events(c);
    c->in.release(); c->out.release(); c->aux.release(); c->aux3.release();
    if (c->gf_tab) (void)hipFree(c->gf_tab);
    if (c->enc_matrix) (void)hipFree(c->enc_matrix);
    if (c->enc_leaves) (void)hipFree(c->enc_leaves);
    if (c->enc_fft) (void)hipFree(c->enc_fft);
    if (c->decim_dump) (void)hipFree(c->decim_dump);
    if (c->fused_roles) (void)hipFree(c->fused_roles);
    if (c->gf_explog) (void)hipFree(c->gf_explog);
    if (c->dec_stats) (void)hipFree(c->dec_stats);
    c->dec_plan.release();
    c->pin.release();
    c->zin.release(); c->zout.release();
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

extern "C" int sdrhip_ctx_synchronize(sdrhip_ctx *c)
{
    if (!c) return fail(SDRHIP_EINVAL, "ctx is NULL");
    sdrhip::CtxLock lock_(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
    return SDRHIP_OK;
}

extern "C" int sdrhip_ctx_timing_begin(sdrhip_ctx *c)
{
    if (!c) return fail(SDRHIP_EINVAL, "ctx is NULL");
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return SDRHIP_OK;
}

extern "C" int sdrhip_ctx_timing_end(sdrhip_ctx *c, float *ms)
{
    if (!c || !ms) return fail(SDRHIP_EINVAL, "ctx/ms is NULL");
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return SDRHIP_OK;
}

extern "C" int sdrhip_ctx_kernel_timing(sdrhip_ctx *c, int enable)
{
    if (!c) return fail(SDRHIP_EINVAL, "ctx is NULL");
    sdrhip::CtxLock lock_(c);
    c->ktime_on = enable != 0;
    for (int i = 0; i < 4; ++i) c->ktime_seen[i] = 0;
    if (!enable) drop_kernel_events(c); // pairs nobody read
    return SDRHIP_OK;
}

extern "C" int sdrhip_ctx_kernel_timing_read(sdrhip_ctx *c, int cls, double *total_ms, unsigned *launches)
{
    if (!c || cls < 0 || cls > 3 || !total_ms || !launches) return fail(SDRHIP_EINVAL, "kernel_timing_read: bad argument");
    sdrhip::CtxLock lock_(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
    double sum = 0;
    unsigned n = 0;
    for (auto &pr : c->kev[cls]) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { sum += ms; ++n; }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    c->kev[cls].clear();
    *total_ms = sum;
    *launches = n;
    return SDRHIP_OK;
}

// ------------------------------------------------------------------------------ decimators
struct sdrhip_decimators {
    sdrhip_ctx *ctx;
    int nstreams;
    int bias;
    int32_t *state[2]; // double buffered [nstreams][DEC_STATE_WORDS]
    int cur;
    bool stage0_int16; // history of m_decimator2 fits int16 (it last saw raw samples or nothing)
    sdrhip::DecimPlanInfo last; // what the last call launched
};

extern "C" int sdrhip_decimators_last_plan(const sdrhip_decimators *d, sdrhip_decim_plan *out)
{
    if (!d || !out) return fail(SDRHIP_EINVAL, "decimators_last_plan: NULL argument");
    sdrhip::CtxLock lock_(d->ctx);
    out->path = d->last.path; out->wps = d->last.wps; out->npieces = d->last.npieces; out->nseg = d->last.nseg;
    out->span = d->last.span; out->head = d->last.head; out->tail_start = d->last.tail_start;
    return SDRHIP_OK;
}

extern "C" int sdrhip_decimators_create(sdrhip_ctx *ctx, int nstreams, int hb_variant, sdrhip_decimators **out)
{
    if (!ctx || !out || nstreams <= 0 || nstreams > 65535) return fail(SDRHIP_EINVAL, "decimators_create: bad argument");
    if (hb_variant != SDRHIP_HB_EO1 && hb_variant != SDRHIP_HB_DB) return fail(SDRHIP_EINVAL, "hb_variant must be SDRHIP_HB_EO1 or SDRHIP_HB_DB");
    HIP_TRY(hipSetDevice(ctx->device));
    sdrhip_decimators *d = new (std::nothrow) sdrhip_decimators();
    if (!d) return fail(SDRHIP_ENOMEM, "out of host memory");
    d->ctx = ctx; d->nstreams = nstreams; d->bias = hb_variant; d->cur = 0; d->stage0_int16 = true;
    size_t bytes = (size_t)nstreams * DEC_STATE_WORDS * sizeof(int32_t);
    d->state[0] = d->state[1] = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d->state[0]), bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&d->state[1]), bytes) != hipSuccess) {
        if (d->state[0]) (void)hipFree(d->state[0]);
        delete d;
        return fail(SDRHIP_ENOMEM, "hipMalloc decimator state");
    }
    ctx_retain(ctx);
    *out = d;
    return sdrhip_decimators_reset(d);
}

extern "C" void sdrhip_decimators_destroy(sdrhip_decimators *d)
{
    if (!d) return;
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    (void)hipFree(d->state[0]);
    (void)hipFree(d->state[1]);
    ctx_release(d->ctx);
    delete d;
}

extern "C" int sdrhip_decimators_reset(sdrhip_decimators *d)
{
    if (!d) return fail(SDRHIP_EINVAL, "decimators is NULL");
    sdrhip::CtxLock lock_(d->ctx);
    size_t bytes = (size_t)d->nstreams * DEC_STATE_WORDS * sizeof(int32_t);
    HIP_TRY(hipMemsetAsync(d->state[0], 0, bytes, d->ctx->stream)); // ctor zero fill, EO1.h:171-188
    HIP_TRY(hipMemsetAsync(d->state[1], 0, bytes, d->ctx->stream));
    d->cur = 0;
    d->stage0_int16 = true;
    return SDRHIP_OK;
}

namespace {
// A launch of the matrix-core kernel lasts at least one warm-up + the shortest span, i.e. ~20 us at decimate16 and
// twice as long per further stage, however small the call; below these sizes the VALU kernel is faster
// (tools/bench_small.py): 2^22 samples over all streams up to decimate8, 2^23 for decimate16, 2^24, 2^25.
size_t mfma_min_samples(const CtxOptions &o, int log2decim) { return o.mfma_min << (log2decim > 3 ? log2decim - 3 : 0); }
} // namespace

namespace sdrhip {
// device-pointer core shared with the fused Rx pipe; frame_* = 0 for plain output
int decimate_device(sdrhip_decimators *d, int log2decim, int fcpos, unsigned *sampleSize, const int16_t *in,
                    size_t n_in, size_t in_stride, int16_t *out, size_t out_stride, size_t *n_out, int frame_mode,
                    int frame_blocks, uint64_t frame_sample_base, const RxMeta *meta, const Enc128Args *fuse, bool *fused, bool coresident)
{
    if (fused) *fused = false;
    sdrhip_ctx *c = d->ctx;
    const unsigned L = (unsigned)log2decim;
    const unsigned ss = *sampleSize;
    if (L == 0) {
        // Downsampler::process m_decim == 0: copy + decimate1 (Downsampler.cpp:76-80, Decimators.cpp:22-35)
        if (n_out) *n_out = n_in;
        d->last = DecimPlanInfo(); // (no cascade launch: last_plan reports path 0, not the previous call's plan)
        if (n_in == 0) return SDRHIP_OK;
        if (frame_mode) return fail(SDRHIP_EINVAL, "internal: frame mode needs log2decim >= 1");
        int norm = ss < 16 ? (int)(16 - ss) : 0;
        hipError_t e = launch_decimate_simple(0, fcpos, in, in_stride, out, out_stride, n_in, d->nstreams, norm, 0, c->stream);
        if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "decimate1 launch: %s", hipGetErrorString(e));
        return SDRHIP_OK;
    }
    const unsigned target = 16 - L; // Decimators.cpp:43-44, 132-133, 222-223 ...
    const unsigned trunk = ss < target ? 0 : ss - target;
    const unsigned norm = ss < target ? target - ss : 0;
    const size_t n_resize = n_in >> L; // out.resize(len / N)
    *sampleSize = ss + L - trunk;
    if (n_out) *n_out = n_resize;
    d->last = DecimPlanInfo(); // (set below when a cascade kernel is launched; filter-less and empty calls report path 0)
    if (n_resize == 0) return SDRHIP_OK; // (the reference's unsigned loop bound would wrap here)

    if (fcpos != SDRHIP_FC_CEN && L <= 2) {
        if (frame_mode) return fail(SDRHIP_EINVAL, "internal: frame mode unsupported for filter-less decimation");
        hipError_t e = launch_decimate_simple((int)L, fcpos, in, in_stride, out, out_stride, n_in, d->nstreams, (int)norm, (int)trunk, c->stream);
        if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "decimate%u launch: %s", 1u << L, hipGetErrorString(e));
        return SDRHIP_OK;
    }

    DecimArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.in_stride = in_stride; a.out_stride = out_stride;
    a.n_used = n_resize << L;
    a.state_cur = d->state[d->cur]; a.state_next = d->state[d->cur ^ 1];
    a.nstreams = d->nstreams;
    a.bias = d->bias; a.norm = (int)norm; a.trunk = (int)trunk;
    a.frame_mode = frame_mode; a.frame_blocks = frame_blocks; a.frame_sample_base = frame_sample_base;
    if (meta) {
        a.meta_first = meta->first; a.meta_count = meta->count; a.meta_frame_count0 = meta->frame_count0;
        memcpy(a.meta_w, meta->w, sizeof(a.meta_w));
        a.meta_idx0 = meta->idx0; a.meta_rate = meta->rate;
    }
    plan_decimate((int)L, fcpos, a.n_used, d->nstreams, &a.nsub_per_seg, &a.nseg);
    const bool cen = (fcpos == SDRHIP_FC_CEN);
    const bool pack16 = cen && d->stage0_int16;
    // matrix-core cascade for the centred modes when the call is long enough to fill the chip;
    // SDRHIP_DECIM_PATH = valu | mfma | auto (default), SDRHIP_MFMA_SPAN = span length in samples (tests)
    const CtxOptions &env = c->opt;
    bool use_mfma = false;
    // (frame mode on the matrix cores: the waves' store offsets are 32 bits with the top two reserved, decim_mfma.hip)
    const bool mfma_frames = env.rx_direct && out_stride * 4 < 0x3fffffffu;
    a.mf_ring = coresident ? 3 : env.mfma_ring; // (the planner reads it: ring 2 = two waves per SIMD)
    if ((!frame_mode || mfma_frames) && env.decim_path != DECIM_PATH_VALU && (env.decim_path == DECIM_PATH_MFMA || a.n_used * (size_t)d->nstreams >= mfma_min_samples(env, (int)L)))
        use_mfma = plan_decimate_mfma((int)L, fcpos, a.n_used, d->nstreams, env.mfma_span, c->n_cu, &a);
    a.mf_dump = c->decim_dump;
    a.mf_prio = coresident ? 1 : 0;
    hipError_t e;
    {
        KTimer kt(c, SDRHIP_K_DECIMATE);
        if (use_mfma && fuse && L <= 4) { // (decimate32 / 64: 180-220 VGPRs, no room for encoder waves beside them)
            if (++c->fused_tag == 0xffffffffu) { // (the per-CU words hold the highest tag seen: start over before it wraps)
                (void)hipMemsetAsync(c->fused_roles, 0, SDRHIP_FUSED_ROLE_WORDS * 4, c->stream);
                c->fused_tag = 1;
            }
            if (c->opt.rx_fused == 2) { // (experiment: the fused kernel without encoder units, the encoder as its own launch)
                Enc128Args none = *fuse;
                none.nlist = 0;
                e = launch_rx_fused((int)L, pack16, a, none, c->fused_roles, c->fused_tag, c->stream);
                if (e == hipSuccess) e = launch_gf_encode128(*fuse, c->stream);
            } else
            e = launch_rx_fused((int)L, pack16, a, *fuse, c->fused_roles, c->fused_tag, c->stream);
            if (e != hipSuccess) {
                // a fused launch that never started has not cleared the NEXT launch's unit counters (block 0 of every launch
                // does that): start the role table over, or the next fused launch would find them exhausted and do nothing
                (void)hipMemsetAsync(c->fused_roles, 0, SDRHIP_FUSED_ROLE_WORDS * 4, c->stream);
                c->fused_tag = 0;
            }
            if (fused) *fused = true;
        } else {
            e = use_mfma ? launch_decimate_mfma((int)L, pack16, a, c->stream) : launch_decimate((int)L, fcpos, pack16, a, c->stream);
        }
    }
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "decimate launch: %s", hipGetErrorString(e));
    d->last.path = use_mfma ? DECIM_PATH_MFMA : DECIM_PATH_VALU;
    d->last.span = use_mfma ? a.mf_span : 0; d->last.head = use_mfma ? a.mf_head : 0; d->last.tail_start = use_mfma ? a.mf_tail_start : 0;
    d->last.wps = use_mfma ? a.mf_wps : 0; d->last.npieces = use_mfma ? a.mf_npieces : 0; d->last.nseg = use_mfma ? 0 : a.nseg;
    d->cur ^= 1;
    // m_decimator2's 64-entry history now holds raw int16 samples (cen) or rotate-sums (inf/sup); a centred
    // call shorter than the history leaves older rotate-sums in it, which the packed-int16 first stage
    // cannot represent
    d->stage0_int16 = cen && (a.n_used >= (size_t)2 * DEC_HIST || d->stage0_int16);
    return SDRHIP_OK;
}
} // namespace sdrhip

namespace sdrhip {
// would a plain (stream-order) decimate call of this size run on the matrix cores?  (The Rx pipe then decimates into
// a stream-order buffer and frames it with K2 instead of using the VALU kernel's fused frame epilogue.)
bool decimate_mfma_applies(const sdrhip_decimators *d, int log2decim, int fcpos, size_t n_in)
{
    const CtxOptions &env = d->ctx->opt;
    const size_t n_used = (n_in >> log2decim) << log2decim;
    if (env.decim_path == DECIM_PATH_VALU || (env.decim_path != DECIM_PATH_MFMA && n_used * (size_t)d->nstreams < mfma_min_samples(env, log2decim))) return false;
    DecimArgs tmp;
    memset(&tmp, 0, sizeof(tmp));
    tmp.mf_ring = env.mfma_ring;
    return plan_decimate_mfma(log2decim, fcpos, n_used, d->nstreams, env.mfma_span, d->ctx->n_cu, &tmp);
}
} // namespace sdrhip

extern "C" int sdrhip_decimate(sdrhip_decimators *d, int log2decim, int fcpos, unsigned *sampleSize, const int16_t *iq_in,
                               size_t n_in, size_t in_stride, int16_t *iq_out, size_t out_stride, size_t *n_out, int mem)
{
    if (!d || !sampleSize) return fail(SDRHIP_EINVAL, "decimate: NULL handle or sampleSize");
    sdrhip::CtxLock lock_(d->ctx);
    if (log2decim < 0 || log2decim > 6) return fail(SDRHIP_EINVAL, "Invalid log2 decimation factor"); // Downsampler.cpp:39-43
    if (fcpos < SDRHIP_FC_INF || fcpos > SDRHIP_FC_CEN) return fail(SDRHIP_EINVAL, "Invalid Fc position index"); // :55-59
    if (*sampleSize < 1 || *sampleSize > 16) return fail(SDRHIP_EINVAL, "sampleSize must be 1..16");
    if (n_in && (!iq_in || (!iq_out && (n_in >> log2decim)))) return fail(SDRHIP_EINVAL, "decimate: NULL buffer"); // (no output, no buffer needed)
    sdrhip_ctx *c = d->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int S = d->nstreams;
    const size_t n_res = n_in >> log2decim;
    if (S == 1) { in_stride = n_in; out_stride = n_res; }
    if (S > 1 && (in_stride < n_in || out_stride < n_res)) return fail(SDRHIP_EINVAL, "decimate: stride smaller than the per-stream length");

    if (mem == SDRHIP_MEM_DEVICE) {
        if (n_in && (!aligned16(iq_in) || !aligned16(iq_out) || (S > 1 && ((in_stride & 3) || (out_stride & 3)))))
            return fail(SDRHIP_EALIGN, "decimate: device pointers must be 16-byte aligned and strides multiples of 4 samples");
        return decimate_device(d, log2decim, fcpos, sampleSize, iq_in, n_in, in_stride, iq_out, out_stride, n_out, 0, 0, 0);
    }
    if (mem != SDRHIP_MEM_HOST) return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");

    // host buffers: stage through device memory with padded (16-byte aligned) per-stream strides
    const size_t dis = (n_in + 3) & ~(size_t)3, dos = (n_res + 3) & ~(size_t)3;
    if (n_in == 0) // nothing to stage; sampleSize still advances like in the device path
        return decimate_device(d, log2decim, fcpos, sampleSize, nullptr, 0, 0, nullptr, 0, n_out, 0, 0, 0);
    int rc;
    if ((size_t)S * dis * 4 <= SDRHIP_ZEROCOPY_MAX) {
        // small call (one TestSource block ...): no copy engine, the kernel reads and writes pinned host memory itself
        if ((rc = c->zin.reserve((size_t)S * dis * 4 + 16))) return rc;
        if ((rc = c->zout.reserve((size_t)S * dos * 4 + 16))) return rc;
        for (int s = 0; s < S; ++s) memcpy(c->zin.as<int16_t>() + (size_t)s * dis * 2, iq_in + (size_t)s * in_stride * 2, n_in * 4);
        rc = decimate_device(d, log2decim, fcpos, sampleSize, c->zin.as<int16_t>(), n_in, dis, c->zout.as<int16_t>(), dos, n_out, 0, 0, 0);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int s = 0; s < S && n_res; ++s) memcpy(iq_out + (size_t)s * out_stride * 2, c->zout.as<int16_t>() + (size_t)s * dos * 2, n_res * 4);
        return SDRHIP_OK;
    }
    if ((rc = c->in.reserve((size_t)S * dis * 4 + 16))) return rc;
    if ((rc = c->out.reserve((size_t)S * dos * 4 + 16))) return rc;
    HIP_TRY(hipMemcpy2DAsync(c->in.p, dis * 4, iq_in, in_stride * 4, n_in * 4, S, hipMemcpyHostToDevice, c->stream));
    rc = decimate_device(d, log2decim, fcpos, sampleSize, static_cast<const int16_t *>(c->in.p), n_in, dis,
                         static_cast<int16_t *>(c->out.p), dos, n_out, 0, 0, 0);
    if (rc) return rc;
    if (n_res) HIP_TRY(hipMemcpy2DAsync(iq_out, out_stride * 4, c->out.p, dos * 4, n_res * 4, S, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDRHIP_OK;
}
