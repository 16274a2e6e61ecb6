// sdrhip_testsource.cpp -- host side of the TestSource bank (include/sdrhip.h): the configuration semantics of the
// reference's TestSource (TestSource.cpp:59-258: keys srate, freq, dfp, dfn, power, blklen, fcpos, decim; the same
// range checks and error strings), one generator per stream, samples produced on the device (testsource_kernels.hip).
#include "sdrhip_host.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace sdrhip;

namespace {

// 2^30 cos(2 pi i / 4096) by a 31-step integer CORDIC: integer-only, the same on every machine.
// atan(2^-k) in units of 2^-32 turn, and the CORDIC gain 2^30 * prod 1/sqrt(1 + 2^-2k).
const int NCO_ATAN[31] = {536870912, 316933406, 167458907, 85004756, 42667331, 21354465, 10679838, 5340245, 2670163, 1335087, 667544,
                          333772, 166886, 83443, 41722, 20861, 10430, 5215, 2608, 1304, 652, 326, 163, 81, 41, 20, 10, 5, 3, 1, 1};
const int NCO_X0 = 652032874;

void nco_table(int *T)
{
    for (int i = 0; i < 4096; ++i) {
        long long z = (long long)i << 20; // phase, 2^32 per turn
        if (z >= 0x80000000LL) z -= 0x100000000LL;
        bool neg = false;
        if (z > 0x40000000LL) { z -= 0x80000000LL; neg = true; }
        else if (z < -0x40000000LL) { z += 0x80000000LL; neg = true; }
        long long x = NCO_X0, y = 0;
        for (int k = 0; k < 31; ++k) {
            const long long xs = x >> k, ys = y >> k;
            if (z >= 0) { x -= ys; y += xs; z -= NCO_ATAN[k]; }
            else { x += ys; y -= xs; z += NCO_ATAN[k]; }
        }
        T[i] = (int)(neg ? -x : x);
    }
}

// amplitude 10^(-dB / 20) in Q15, integer-only: dB steps of 10^(-1/20) in Q30, rounded
int amp_q15_from_db(int db)
{
    long long a = 1LL << 30;
    for (int i = 0; i < db && a > 0; ++i) a = (a * 956973408LL + (1LL << 29)) >> 30;
    return (int)((a + (1LL << 14)) >> 15);
}

// phase increment of a carrier offset: round(2^32 * df / srate)
unsigned phase_inc(long long df, long long srate)
{
    const long long num = df * 4294967296LL;
    const long long q = num >= 0 ? (num + srate / 2) / srate : -((-num + srate / 2) / srate);
    return (unsigned)(unsigned long long)q;
}

struct Gen {
    unsigned srate = 64000, freq = 435000000, conf_freq = 435000000; // TestSource.cpp:40-50
    int carrier_offset = 10000;  // m_carrierOffset (only ever the constructor's value: see configure)
    unsigned inc = phase_inc(10000, 64000);
    int amp = 3277;              // m_amplitude = 0.1 -> round(0.1 * 32768)
    int block_length = 65536;    // TestSource.h:33
    int fcpos = 2, decim = 0;
    unsigned phase = 0;
};

// key[=value] pairs separated by ',' or '&' (the grammar of parsekv.h:40-43)
std::map<std::string, std::string> parse_kv(const char *s)
{
    std::map<std::string, std::string> m;
    std::string str(s ? s : "");
    size_t i = 0;
    while (i <= str.size()) {
        size_t j = str.find_first_of(",&", i);
        if (j == std::string::npos) j = str.size();
        std::string item = str.substr(i, j - i);
        if (!item.empty()) {
            size_t e = item.find('=');
            if (e == std::string::npos) m[item] = "";
            else m[item.substr(0, e)] = item.substr(e + 1);
        }
        i = j + 1;
    }
    return m;
}

} // namespace

struct sdrhip_testsource {
    sdrhip_ctx *ctx;
    int nstreams;
    std::vector<Gen> gen;
    int *table = nullptr;              // device: 4096 x int32
    TestSourceParams *par = nullptr;   // device: [nstreams]
    PinnedBuf pin;
};

extern "C" int sdrhip_testsource_create(sdrhip_ctx *ctx, int nstreams, sdrhip_testsource **out)
{
    if (!ctx || !out || nstreams <= 0 || nstreams > 65535) return fail(SDRHIP_EINVAL, "testsource_create: bad argument");
    CtxLock lock_(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    sdrhip_testsource *t = new (std::nothrow) sdrhip_testsource();
    if (!t) return fail(SDRHIP_ENOMEM, "out of host memory");
    t->ctx = ctx; t->nstreams = nstreams; t->gen.resize((size_t)nstreams);
    std::vector<int> T(4096);
    nco_table(T.data());
    if (hipMalloc(reinterpret_cast<void **>(&t->table), 4096 * sizeof(int)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&t->par), (size_t)nstreams * sizeof(TestSourceParams)) != hipSuccess ||
        hipMemcpy(t->table, T.data(), 4096 * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        if (t->table) (void)hipFree(t->table);
        if (t->par) (void)hipFree(t->par);
        delete t;
        return fail(SDRHIP_ENOMEM, "testsource tables");
    }
    ctx_retain(ctx);
    *out = t;
    return SDRHIP_OK;
}

extern "C" void sdrhip_testsource_destroy(sdrhip_testsource *t)
{
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    t->pin.release();
    (void)hipFree(t->table);
    (void)hipFree(t->par);
    ctx_release(t->ctx);
    delete t;
}

// TestSource::configure(parsekv::pairs_type&) (TestSource.cpp:59-215) on stream `stream` (-1 = every stream).
// Mirrored on purpose: a message without `power` leaves the amplitude alone (the local default 1.0 is never applied,
// :65 and :153-167); `dfp` / `dfn` change the phase step but m_carrierOffset keeps the constructor's 10 kHz (the
// parsed value shadows it, :110,:130), so a later srate-only message falls back to a 10 kHz offset (:81).
extern "C" int sdrhip_testsource_configure(sdrhip_testsource *t, int stream, const char *kv)
{
    if (!t) return fail(SDRHIP_EINVAL, "testsource is NULL");
    if (stream < -1 || stream >= t->nstreams) return fail(SDRHIP_EINVAL, "testsource_configure: stream out of range");
    CtxLock lock_(t->ctx);
    std::map<std::string, std::string> m = parse_kv(kv);
    // all addressed streams are validated before any of them changes (a range that depends on a stream's own sample rate must
    // not leave the bank half updated); the one thing the reference commits before it can fail is m_fcPos (TestSource.cpp:175-186
    // stores it, the decim check comes after, :188-197): a valid fcpos survives an invalid decim, here as well
    const int s_begin = stream < 0 ? 0 : stream, s_end = stream < 0 ? t->nstreams : stream + 1;
    std::vector<Gen> fresh((size_t)(s_end - s_begin));
    for (int s = s_begin; s < s_end; ++s) {
        Gen g = t->gen[(size_t)s];
        unsigned sample_rate = g.srate, frequency = g.conf_freq;
        bool ch_srate = false, ch_freq = false, ch_phase = false, dfp = false;
        unsigned inc = g.inc;
        if (m.count("srate")) {
            sample_rate = (unsigned)atoi(m["srate"].c_str());
            if (sample_rate < 8000 || sample_rate > 10000000) return fail(SDRHIP_EINVAL, "Invalid sample rate");
            inc = phase_inc(g.carrier_offset, sample_rate);
            ch_srate = ch_phase = true;
            if (g.fcpos != 2) ch_freq = true;
        }
        if (m.count("freq")) {
            frequency = (unsigned)atoi(m["freq"].c_str());
            if (frequency < 10000) return fail(SDRHIP_EINVAL, "Invalid frequency");
            ch_freq = true;
        }
        if (m.count("dfp")) {
            const int off = atoi(m["dfp"].c_str());
            if (off > (int)sample_rate / 2 || off < 0) return fail(SDRHIP_EINVAL, "Invalid positive carrier offset");
            inc = phase_inc(off, sample_rate);
            dfp = true; ch_phase = true;
        }
        if (m.count("dfn") && !dfp) {
            const int off = atoi(m["dfn"].c_str());
            if (off > (int)sample_rate / 2 || off < 0) return fail(SDRHIP_EINVAL, "Invalid negative carrier offset");
            inc = phase_inc(-(long long)off, sample_rate);
            ch_phase = true;
        }
        if (m.count("power")) {
            const int dbn = atoi(m["power"].c_str());
            if (dbn < 0) return fail(SDRHIP_EINVAL, "Invalid peak power");
            g.amp = amp_q15_from_db(dbn); // db2A(-dbn), util.h:59-62
        }
        if (m.count("blklen")) {
            const int bl = atoi(m["blklen"].c_str());
            g.block_length = bl < 4096 ? 4096 : (bl > 1024 * 1024 ? 1024 * 1024 : bl); // :246-251
        }
        if (m.count("fcpos")) {
            const int fc = atoi(m["fcpos"].c_str());
            if (fc < 0 || fc > 2) return fail(SDRHIP_EINVAL, "Invalid center frequency position");
            g.fcpos = fc;
            ch_freq = true;
        }
        if (m.count("decim")) {
            const int d = atoi(m["decim"].c_str());
            if (d < 0 || d > 6) {
                if (m.count("fcpos")) // (validated above, the same for every stream)
                    for (int u = s_begin; u < s_end; ++u) t->gen[(size_t)u].fcpos = g.fcpos;
                return fail(SDRHIP_EINVAL, "Invalid log2 decimation factor");
            }
            g.decim = d;
        }
        g.conf_freq = frequency;
        // "Intentionally tune at a higher frequency to avoid DC offset" (:199-209)
        double tuner = frequency;
        if (g.fcpos == 0) tuner = frequency + 0.25 * sample_rate;
        else if (g.fcpos == 1) tuner = frequency - 0.25 * sample_rate;
        if (ch_srate) g.srate = sample_rate;
        if (ch_freq) g.freq = (unsigned)tuner;
        if (ch_phase) g.inc = inc;
        fresh[(size_t)(s - s_begin)] = g;
    }
    for (int s = s_begin; s < s_end; ++s) t->gen[(size_t)s] = fresh[(size_t)(s - s_begin)];
    return SDRHIP_OK;
}

extern "C" int sdrhip_testsource_get(const sdrhip_testsource *t, int stream, uint32_t *sample_rate, uint32_t *frequency, int *block_length,
                                     int *log2decim, int *fcpos)
{
    if (!t || stream < 0 || stream >= t->nstreams) return fail(SDRHIP_EINVAL, "testsource_get: bad argument");
    const Gen &g = t->gen[(size_t)stream];
    if (sample_rate) *sample_rate = g.srate;   // get_sample_rate(), :261
    if (frequency) *frequency = g.freq;        // get_frequency(), :267
    if (block_length) *block_length = g.block_length;
    if (log2decim) *log2decim = g.decim;
    if (fcpos) *fcpos = g.fcpos;
    return SDRHIP_OK;
}

// The next n samples of every stream (TestSource::get_samples / read_samples, :335-422, without the real-time pacing):
// stream s at iq_out + 2 * s * out_stride.
extern "C" int sdrhip_testsource_read(sdrhip_testsource *t, int16_t *iq_out, size_t n, size_t out_stride, int mem)
{
    if (!t) return fail(SDRHIP_EINVAL, "testsource is NULL");
    CtxLock lock_(t->ctx);
    if (n == 0) return SDRHIP_OK;
    if (!iq_out) return fail(SDRHIP_EINVAL, "testsource_read: NULL buffer");
    sdrhip_ctx *c = t->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int S = t->nstreams;
    if (S == 1) out_stride = n;
    if (out_stride < n) return fail(SDRHIP_EINVAL, "testsource_read: stride smaller than n");
    int rc;
    if ((rc = t->pin.reserve((size_t)S * sizeof(TestSourceParams)))) return rc;
    TestSourceParams *hp = t->pin.as<TestSourceParams>();
    for (int s = 0; s < S; ++s) {
        Gen &g = t->gen[(size_t)s];
        hp[s].phase0 = g.phase; hp[s].inc = g.inc; hp[s].amp = g.amp;
        g.phase += (unsigned)n * g.inc; // mod 2^32
    }
    HIP_TRY(hipMemcpyAsync(t->par, hp, (size_t)S * sizeof(TestSourceParams), hipMemcpyHostToDevice, c->stream));
    t->pin.mark(c->stream);
    int16_t *dout = iq_out;
    size_t dstride = out_stride;
    if (mem == SDRHIP_MEM_HOST) {
        dstride = (n + 3) & ~(size_t)3;
        if ((rc = c->out.reserve((size_t)S * dstride * 4 + 16))) return rc;
        dout = c->out.as<int16_t>();
    } else if (mem == SDRHIP_MEM_DEVICE) {
        if (!aligned16(iq_out) || (S > 1 && (out_stride & 3))) return fail(SDRHIP_EALIGN, "testsource_read: device output must be 16-byte aligned");
    } else {
        return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    }
    hipError_t e = launch_testsource(t->table, t->par, dout, dstride, n, S, c->stream);
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "testsource launch: %s", hipGetErrorString(e));
    if (mem == SDRHIP_MEM_HOST) {
        HIP_TRY(hipMemcpy2DAsync(iq_out, out_stride * 4, dout, dstride * 4, n * 4, S, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SDRHIP_OK;
}
