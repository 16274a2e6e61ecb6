// sdrhip_adapter_common.h -- shared by the drop-in C++11 adapter headers (Decimators.h,
// Interpolators.h, cm256.h).  Header-only; link with -lsdrhip.
//
// One process-wide sdrhip context (device from $SDRHIP_DEVICE, default 0) is created on first
// use.  The adapters use host pointers (SDRHIP_MEM_HOST): the library stages the vectors
// through the GPU and returns when the result is back, which is the reference's synchronous
// call contract.  Errors: the reference's DSP methods cannot fail, so an sdrhip failure
// (no GPU, out of memory) throws std::runtime_error with sdrhip_last_error().
#ifndef SDRHIP_ADAPTER_COMMON_H
#define SDRHIP_ADAPTER_COMMON_H

#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "sdrhip.h"

#if defined(__has_include)
#if __has_include("SDRDaemon.h")
#include "SDRDaemon.h" // the reference's IQSample / IQSampleVector (SDRDaemon.h:52-70)
#define SDRHIP_HAVE_SDRDAEMON_H 1
#endif
#endif
#ifndef SDRHIP_HAVE_SDRDAEMON_H
// stand-alone use of the adapters (tests): a layout-compatible interleaved int16 IQ sample
#pragma pack(push, 1)
struct IQSample {
    IQSample() : m_real(0), m_imag(0) {}
    IQSample(std::int16_t re, std::int16_t im) : m_real(re), m_imag(im) {}
    std::int16_t real() const { return m_real; }
    std::int16_t imag() const { return m_imag; }
    void setReal(std::int16_t v) { m_real = v; }
    void setImag(std::int16_t v) { m_imag = v; }
    std::int16_t m_real, m_imag;
};
#pragma pack(pop)
typedef std::vector<IQSample> IQSampleVector;
#endif

static_assert(sizeof(IQSample) == 4, "IQSample must be two packed int16");

namespace sdrhip_adapter {

inline void check(int rc, const char *what)
{
    if (rc != SDRHIP_OK) throw std::runtime_error(std::string(what) + ": " + sdrhip_last_error());
}

inline sdrhip_ctx *new_context()
{
    sdrhip_ctx *ctx = nullptr;
    const char *dev = std::getenv("SDRHIP_DEVICE");
    check(sdrhip_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &ctx), "sdrhip_ctx_create");
    return ctx;
}

// The process-wide context of the DSP adapters (Decimators, Interpolators): created once (C++11 guarantees that the
// initialisation of a function-local static is thread-safe), lives for the process like the reference's static filter
// tables.  The library serialises the calls made on one context, so objects that are driven from ANOTHER thread
// (CM256 inside the reference's transmit / receive threads) take a context of their own instead: new_context().
inline sdrhip_ctx *context()
{
    static sdrhip_ctx *ctx = new_context();
    return ctx;
}

inline int hb_variant()
{
    // the reference picks IntHalfbandFilterEO1 when built with USE_SSE4_1, else DB (Decimators.h:24-28);
    // $SDRHIP_HB_VARIANT=DB selects the DB rounding at run time
    const char *v = std::getenv("SDRHIP_HB_VARIANT");
#if defined(USE_SSE4_1)
    const int dflt = SDRHIP_HB_EO1;
#else
    const int dflt = SDRHIP_HB_DB;
#endif
    if (!v) return dflt;
    return (v[0] == 'D' || v[0] == 'd') ? SDRHIP_HB_DB : SDRHIP_HB_EO1;
}

} // namespace sdrhip_adapter
#endif
