// Decimators.h -- drop-in replacement of the reference's include/Decimators.h +
// sdmnbase/Decimators.cpp (+ IntHalfbandFilter*.h, HBFilterTraits.cpp): the same class name and
// the same 19 method signatures (Decimators.h:35-53), computed by libsdrhip.so on the GPU.
// Downsampler.{h,cpp} (Downsampler.h:26-83) compile against it unchanged.
#ifndef INCLUDE_DECIMATORS_H_
#define INCLUDE_DECIMATORS_H_

#include <mutex>

#include "sdrhip_adapter_common.h"

class Decimators
{
public:
    Decimators() : m_h(nullptr) {}
    ~Decimators() { if (m_h) sdrhip_decimators_destroy(m_h); }
    Decimators(const Decimators&) = delete;
    Decimators& operator=(const Decimators&) = delete;

    static void decimate1(unsigned int& sampleSize, IQSampleVector& inout) { staticRun(0, 2, sampleSize, inout, inout); }
    static void decimate2_inf(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { staticRun(1, 0, sampleSize, in, out); }
    static void decimate2_sup(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { staticRun(1, 1, sampleSize, in, out); }
    void decimate2_cen(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(1, 2, sampleSize, in, out); }
    static void decimate4_inf(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { staticRun(2, 0, sampleSize, in, out); }
    static void decimate4_sup(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { staticRun(2, 1, sampleSize, in, out); }
    void decimate4_cen(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(2, 2, sampleSize, in, out); }
    void decimate8_inf(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(3, 0, sampleSize, in, out); }
    void decimate8_sup(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(3, 1, sampleSize, in, out); }
    void decimate8_cen(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(3, 2, sampleSize, in, out); }
    void decimate16_inf(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(4, 0, sampleSize, in, out); }
    void decimate16_sup(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(4, 1, sampleSize, in, out); }
    void decimate16_cen(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(4, 2, sampleSize, in, out); }
    void decimate32_inf(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(5, 0, sampleSize, in, out); }
    void decimate32_sup(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(5, 1, sampleSize, in, out); }
    void decimate32_cen(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(5, 2, sampleSize, in, out); }
    void decimate64_inf(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(6, 0, sampleSize, in, out); }
    void decimate64_sup(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(6, 1, sampleSize, in, out); }
    void decimate64_cen(unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { run(6, 2, sampleSize, in, out); }

private:
    sdrhip_decimators *m_h; // the six filter states m_decimator2..64 live behind this handle

    sdrhip_decimators *handle()
    {
        if (!m_h) sdrhip_adapter::check(sdrhip_decimators_create(sdrhip_adapter::context(), 1, sdrhip_adapter::hb_variant(), &m_h), "sdrhip_decimators_create");
        return m_h;
    }
    static void call(sdrhip_decimators *h, int log2, int fcpos, unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out)
    {
        const std::size_t n = in.size();
        if (&out != &in) out.resize(n >> log2); // out.resize(len / N), Decimators.cpp:41,130,176 ...
        std::size_t n_out = 0;
        unsigned ss = sampleSize;
        sdrhip_adapter::check(sdrhip_decimate(h, log2, fcpos, &ss, reinterpret_cast<const std::int16_t *>(in.data()), n, n,
                                              reinterpret_cast<std::int16_t *>(out.data()), n >> log2, &n_out, SDRHIP_MEM_HOST),
                              "sdrhip_decimate");
        sampleSize = ss;
    }
    void run(int log2, int fcpos, unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out) { call(handle(), log2, fcpos, sampleSize, in, out); }
    static sdrhip_decimators *staticHandle()
    {
        sdrhip_decimators *h = nullptr;
        sdrhip_adapter::check(sdrhip_decimators_create(sdrhip_adapter::context(), 1, sdrhip_adapter::hb_variant(), &h), "sdrhip_decimators_create");
        return h;
    }
    static void staticRun(int log2, int fcpos, unsigned int& sampleSize, const IQSampleVector& in, IQSampleVector& out)
    {
        // the filter-less entry points are static in the reference (Decimators.h:35-39) and callable from any thread:
        // one shared handle serves them (created once: thread-safe static initialisation), a mutex makes the calls on
        // it single-threaded as the library asks of a handle (they carry no filter state)
        static sdrhip_decimators *h = staticHandle();
        static std::mutex mtx;
        std::lock_guard<std::mutex> guard(mtx);
        call(h, log2, fcpos, sampleSize, in, out);
    }
};

#endif /* INCLUDE_DECIMATORS_H_ */
