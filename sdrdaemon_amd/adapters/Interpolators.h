// Interpolators.h -- drop-in replacement of the reference's include/Interpolators.h +
// sdmnbase/Interpolators.cpp: same class, same six signatures (Interpolators.h:38-43), computed
// by libsdrhip.so.  Upsampler.{h,cpp} (Upsampler.h:27-70) compile against it unchanged.
#ifndef INCLUDE_INTERPOLATORS_H_
#define INCLUDE_INTERPOLATORS_H_

#include "sdrhip_adapter_common.h"

class Interpolators
{
public:
    Interpolators() : m_h(nullptr) {}
    ~Interpolators() { if (m_h) sdrhip_interpolators_destroy(m_h); }
    Interpolators(const Interpolators&) = delete;
    Interpolators& operator=(const Interpolators&) = delete;

    void interpolate2_cen(const IQSampleVector& in, IQSampleVector& out) { run(1, in, out); }
    void interpolate4_cen(const IQSampleVector& in, IQSampleVector& out) { run(2, in, out); }
    void interpolate8_cen(const IQSampleVector& in, IQSampleVector& out) { run(3, in, out); }
    void interpolate16_cen(const IQSampleVector& in, IQSampleVector& out) { run(4, in, out); }
    void interpolate32_cen(const IQSampleVector& in, IQSampleVector& out) { run(5, in, out); }
    void interpolate64_cen(const IQSampleVector& in, IQSampleVector& out) { run(6, in, out); }

private:
    sdrhip_interpolators *m_h; // m_interpolator2..64 live behind this handle

    void run(int log2, const IQSampleVector& in, IQSampleVector& out)
    {
        if (!m_h) sdrhip_adapter::check(sdrhip_interpolators_create(sdrhip_adapter::context(), 1, &m_h), "sdrhip_interpolators_create");
        const std::size_t n = in.size();
        out.resize(n << log2); // out.resize(len * N), Interpolators.cpp:26,50,83,133 ...
        std::size_t n_out = 0;
        sdrhip_adapter::check(sdrhip_interpolate(m_h, log2, reinterpret_cast<const std::int16_t *>(in.data()), n, n,
                                                 reinterpret_cast<std::int16_t *>(out.data()), n << log2, &n_out, SDRHIP_MEM_HOST),
                              "sdrhip_interpolate");
    }
};

#endif /* INCLUDE_INTERPOLATORS_H_ */
