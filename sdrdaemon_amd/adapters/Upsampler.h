// Upsampler.h -- optional drop-in replacement of the reference's Upsampler (Upsampler.h:36-50,
// Upsampler.cpp:25-84) for builds without boost, the twin of Downsampler.h in this directory: same
// constructor, configure() key (interp), dispatch and error() contract on top of the Interpolators adapter.
#ifndef SDRHIP_UPSAMPLER_ADAPTER_H
#define SDRHIP_UPSAMPLER_ADAPTER_H

#include <cstdlib>
#include <iostream>
#include <map>
#include <string>

#include "Interpolators.h"

namespace parsekv {
typedef std::map<std::string, std::string> pairs_type; // parsekv.h:31
}

class Upsampler
{
public:
    Upsampler(unsigned int interp = 0) : m_interp(interp) {}
    ~Upsampler() {}

    /** Configure dynamically: key interp (log2, 0..6), Upsampler.cpp:31-49 */
    bool configure(parsekv::pairs_type &m)
    {
        if (m.find("interp") != m.end()) {
            std::cerr << "Upsampler::configure: interp: " << m["interp"] << std::endl;
            const int log2Interp = std::atoi(m["interp"].c_str());
            if (log2Interp < 0 || log2Interp > 6) { m_error = "Invalid log2 interpolation factor"; return false; }
            m_interp = (unsigned int)log2Interp;
        }
        return true;
    }

    unsigned int getLog2Interpolation() const { return m_interp; }

    /** Upsampler::process (Upsampler.cpp:51-84): interp 0 = copy, else interpolate<2^interp>_cen */
    void process(const IQSampleVector &samples_in, IQSampleVector &samples_out)
    {
        switch (m_interp) {
        case 0: samples_out = samples_in; break;
        case 1: m_interpolators.interpolate2_cen(samples_in, samples_out); break;
        case 2: m_interpolators.interpolate4_cen(samples_in, samples_out); break;
        case 3: m_interpolators.interpolate8_cen(samples_in, samples_out); break;
        case 4: m_interpolators.interpolate16_cen(samples_in, samples_out); break;
        case 5: m_interpolators.interpolate32_cen(samples_in, samples_out); break;
        case 6: m_interpolators.interpolate64_cen(samples_in, samples_out); break;
        default: break;
        }
    }

    operator bool() const { return m_error.empty(); }

    std::string error()
    {
        std::string ret(m_error);
        m_error.clear();
        return ret;
    }

private:
    unsigned int m_interp;
    Interpolators m_interpolators;
    std::string m_error;
};

#endif
