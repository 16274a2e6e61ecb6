// Downsampler.h -- optional drop-in replacement of the reference's Downsampler (Downsampler.h:26-83,
// Downsampler.cpp:25-162) for builds without boost: the reference's own Downsampler.cpp compiles unchanged
// on top of the Decimators adapter, this header only spares it parsekv.h (boost::spirit).  Same
// constructor, configure() keys (decim, fcpos), dispatch and error() / operator bool contract.
#ifndef SDRHIP_DOWNSAMPLER_ADAPTER_H
#define SDRHIP_DOWNSAMPLER_ADAPTER_H

#include <cstdlib>
#include <iostream>
#include <map>
#include <string>

#include "Decimators.h"

namespace parsekv {
typedef std::map<std::string, std::string> pairs_type; // parsekv.h:31 (an identical typedef may be repeated)
}

class Downsampler
{
public:
    /** Center frequency relative position when downsampling (Downsampler.h:30-34) */
    typedef enum { FC_POS_INFRA = 0, FC_POS_SUPRA, FC_POS_CENTER } fcPos_t;

    Downsampler(unsigned int decim = 0, fcPos_t fcPos = FC_POS_CENTER) : m_decim(decim), m_fcPos(fcPos) {}
    ~Downsampler() {}

    /** Configure dynamically: keys decim (log2, 0..6) and fcpos (0..2), Downsampler.cpp:32-67 */
    bool configure(parsekv::pairs_type &m)
    {
        if (m.find("decim") != m.end()) {
            std::cerr << "Downsampler::configure: decim: " << m["decim"] << std::endl;
            const int log2Decim = std::atoi(m["decim"].c_str());
            if (log2Decim < 0 || log2Decim > 6) { m_error = "Invalid log2 decimation factor"; return false; }
            m_decim = (unsigned int)log2Decim;
        }
        if (m.find("fcpos") != m.end()) {
            std::cerr << "Downsampler::configure: fcpos: " << m["fcpos"] << std::endl;
            const int fcPosIndex = std::atoi(m["fcpos"].c_str());
            if (fcPosIndex < (int)FC_POS_INFRA || fcPosIndex > (int)FC_POS_CENTER) { m_error = "Invalid Fc position index"; return false; }
            m_fcPos = (fcPos_t)fcPosIndex;
        }
        return true;
    }

    unsigned int getLog2Decimation() const { return m_decim; }

    /** Downsampler::process (Downsampler.cpp:74-162): decim 0 = copy + decimate1, else decimate<2^decim>_<fcpos> */
    void process(unsigned int &sampleSize, const IQSampleVector &samples_in, IQSampleVector &samples_out)
    {
        if (m_decim == 0) {
            samples_out = samples_in;
            Decimators::decimate1(sampleSize, samples_out);
            return;
        }
        typedef void (Decimators::*method)(unsigned int &, const IQSampleVector &, IQSampleVector &);
        static const method cen[7] = {0, &Decimators::decimate2_cen, &Decimators::decimate4_cen, &Decimators::decimate8_cen,
                                      &Decimators::decimate16_cen, &Decimators::decimate32_cen, &Decimators::decimate64_cen};
        static const method inf[7] = {0, 0, 0, &Decimators::decimate8_inf, &Decimators::decimate16_inf, &Decimators::decimate32_inf,
                                      &Decimators::decimate64_inf};
        static const method sup[7] = {0, 0, 0, &Decimators::decimate8_sup, &Decimators::decimate16_sup, &Decimators::decimate32_sup,
                                      &Decimators::decimate64_sup};
        if (m_decim > 6) return;
        if (m_fcPos == FC_POS_CENTER) {
            (m_decimators.*cen[m_decim])(sampleSize, samples_in, samples_out);
        } else if (m_decim <= 2) { // the filter-less entry points are static (Decimators.h:35-39)
            if (m_decim == 1) {
                if (m_fcPos == FC_POS_INFRA) Decimators::decimate2_inf(sampleSize, samples_in, samples_out);
                else Decimators::decimate2_sup(sampleSize, samples_in, samples_out);
            } else {
                if (m_fcPos == FC_POS_INFRA) Decimators::decimate4_inf(sampleSize, samples_in, samples_out);
                else Decimators::decimate4_sup(sampleSize, samples_in, samples_out);
            }
        } else {
            (m_decimators.*(m_fcPos == FC_POS_INFRA ? inf : sup)[m_decim])(sampleSize, samples_in, samples_out);
        }
    }

    /** Rescale: the decimation-less alternative to process (Downsampler.cpp:69-72) */
    void rescale(unsigned int &sampleSize, IQSampleVector &samples_inout) { Decimators::decimate1(sampleSize, samples_inout); }

    operator bool() const { return m_error.empty(); }

    std::string error()
    {
        std::string ret(m_error);
        m_error.clear();
        return ret;
    }

private:
    unsigned int m_decim;
    fcPos_t m_fcPos;
    Decimators m_decimators;
    std::string m_error;
};

#endif
