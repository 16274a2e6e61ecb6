// ref_shim.cpp -- C entry points around the REAL reference classes, so that the
// oracle restatement (sdr_oracle.c) can be diffed against the reference itself and
// so that bench.py can time the true reference CPU code ("cpu_baseline.kind":
// "reference").  TEST INFRASTRUCTURE ONLY.
//
// This file contains no reference code: it only #includes the reference's own
// headers (found via -I/root/reference/include at build time) and calls
// Decimators::decimate*_{inf,sup,cen} / Interpolators::interpolate*_cen the way
// Downsampler::process (Downsampler.cpp:74-162) and Upsampler::process
// (Upsampler.cpp:52-84) do.  oracle/Makefile compiles it together with
// /root/reference/sdmnbase/{Decimators,Interpolators,HBFilterTraits}.cpp where
// they lie; the result goes to oracle/_ref/ (git-ignored, shipped to the GPU box).
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "Decimators.h"
#include "Interpolators.h"

extern "C" {

// 0 = IntHalfbandFilterEO1 (built with -DUSE_SSE4_1), 1 = IntHalfbandFilterDB
int sdrref_bias(void)
{
#if defined(USE_SSE4_1)
    return 0;
#else
    return 1;
#endif
}

void *sdrref_decimators_new(void) { return new Decimators(); }
void sdrref_decimators_free(void *p) { delete static_cast<Decimators *>(p); }

} // extern "C"

// the dispatch of Downsampler::process (Downsampler.cpp:74-162), which itself cannot be
// compiled here (it needs boost::spirit through parsekv.h)
static void run_decimate(Decimators &d, int log2decim, int fcpos, unsigned int &ss, const IQSampleVector &in,
                         IQSampleVector &out)
{
    if (log2decim == 0) {
        out = in;
        Decimators::decimate1(ss, out);
    } else if (fcpos == 0) {
        switch (log2decim) {
        case 1: Decimators::decimate2_inf(ss, in, out); break;
        case 2: Decimators::decimate4_inf(ss, in, out); break;
        case 3: d.decimate8_inf(ss, in, out); break;
        case 4: d.decimate16_inf(ss, in, out); break;
        case 5: d.decimate32_inf(ss, in, out); break;
        case 6: d.decimate64_inf(ss, in, out); break;
        }
    } else if (fcpos == 1) {
        switch (log2decim) {
        case 1: Decimators::decimate2_sup(ss, in, out); break;
        case 2: Decimators::decimate4_sup(ss, in, out); break;
        case 3: d.decimate8_sup(ss, in, out); break;
        case 4: d.decimate16_sup(ss, in, out); break;
        case 5: d.decimate32_sup(ss, in, out); break;
        case 6: d.decimate64_sup(ss, in, out); break;
        }
    } else {
        switch (log2decim) {
        case 1: d.decimate2_cen(ss, in, out); break;
        case 2: d.decimate4_cen(ss, in, out); break;
        case 3: d.decimate8_cen(ss, in, out); break;
        case 4: d.decimate16_cen(ss, in, out); break;
        case 5: d.decimate32_cen(ss, in, out); break;
        case 6: d.decimate64_cen(ss, in, out); break;
        }
    }
}

static void run_interpolate(Interpolators &u, int log2interp, const IQSampleVector &in, IQSampleVector &out)
{
    switch (log2interp) { // Upsampler::process, Upsampler.cpp:52-84
    case 0: out = in; break;
    case 1: u.interpolate2_cen(in, out); break;
    case 2: u.interpolate4_cen(in, out); break;
    case 3: u.interpolate8_cen(in, out); break;
    case 4: u.interpolate16_cen(in, out); break;
    case 5: u.interpolate32_cen(in, out); break;
    case 6: u.interpolate64_cen(in, out); break;
    }
}

extern "C" {

// returns out.size() after the call; iq_out must hold n_in >> log2decim samples
size_t sdrref_decimate(void *p, int log2decim, int fcpos, unsigned *sampleSize, const int16_t *iq_in,
                       size_t n_in, int16_t *iq_out)
{
    IQSampleVector in(n_in), out;
    std::memcpy(in.data(), iq_in, n_in * sizeof(IQSample));
    unsigned int ss = *sampleSize;
    run_decimate(*static_cast<Decimators *>(p), log2decim, fcpos, ss, in, out);
    *sampleSize = ss;
    std::memcpy(iq_out, out.data(), out.size() * sizeof(IQSample));
    return out.size();
}

void *sdrref_interpolators_new(void) { return new Interpolators(); }
void sdrref_interpolators_free(void *p) { delete static_cast<Interpolators *>(p); }

size_t sdrref_interpolate(void *p, int log2interp, const int16_t *iq_in, size_t n_in, int16_t *iq_out)
{
    IQSampleVector in(n_in), out;
    std::memcpy(in.data(), iq_in, n_in * sizeof(IQSample));
    run_interpolate(*static_cast<Interpolators *>(p), log2interp, in, out);
    std::memcpy(iq_out, out.data(), out.size() * sizeof(IQSample));
    return out.size();
}

// Timing helpers for bench.py: `reps` calls on the same input vector inside one native
// call (no Python, no copies in the timed loop); return the last call's out.size().
size_t sdrref_decimate_repeat(void *p, int log2decim, int fcpos, unsigned sampleSize0, const int16_t *iq_in,
                              size_t n_in, int16_t *iq_out, int reps)
{
    IQSampleVector in(n_in), out;
    std::memcpy(in.data(), iq_in, n_in * sizeof(IQSample));
    for (int r = 0; r < reps; ++r) {
        unsigned int ss = sampleSize0;
        run_decimate(*static_cast<Decimators *>(p), log2decim, fcpos, ss, in, out);
    }
    std::memcpy(iq_out, out.data(), out.size() * sizeof(IQSample));
    return out.size();
}

size_t sdrref_interpolate_repeat(void *p, int log2interp, const int16_t *iq_in, size_t n_in, int16_t *iq_out,
                                 int reps)
{
    IQSampleVector in(n_in), out;
    std::memcpy(in.data(), iq_in, n_in * sizeof(IQSample));
    for (int r = 0; r < reps; ++r) run_interpolate(*static_cast<Interpolators *>(p), log2interp, in, out);
    std::memcpy(iq_out, out.data(), out.size() * sizeof(IQSample));
    return out.size();
}

} // extern "C"
