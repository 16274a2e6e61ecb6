// adapter_test.cpp -- exercises the drop-in C++ adapters the way Downsampler::process
// (Downsampler.cpp:74-162), Upsampler::process (Upsampler.cpp:52-84), UDPSinkFEC::transmitUDP
// (UDPSinkFEC.cpp:228-256) and SDRdaemonFECBuffer::writeAndRead (.cpp:148-213) use the
// reference classes.  Reads an int16 IQ file, writes the results; tests/test_gpu_adapters.py
// compares them with the oracle.
//   adapter_test in.bin dec.bin int.bin fec.bin
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "Decimators.h"
#include "Interpolators.h"
#include "cm256.h"
#include "Downsampler.h" // the optional boost-free Downsampler / Upsampler adapters
#include "Upsampler.h"

static std::vector<IQSample> read_iq(const char *path)
{
    std::vector<IQSample> v;
    FILE *f = std::fopen(path, "rb");
    if (!f) return v;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / 4);
    if (std::fread(v.data(), 4, v.size(), f) != v.size()) v.clear();
    std::fclose(f);
    return v;
}

static bool differ(const IQSampleVector &x, const IQSampleVector &y)
{
    return x.size() != y.size() || std::memcmp(x.data(), y.data(), x.size() * 4) != 0;
}

static void write_bin(const char *path, const void *p, size_t n)
{
    FILE *f = std::fopen(path, "wb");
    std::fwrite(p, 1, n, f);
    std::fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    IQSampleVector in = read_iq(argv[1]);
    // --- Rx side: two blocks through decimate16_cen, state carried by the object, then a mode switch
    Decimators dec;
    IQSampleVector a(in.begin(), in.begin() + in.size() / 2), b(in.begin() + in.size() / 2, in.end()), o1, o2, o3;
    unsigned int ss = 16;
    dec.decimate16_cen(ss, a, o1);
    ss = 16;
    dec.decimate16_cen(ss, b, o2);
    ss = 16;
    dec.decimate8_inf(ss, a, o3);
    unsigned int ss4 = 12;
    IQSampleVector o4;
    Decimators::decimate4_sup(ss4, a, o4);
    std::vector<IQSample> dall(o1);
    dall.insert(dall.end(), o2.begin(), o2.end());
    dall.insert(dall.end(), o3.begin(), o3.end());
    dall.insert(dall.end(), o4.begin(), o4.end());
    write_bin(argv[2], dall.data(), dall.size() * 4);
    std::printf("sampleSize %u %u\n", ss, ss4);
    // --- Tx side
    Interpolators itp;
    IQSampleVector u1, u2;
    IQSampleVector head(o1.begin(), o1.begin() + 1000), tail(o1.begin() + 1000, o1.end());
    itp.interpolate16_cen(head, u1);
    itp.interpolate16_cen(tail, u2);
    u1.insert(u1.end(), u2.begin(), u2.end());
    write_bin(argv[3], u1.data(), u1.size() * 4);
    // --- FEC: 128 blocks of 508 bytes cut from the input, 32 recovery blocks, lose 24, decode
    CM256 cm;
    if (!cm.isInitialized()) return 3;
    const int K = 128, R = 32, BB = 508;
    std::vector<unsigned char> orig((size_t)K * BB), rec((size_t)R * BB);
    std::memcpy(orig.data(), in.data(), orig.size());
    CM256::cm256_encoder_params p = {K, R, BB};
    std::vector<CM256::cm256_block> d(K);
    for (int i = 0; i < K; ++i) { d[i].Block = &orig[(size_t)i * BB]; d[i].Index = (unsigned char)i; }
    if (cm.cm256_encode(p, d.data(), rec.data())) return 4;
    std::vector<unsigned char> work(orig);
    int nrec = 0;
    for (int i = 0; i < K; ++i) {
        if (i % 5 == 1 && nrec < 24) { // erased: deliver recovery row nrec in its place, at the END like the wire order
            ++nrec;
        }
    }
    // survivors in index order, then the recovery blocks (SDRdaemonFECBuffer.cpp:148-163)
    std::vector<CM256::cm256_block> rxd;
    std::vector<unsigned char> recw(rec);
    int used = 0;
    for (int i = 0; i < K; ++i) {
        bool erased = (i % 5 == 1) && used < 24;
        if (erased) { ++used; std::memset(&work[(size_t)i * BB], 0, BB); continue; }
        CM256::cm256_block bk = {&work[(size_t)i * BB], (unsigned char)i};
        rxd.push_back(bk);
    }
    for (int r = 0; r < used; ++r) { CM256::cm256_block bk = {&recw[(size_t)r * BB], (unsigned char)(K + r)}; rxd.push_back(bk); }
    CM256::cm256_encoder_params pd = {K, used, BB};
    if (cm.cm256_decode(pd, rxd.data())) return 5;
    for (int ir = 0; ir < used; ++ir) { // the fix-up loop of SDRdaemonFECBuffer.cpp:208-213
        const CM256::cm256_block &bk = rxd[K - used + ir];
        std::memcpy(&work[(size_t)bk.Index * BB], bk.Block, BB);
    }
    std::printf("fec roundtrip %s\n", std::memcmp(work.data(), orig.data(), orig.size()) == 0 ? "OK" : "MISMATCH");
    write_bin(argv[4], rec.data(), rec.size());
    // --- Downsampler / Upsampler the way sdrdaemonrx.cpp:517,579-663 and sdrdaemontx.cpp:381,493 use them
    if (argc >= 6) {
        Downsampler dn(4, Downsampler::FC_POS_CENTER);
        IQSampleVector s1, s2, s3, s4;
        unsigned int s = 16;
        dn.process(s, a, s1);
        parsekv::pairs_type m;
        m["decim"] = "3"; m["fcpos"] = "0";
        if (!dn.configure(m) || dn.getLog2Decimation() != 3) return 6;
        s = 16;
        dn.process(s, b, s2);                       // decimate8_inf on the same filter bank
        m.clear(); m["decim"] = "7";
        if (dn.configure(m) || dn || dn.error() != "Invalid log2 decimation factor" || !dn) return 7; // error() clears
        m.clear(); m["decim"] = "0";
        if (!dn.configure(m)) return 8;
        s = 12;
        dn.process(s, a, s3);                       // copy + decimate1: 12-bit samples shifted to 16
        Upsampler up(2);
        up.process(s1, s4);
        m.clear(); m["interp"] = "9";
        if (up.configure(m) || up.error() != "Invalid log2 interpolation factor") return 9;
        std::vector<IQSample> all(s1);
        all.insert(all.end(), s2.begin(), s2.end());
        all.insert(all.end(), s3.begin(), s3.end());
        all.insert(all.end(), s4.begin(), s4.end());
        write_bin(argv[5], all.data(), all.size() * 4);
        std::printf("samplers OK %u\n", s);
    }
    // --- threads, the way the reference uses the classes: the main thread decimates (sdrdaemonrx.cpp:640) while
    // UDPSinkFEC's transmit thread encodes (UDPSinkFEC.cpp:246) and a third thread calls a static entry point
    {
        IQSampleVector ref16, ref2;
        unsigned int s = 16;
        { Decimators d0; d0.decimate16_cen(s, a, ref16); }
        s = 16;
        Decimators::decimate2_inf(s, a, ref2);
        std::vector<unsigned char> orig(128 * 508), rec_ref(32 * 508);
        std::memcpy(orig.data(), in.data(), orig.size());
        CM256::cm256_encoder_params params = {128, 32, 508};
        std::vector<CM256::cm256_block> blocks(128);
        for (int i = 0; i < 128; ++i) { blocks[i].Block = orig.data() + i * 508; blocks[i].Index = (unsigned char)i; }
        { CM256 c0; if (c0.cm256_encode(params, blocks.data(), rec_ref.data())) return 5; }
        int bad[3] = {0, 0, 0};
        std::thread t1([&] {
            for (int r = 0; r < 40; ++r) { Decimators d; IQSampleVector o; unsigned int q = 16; d.decimate16_cen(q, a, o); bad[0] += differ(o, ref16); }
        });
        std::thread t2([&] {
            CM256 cm;
            std::vector<unsigned char> rec(32 * 508);
            for (int r = 0; r < 40; ++r) { std::memset(rec.data(), 0, rec.size()); bad[1] += cm.cm256_encode(params, blocks.data(), rec.data()) != 0 || rec != rec_ref; }
        });
        std::thread t3([&] {
            for (int r = 0; r < 40; ++r) { IQSampleVector o; unsigned int q = 16; Decimators::decimate2_inf(q, a, o); bad[2] += differ(o, ref2); }
        });
        for (int r = 0; r < 40; ++r) { IQSampleVector o; unsigned int q = 16; Decimators::decimate2_inf(q, a, o); bad[2] += differ(o, ref2); }
        t1.join(); t2.join(); t3.join();
        if (bad[0] || bad[1] || bad[2]) { std::printf("threads FAILED %d %d %d\n", bad[0], bad[1], bad[2]); return 6; }
        std::printf("threads OK\n");
    }
    return 0;
}
