// udp_adapter_test.cpp -- drives the UDPSinkFEC / UDPSourceFEC drop-in adapters the way the reference's
// main loops do (sdrdaemonrx.cpp:480-490,640-655: construct through the base class, set the stream
// parameters, write() every block; sdrdaemontx.cpp:381,449-470: read() one frame at a time,
// getStatusMessage()).  tests/test_gpu_udp_adapters.py is the UDP peer.
//   udp_adapter_test tx <port> <nb_fec> <txdelay_us> <in.bin> <chunk_samples>
//   udp_adapter_test rx <port> <nframes> <out.bin>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "UDPSinkFEC.h"
#include "UDPSourceFEC.h"

static IQSampleVector read_iq(const char *path)
{
    IQSampleVector v;
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

int main(int argc, char **argv)
{
    if (argc >= 7 && !std::strcmp(argv[1], "tx")) {
        const unsigned port = (unsigned)std::atoi(argv[2]);
        std::unique_ptr<UDPSink> sink(new UDPSinkFEC("127.0.0.1", port));
        if (!(*sink)) { std::fprintf(stderr, "sink: %s\n", sink->error().c_str()); return 1; }
        sink->setCenterFrequency(435000000ull); // Hz
        sink->setSampleRate(625000);
        sink->setSampleBytes(2);
        sink->setSampleBits(16);
        sink->setNbBlocksFEC(std::atoi(argv[3]));
        sink->setTxDelay(std::atoi(argv[4]));
        IQSampleVector in = read_iq(argv[5]);
        const size_t chunk = (size_t)std::atol(argv[6]);
        for (size_t pos = 0; pos < in.size(); pos += chunk) {
            const size_t n = in.size() - pos < chunk ? in.size() - pos : chunk;
            IQSampleVector blk(in.begin() + pos, in.begin() + pos + n);
            sink->write(blk);
            if (!(*sink)) { std::fprintf(stderr, "write: %s\n", sink->error().c_str()); return 1; }
        }
        usleep(1500000); // let the transmit thread drain its ring (the destructor drops what is queued, like the reference)
        std::printf("tx done\n");
        return 0;
    }
    if (argc >= 5 && !std::strcmp(argv[1], "rx")) {
        const unsigned port = (unsigned)std::atoi(argv[2]);
        const int nframes = std::atoi(argv[3]);
        std::unique_ptr<UDPSource> src(new UDPSourceFEC("127.0.0.1", port));
        if (!(*src)) { std::fprintf(stderr, "source: %s\n", src->error().c_str()); return 1; }
        std::printf("ready\n");
        std::fflush(stdout);
        FILE *f = std::fopen(argv[4], "wb");
        for (int i = 0; i < nframes; ++i) {
            IQSampleVector frame;
            src->read(frame);
            if (!(*src)) { std::fprintf(stderr, "read: %s\n", src->error().c_str()); return 1; }
            std::fwrite(frame.data(), 4, frame.size(), f);
            char msg[64] = "st";
            src->getStatusMessage(msg);
            std::printf("frame %d samples %zu status %s bytes %d bits %d\n", i, frame.size(), msg, (int)src->getSampleBytes(), (int)src->getSampleBits());
        }
        std::fclose(f);
        return 0;
    }
    return 2;
}
