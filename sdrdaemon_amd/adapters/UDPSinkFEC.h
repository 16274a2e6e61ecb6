// UDPSinkFEC.h -- drop-in replacement of the reference's UDPSinkFEC (UDPSinkFEC.h:69-74,
// UDPSinkFEC.cpp:28-288): same class name, constructor and virtuals, same datagrams on the wire.
//
//   write()      packs the incoming (already decimated) IQ samples into frames of 128 super blocks of
//                512 bytes: block 0 = header + MetaDataFEC (24 bytes, CRC-32 over the first 20) + zero
//                fill, blocks 1..127 = header {frameIndex, blockIndex, 0} + 127 samples
//                (UDPSinkFEC.cpp:79-191); a finished frame goes to the transmit thread through a ring
//                of 8 slots (UDPSINKFEC_NBTXBLOCKS); write() blocks while the ring is full (the
//                reference spins with usleep(100) and warns "UDP transmit too slow").
//   tx thread    CM256-encodes the recovery blocks on the GPU (sdrhip_fec_encode_frames = the encode section of
//                transmitUDP, :228-256): ALL the frames that are waiting in the ring with the same nbBlocksFEC go
//                through one call; then sends each frame's 128 + nbBlocksFEC datagrams with usleep(txDelay)
//                after each one (:259-282).
//
// Header-only, C++11, link with -lsdrhip -lpthread.  Owns a private sdrhip context (its transmit thread
// must not share the process-wide one with the Decimators adapter on the main thread).  Without a GPU
// it behaves like the reference without a valid CM256: the originals are sent, no recovery blocks.
#ifndef SDRHIP_UDPSINKFEC_ADAPTER_H
#define SDRHIP_UDPSINKFEC_ADAPTER_H

#include <sys/time.h>

#include <atomic>
#include <condition_variable>
#include <iostream>
#include <mutex>
#include <thread>
#include <vector>

#include "UDPSink.h"

#define UDPSINKFEC_UDPSIZE 512
#define UDPSINKFEC_NBORIGINALBLOCKS 128
#define UDPSINKFEC_NBTXBLOCKS 8

class UDPSinkFEC : public UDPSink
{
public:
    UDPSinkFEC(const std::string &address, unsigned int port)
        : UDPSink(address, port, UDPSINKFEC_UDPSIZE), m_ctx(0), m_nbBlocksFEC(0), m_txDelay(0), m_slots(UDPSINKFEC_NBTXBLOCKS),
          m_fill(0), m_send(0), m_queued(0), m_blockIndex(0), m_sampleIndex(0), m_frameCount(0), m_running(true)
    {
        const char *dev = std::getenv("SDRHIP_DEVICE");
        if (sdrhip_ctx_create(dev ? std::atoi(dev) : 0, 0, &m_ctx) != SDRHIP_OK) {
            std::cerr << "UDPSinkFEC: no GPU context (" << sdrhip_last_error() << "): sending without FEC" << std::endl;
            m_ctx = 0;
        }
        std::memset(&m_lastMeta, 0, sizeof(m_lastMeta));
        m_lastMeta.m_nbFECBlocks = 0xFF; // MetaDataFEC::init(), UDPSinkFEC.h:97-101
        m_thread = std::thread(&UDPSinkFEC::transmit, this);
    }

    virtual ~UDPSinkFEC()
    {
        {
            std::lock_guard<std::mutex> lk(m_mutex);
            m_running = false;
        }
        m_cond.notify_all();
        if (m_thread.joinable()) m_thread.join();
        if (m_ctx) sdrhip_ctx_destroy(m_ctx);
    }

    virtual void setNbBlocksFEC(int nbBlocksFEC)
    {
        std::cerr << "UDPSinkFEC::setNbBlocksFEC: nbBlocksFEC: " << nbBlocksFEC << std::endl;
        m_nbBlocksFEC = nbBlocksFEC;
    }

    virtual void setTxDelay(int txDelay)
    {
        std::cerr << "UDPSinkFEC::setTxDelay: txDelay: " << txDelay << std::endl;
        m_txDelay = txDelay;
    }

    virtual void write(const IQSampleVector &samples_in)
    {
        size_t pos = 0;
        const size_t n = samples_in.size();
        while (pos < n) {
            Slot &slot = m_slots[m_fill];
            if (m_blockIndex == 0) startFrame(slot); // the first sample of a frame stamps its meta block
            unsigned char *blk = slot.blocks[m_blockIndex];
            const size_t room = (size_t)samplesPerBlock - m_sampleIndex;
            const size_t take = n - pos < room ? n - pos : room;
            std::memcpy(blk + 4 + 4 * m_sampleIndex, &samples_in[pos], 4 * take);
            pos += take;
            m_sampleIndex += (int)take;
            if (m_sampleIndex < samplesPerBlock) break; // input used up inside a block
            m_sampleIndex = 0;
            putHeader(blk, m_frameCount, m_blockIndex);
            if (m_blockIndex < UDPSINKFEC_NBORIGINALBLOCKS - 1) {
                ++m_blockIndex;
                continue;
            }
            // frame complete: settings as they are now travel with it (UDPSinkFEC.cpp:160-165)
            slot.frameIndex = m_frameCount;
            slot.nbBlocksFEC = m_nbBlocksFEC;
            slot.txDelay = m_txDelay;
            submit();
            m_blockIndex = 0;
            ++m_frameCount;
        }
    }

private:
#pragma pack(push, 1)
    struct MetaDataFEC { // UDPSinkFEC.h:77-101
        std::uint32_t m_centerFrequency; // kHz
        std::uint32_t m_sampleRate;      // Hz
        std::uint8_t m_sampleBytes;
        std::uint8_t m_sampleBits;
        std::uint8_t m_nbOriginalBlocks;
        std::uint8_t m_nbFECBlocks;
        std::uint32_t m_tv_sec;
        std::uint32_t m_tv_usec;
        std::uint32_t m_crc32;
    };
#pragma pack(pop)
    static_assert(sizeof(MetaDataFEC) == 24, "MetaDataFEC is 24 bytes on the wire");
    static const int samplesPerBlock = (UDPSINKFEC_UDPSIZE - 4) / 4; // 127

    struct Slot {
        unsigned char blocks[UDPSINKFEC_NBORIGINALBLOCKS][UDPSINKFEC_UDPSIZE]; // the 128 originals (recovery blocks: the tx thread's batch buffer)
        std::uint16_t frameIndex;
        int nbBlocksFEC;
        int txDelay;
    };

    static void putHeader(unsigned char *blk, std::uint16_t frameIndex, int blockIndex)
    {
        blk[0] = (unsigned char)(frameIndex & 0xff);
        blk[1] = (unsigned char)(frameIndex >> 8);
        blk[2] = (unsigned char)blockIndex;
        blk[3] = 0;
    }

    void startFrame(Slot &slot)
    {
        timeval tv;
        gettimeofday(&tv, 0);
        MetaDataFEC meta;
        meta.m_centerFrequency = m_centerFrequency;
        meta.m_sampleRate = m_sampleRate;
        meta.m_sampleBytes = m_sampleBytes;
        meta.m_sampleBits = m_sampleBits;
        meta.m_nbOriginalBlocks = UDPSINKFEC_NBORIGINALBLOCKS;
        meta.m_nbFECBlocks = (std::uint8_t)m_nbBlocksFEC.load();
        meta.m_tv_sec = (std::uint32_t)tv.tv_sec;
        meta.m_tv_usec = (std::uint32_t)tv.tv_usec;
        meta.m_crc32 = sdrhip_adapter::crc32(&meta, 20);
        unsigned char *b0 = slot.blocks[0];
        std::memset(b0, 0, UDPSINKFEC_UDPSIZE);
        putHeader(b0, m_frameCount, 0);
        std::memcpy(b0 + 4, &meta, sizeof(meta));
        if (std::memcmp(&meta, &m_lastMeta, 12) != 0) { // the stream parameters changed (UDPSinkFEC.cpp:117-132)
            std::cerr << "UDPSinkFEC::write: meta: |" << meta.m_centerFrequency << ":" << meta.m_sampleRate << ":"
                      << (int)(meta.m_sampleBytes & 0xF) << ":" << (int)meta.m_sampleBits << "|" << (int)meta.m_nbOriginalBlocks << ":"
                      << (int)meta.m_nbFECBlocks << "|" << meta.m_tv_sec << ":" << meta.m_tv_usec << "|" << std::endl;
            m_lastMeta = meta;
        }
        m_blockIndex = 1;
        m_sampleIndex = 0;
    }

    // hand slot m_fill to the transmit thread; wait while every other slot is still waiting to be sent
    void submit()
    {
        std::unique_lock<std::mutex> lk(m_mutex);
        ++m_queued;
        m_cond.notify_all();
        bool warned = false;
        while (m_queued >= UDPSINKFEC_NBTXBLOCKS - 1 && m_running) {
            if (!warned) {
                std::cerr << "UDPSinkFEC::write: warning: UDP transmit too slow" << std::endl;
                warned = true;
            }
            m_cond.wait(lk);
        }
        m_fill = (m_fill + 1) % UDPSINKFEC_NBTXBLOCKS;
    }

    void transmit()
    {
        std::vector<unsigned char> orig, rec;
        for (;;) {
            int queued;
            {
                std::unique_lock<std::mutex> lk(m_mutex);
                // like the reference's thread, a frame leaves when the NEXT one is complete: write() stores the index of the
                // frame it just finished in m_txIndexCurrent and transmitUDP waits while that equals the frame it is about to
                // process (UDPSinkFEC.cpp:160,206-211) -- so the newest finished frame always stays behind (and the last frame of
                // a run is never sent, there as here)
                while (m_queued < 2 && m_running) m_cond.wait(lk);
                if (!m_running) return;
                queued = m_queued - 1;
            }
            // the waiting frames with the same fecblk as the first one: one GPU call for all of them
            int nb = m_slots[m_send].nbBlocksFEC;
            if (nb < 0 || nb > 128 || !m_ctx) nb = 0;
            int batch = 1;
            while (batch < queued && m_slots[(m_send + batch) % UDPSINKFEC_NBTXBLOCKS].nbBlocksFEC == m_slots[m_send].nbBlocksFEC) ++batch;
            if (nb > 0) {
                const size_t fb = (size_t)UDPSINKFEC_NBORIGINALBLOCKS * UDPSINKFEC_UDPSIZE, rb = (size_t)nb * UDPSINKFEC_UDPSIZE;
                const unsigned char *frames = &m_slots[m_send].blocks[0][0];
                if (batch > 1) { // the slots are not adjacent in memory: gather the originals
                    orig.resize((size_t)batch * fb);
                    for (int b = 0; b < batch; ++b) std::memcpy(&orig[(size_t)b * fb], &m_slots[(m_send + b) % UDPSINKFEC_NBTXBLOCKS].blocks[0][0], fb);
                    frames = &orig[0];
                }
                rec.resize((size_t)batch * rb);
                if (sdrhip_fec_encode_frames(m_ctx, frames, (size_t)batch, nb, &rec[0], SDRHIP_MEM_HOST) != SDRHIP_OK) {
                    std::cerr << "UDPSinkFEC::transmitUDP: CM256 encode failed (" << sdrhip_last_error() << "). No transmission." << std::endl;
                    return; // (the reference's transmit thread ends here as well, UDPSinkFEC.cpp:246-250)
                }
            }
            for (int b = 0; b < batch; ++b) {
                Slot &slot = m_slots[m_send];
                for (int i = 0; i < UDPSINKFEC_NBORIGINALBLOCKS + nb; ++i) {
                    const unsigned char *blk = i < UDPSINKFEC_NBORIGINALBLOCKS ? slot.blocks[i]
                                                                                : &rec[((size_t)b * nb + (size_t)(i - UDPSINKFEC_NBORIGINALBLOCKS)) * UDPSINKFEC_UDPSIZE];
                    m_socket.send(blk, UDPSINKFEC_UDPSIZE);
                    usleep((useconds_t)(slot.txDelay > 0 ? slot.txDelay : 0));
                }
                {
                    std::lock_guard<std::mutex> lk(m_mutex);
                    --m_queued;
                    m_send = (m_send + 1) % UDPSINKFEC_NBTXBLOCKS;
                }
                m_cond.notify_all();
            }
        }
    }

    sdrhip_ctx *m_ctx;
    std::atomic<int> m_nbBlocksFEC;
    std::atomic<int> m_txDelay;
    std::vector<Slot> m_slots;
    int m_fill;   // slot being filled by write()
    int m_send;   // slot the transmit thread works on next
    int m_queued; // finished frames not yet sent (guarded by m_mutex)
    int m_blockIndex;
    int m_sampleIndex;
    std::uint16_t m_frameCount;
    MetaDataFEC m_lastMeta;
    bool m_running;
    std::mutex m_mutex;
    std::condition_variable m_cond;
    std::thread m_thread;
};

#endif
