// UDPSourceFEC.h -- drop-in replacement of the reference's UDPSourceFEC (UDPSourceFEC.h:62-80,
// UDPSourceFEC.cpp:28-105) together with the frame collector it owns (SDRdaemonFECBuffer,
// SDRdaemonFECBuffer.cpp:28-250): same class name, constructor and virtuals, same frames out.
//
//   read()   receives 512-byte super blocks until the collector releases a frame, i.e. until the first
//            datagram of the NEXT frame arrives (one frame of latency, SDRdaemonFECBuffer.cpp:133-139),
//            and returns the 127 x 127 = 16129 samples of blocks 1..127.
//   collector keeps the FIRST 128 super blocks of a frame in arrival order, originals in place, recovery
//            blocks aside; with the 128th block it CM256-decodes the missing originals on the GPU
//            (sdrhip_cm256_decode: the library's in-place contract, SDRdaemonFECBuffer.cpp:197-213) and
//            updates the stream meta data from block 0; later blocks of the frame are only counted.
//   quirks kept: the very first read() returns the collector's initial (zeroed) slot; getSampleBytes() /
//            getSampleBits() stay at the base class defaults (the reference never updates them).
//
// Header-only, C++11, link with -lsdrhip.  Owns a private sdrhip context.  Without a GPU it behaves like
// the reference without a valid CM256: frames with missing originals come out with holes (zeros).
#ifndef SDRHIP_UDPSOURCEFEC_ADAPTER_H
#define SDRHIP_UDPSOURCEFEC_ADAPTER_H

#include <cstdio>
#include <iostream>
#include <vector>

#include "UDPSource.h"

#define UDPSOURCEFEC_UDPSIZE 512
#define UDPSOURCEFEC_NBORIGINALBLOCKS 128

class UDPSourceFEC : public UDPSource
{
public:
#pragma pack(push, 1)
    struct MetaDataFEC { // SDRdaemonFECBuffer.h:43-66
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

    UDPSourceFEC(const std::string &address, unsigned int port)
        : UDPSource(address, port, UDPSOURCEFEC_UDPSIZE), m_ctx(0), m_frame(NB * BLOCK), m_recovery(NB * BLOCK), m_desc(NB),
          m_blockCount(0), m_recoveryCount(0), m_decoded(false), m_metaRetrieved(false), m_frameHead(-1), m_curNbBlocks(0),
          m_curNbRecovery(0), m_minNbBlocks(256), m_maxNbRecovery(0)
    {
        const char *dev = std::getenv("SDRHIP_DEVICE");
        if (sdrhip_ctx_create(dev ? std::atoi(dev) : 0, 0, &m_ctx) != SDRHIP_OK) {
            std::cerr << "UDPSourceFEC: no GPU context (" << sdrhip_last_error() << "): cannot recover lost blocks" << std::endl;
            m_ctx = 0;
        }
        initMeta(m_currentMeta);
        initMeta(m_outputMeta);
        m_socket.bindLocal(m_address, m_port, m_error);
    }

    virtual ~UDPSourceFEC()
    {
        if (m_ctx) sdrhip_ctx_destroy(m_ctx);
    }

    /** Returns a complete protected frame of 127 * 127 samples (UDPSourceFEC.h:72) */
    virtual void read(IQSampleVector &samples_out)
    {
        unsigned char sb[UDPSOURCEFEC_UDPSIZE];
        std::vector<unsigned char> data((NB - 1) * BLOCK);
        size_t dataLength = 0;
        bool dataAvailable = false;
        while (!dataAvailable) {
            const int received = m_socket.recv(sb, sizeof(sb), 1000);
            if (received < 0) { m_error = "UDPSourceFEC::read: receive error"; return; } // (samples_out untouched)
            if (received == UDPSOURCEFEC_UDPSIZE) dataAvailable = writeAndRead(sb, &data[0], dataLength);
        }
        if (dataLength > 0) {
            samples_out.resize(dataLength / 4);
            std::memcpy(&samples_out[0], &data[0], dataLength);
        }
    }

    /** appends ":<status>:<min blocks>/<max recovery>" (UDPSourceFEC.cpp:80-95) */
    virtual void getStatusMessage(char *messageBuffer)
    {
        const size_t msgLen = std::strlen(messageBuffer);
        const int minNbBlocks = getMinNbBlocks();
        int statusCode;
        if (minNbBlocks < NB) statusCode = 1;                                   // some data is definitely lost
        else if (minNbBlocks < NB + m_currentMeta.m_nbFECBlocks) statusCode = 0; // recoverable or unknown
        else statusCode = 2;                                                     // all OK
        std::sprintf(&messageBuffer[msgLen], ":%d:%03d/%03d", statusCode, minNbBlocks, getMaxNbRecovery());
    }

    // the collector's getters (SDRdaemonFECBuffer.h:100-126)
    const MetaDataFEC &getCurrentMeta() const { return m_currentMeta; }
    const MetaDataFEC &getOutputMeta() const { return m_outputMeta; }
    int getCurNbBlocks() const { return m_curNbBlocks; }
    int getCurNbRecovery() const { return m_curNbRecovery; }
    int getMinNbBlocks() { const int v = m_minNbBlocks; m_minNbBlocks = 256; return v; }      // reading resets
    int getMaxNbRecovery() { const int v = m_maxNbRecovery; m_maxNbRecovery = 0; return v; }

    // one super block in, at most one frame (127 x 508 bytes) out -- SDRdaemonFECBuffer::writeAndRead
    bool writeAndRead(const unsigned char *sb, unsigned char *data, size_t &dataLength)
    {
        bool dataAvailable = false;
        dataLength = 0;
        const int frameIndex = sb[0] | (sb[1] << 8);
        if (m_frameHead != frameIndex) { // first datagram of another frame: release the slot as it is
            dataLength = (size_t)(NB - 1) * BLOCK;
            std::memcpy(data, &m_frame[BLOCK], dataLength); // blocks 1..127
            if (m_metaRetrieved && std::memcmp(&m_frame[0], &m_outputMeta, 12) != 0) std::memcpy(&m_outputMeta, &m_frame[0], sizeof(MetaDataFEC));
            if (!m_decoded)
                std::cerr << "SDRdaemonFECBuffer::getSlotData: incomplete frame: m_blockCount: " << m_blockCount
                          << " m_recoveryCount: " << m_recoveryCount << std::endl;
            dataAvailable = true;
            // statistics of the released frame, then an empty slot
            m_curNbBlocks = m_blockCount;
            m_curNbRecovery = m_recoveryCount;
            if (m_curNbBlocks < m_minNbBlocks) m_minNbBlocks = m_curNbBlocks;
            if (m_curNbRecovery > m_maxNbRecovery) m_maxNbRecovery = m_curNbRecovery;
            m_blockCount = 0;
            m_recoveryCount = 0;
            m_decoded = false;
            m_metaRetrieved = false;
            std::fill(m_frame.begin(), m_frame.end(), 0);
            m_frameHead = frameIndex;
        }
        if (m_blockCount < NB) { // still collecting the first 128
            const int blockIndex = sb[2];
            sdrhip_cm256_block &d = m_desc[m_blockCount];
            d.Index = (unsigned char)blockIndex;
            if (blockIndex == 0) m_metaRetrieved = true;
            if (blockIndex < NB) {
                d.Block = &m_frame[(size_t)blockIndex * BLOCK];
            } else {
                d.Block = &m_recovery[(size_t)m_recoveryCount * BLOCK];
                ++m_recoveryCount;
            }
            std::memcpy(d.Block, sb + 4, BLOCK);
        }
        ++m_blockCount;
        if (m_blockCount == NB) { // 128 blocks in: decode
            m_decoded = true;
            if (m_ctx && m_recoveryCount > 0) {
                sdrhip_cm256_params p = {NB, m_recoveryCount, BLOCK};
                if (sdrhip_cm256_decode(m_ctx, p, &m_desc[0]) != SDRHIP_OK) {
                    std::cerr << "SDRdaemonFECBuffer::writeAndRead: CM256 decode error" << std::endl;
                } else {
                    std::cerr << "SDRdaemonFECBuffer::writeAndRead: CM256 decode success: nb recovery blocks: " << m_recoveryCount << std::endl;
                    // the reference takes the LAST m_recoveryCount descriptors for the recovered blocks (it counts
                    // on the recovery blocks arriving after the originals, SDRdaemonFECBuffer.cpp:208-213)
                    for (int ir = 0; ir < m_recoveryCount; ++ir) {
                        const sdrhip_cm256_block &r = m_desc[NB - m_recoveryCount + ir];
                        std::memmove(&m_frame[(size_t)r.Index * BLOCK], r.Block, BLOCK);
                    }
                }
            }
            if (m_metaRetrieved && std::memcmp(&m_frame[0], &m_currentMeta, 12) != 0) {
                std::memcpy(&m_currentMeta, &m_frame[0], sizeof(MetaDataFEC));
                printMeta(m_currentMeta);
            }
        }
        return dataAvailable;
    }

private:
    static const int NB = UDPSOURCEFEC_NBORIGINALBLOCKS;
    static const int BLOCK = UDPSOURCEFEC_UDPSIZE - 4; // 508 protected bytes = 127 samples

    static void initMeta(MetaDataFEC &m)
    {
        std::memset(&m, 0, sizeof(m));
        m.m_nbFECBlocks = 0xFF; // MetaDataFEC::init(): m_nbFECBlocks = -1
    }

    static void printMeta(const MetaDataFEC &m)
    {
        std::cerr << "|" << m.m_centerFrequency << ":" << m.m_sampleRate << ":" << (int)(m.m_sampleBytes & 0xF) << ":" << (int)m.m_sampleBits
                  << ":" << (int)m.m_nbOriginalBlocks << ":" << (int)m.m_nbFECBlocks << "|" << m.m_tv_sec << ":" << m.m_tv_usec << "|"
                  << std::endl;
    }

    sdrhip_ctx *m_ctx;
    std::vector<unsigned char> m_frame;    // 128 x 508: originals in place (block 0 = meta)
    std::vector<unsigned char> m_recovery; // up to 128 x 508: recovery blocks in arrival order
    std::vector<sdrhip_cm256_block> m_desc; // descriptors of the first 128 blocks in arrival order
    int m_blockCount, m_recoveryCount;
    bool m_decoded, m_metaRetrieved;
    int m_frameHead;
    int m_curNbBlocks, m_curNbRecovery, m_minNbBlocks, m_maxNbRecovery;
    MetaDataFEC m_currentMeta, m_outputMeta;
};

#endif
