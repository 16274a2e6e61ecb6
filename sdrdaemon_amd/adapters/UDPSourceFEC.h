// UDPSourceFEC.h -- drop-in replacement of the reference's UDPSourceFEC (UDPSourceFEC.h:62-80,
// UDPSourceFEC.cpp:28-105) together with the frame collector it owns (SDRdaemonFECBuffer,
// SDRdaemonFECBuffer.cpp:28-250): same class name, constructor and virtuals, same frames out.
//
//   read()   returns the 127 x 127 = 16129 samples (blocks 1..127) of the next frame.  A frame is released when the
//            first datagram of ANOTHER frame arrives (one frame of latency, SDRdaemonFECBuffer.cpp:133-139).
//   collector keeps the FIRST 128 super blocks of a frame in arrival order (.cpp:143-166); later blocks are only
//            counted.  Unlike the reference it does not decode frame by frame: read() drains whatever the socket
//            already holds, and every released frame that used recovery blocks goes to the GPU in ONE asynchronous
//            batch (sdrhip_tx_submit on a Tx pipe with interpolation factor 1: upload + plan + decode + download
//            enqueued, the call returns; up to MAXBATCH frames, up to 4 batches in flight); frames
//            without recovery blocks, and incomplete ones (holes stay zero, "incomplete frame" is logged), are put
//            together on the host.  Results wait in a queue; read() hands them out one per call -- a frame whose batch
//            is still on the GPU is waited for only when its turn comes (sdrhip_tx_collect), the datagrams of the next
//            frames are taken off the socket meanwhile -- together with the
//            statistics of that frame (getCurNbBlocks ... getMaxNbRecovery, status string) and the stream meta data.
//   quirks kept: the very first read() returns the collector's initial (zeroed) slot; getSampleBytes() /
//            getSampleBits() stay at the base class defaults (the reference never updates them); exactly one
//            recovery block = cm256's XOR shortcut whatever its row (inside the library).
//
// Header-only, C++11, link with -lsdrhip.  Owns a private sdrhip context.  Without a GPU it behaves like
// the reference without a valid CM256: frames with missing originals come out with holes (zeros).
#ifndef SDRHIP_UDPSOURCEFEC_ADAPTER_H
#define SDRHIP_UDPSOURCEFEC_ADAPTER_H

#include <cstdio>
#include <deque>
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
        : UDPSource(address, port, UDPSOURCEFEC_UDPSIZE), m_ctx(0), m_tx(0), m_seqNext(0), m_seqFront(0), m_curNbBlocks(0), m_curNbRecovery(0),
          m_minNbBlocks(256), m_maxNbRecovery(0)
    {
        const char *dev = std::getenv("SDRHIP_DEVICE");
        if (sdrhip_ctx_create(dev ? std::atoi(dev) : 0, 0, &m_ctx) != SDRHIP_OK) {
            std::cerr << "UDPSourceFEC: no GPU context (" << sdrhip_last_error() << "): cannot recover lost blocks" << std::endl;
            m_ctx = 0;
        }
        // the decoder: a Tx pipe with interpolation factor 1 = SDRdaemonFECBuffer's decode + getSlotData, asynchronous entry
        if (m_ctx && (sdrhip_tx_create(m_ctx, 1, 0, &m_tx) != SDRHIP_OK || sdrhip_tx_set_async(m_tx, MAXINFLIGHT) != SDRHIP_OK)) {
            std::cerr << "UDPSourceFEC: no decoder (" << sdrhip_last_error() << "): cannot recover lost blocks" << std::endl;
            if (m_tx) sdrhip_tx_destroy(m_tx);
            m_tx = 0;
        }
        initMeta(m_currentMeta);
        initMeta(m_outputMeta);
        m_open.reset(-1); // m_frameHead = -1 (SDRdaemonFECBuffer.cpp:36): the first datagram releases this empty slot
        m_socket.bindLocal(m_address, m_port, m_error);
    }

    virtual ~UDPSourceFEC()
    {
        if (m_tx) sdrhip_tx_destroy(m_tx);
        if (m_ctx) sdrhip_ctx_destroy(m_ctx);
    }

    /** Returns a complete protected frame of 127 * 127 samples (UDPSourceFEC.h:72) */
    virtual void read(IQSampleVector &samples_out)
    {
        unsigned char sb[UDPSOURCEFEC_UDPSIZE];
        while (m_ready.empty()) {
            // wait for the datagram that releases a frame ...
            while (m_closed.empty()) {
                const int received = m_socket.recv(sb, sizeof(sb), 1000);
                if (received < 0) { m_error = "UDPSourceFEC::read: receive error"; return; } // (samples_out untouched)
                if (received == UDPSOURCEFEC_UDPSIZE) feed(sb);
            }
            // ... then take what the socket already holds: more released frames make a bigger batch for the GPU
            while ((int)m_closed.size() < MAXBATCH) {
                const int received = m_socket.recv_nowait(sb, sizeof(sb));
                if (received <= 0) break;
                if (received == UDPSOURCEFEC_UDPSIZE) feed(sb);
            }
            decodeClosed();
        }
        while (m_ready.front().pending) collectBatch(); // its batch is still on the GPU: now it is needed
        Ready &r = m_ready.front();
        samples_out.resize((size_t)(NB - 1) * BLOCK / 4);
        std::memcpy(&samples_out[0], &r.frame[BLOCK], (size_t)(NB - 1) * BLOCK); // blocks 1..127
        // meta data and statistics of the frame handed out (SDRdaemonFECBuffer.cpp:72-110,170-247)
        if (r.meta && r.decoded && std::memcmp(&r.frame[0], &m_currentMeta, 12) != 0) {
            std::memcpy(&m_currentMeta, &r.frame[0], sizeof(MetaDataFEC));
            printMeta(m_currentMeta);
        }
        if (r.meta && std::memcmp(&r.frame[0], &m_outputMeta, 12) != 0) std::memcpy(&m_outputMeta, &r.frame[0], sizeof(MetaDataFEC));
        m_curNbBlocks = r.count;
        m_curNbRecovery = r.nrec;
        if (m_curNbBlocks < m_minNbBlocks) m_minNbBlocks = m_curNbBlocks;
        if (m_curNbRecovery > m_maxNbRecovery) m_maxNbRecovery = m_curNbRecovery;
        m_ready.pop_front();
        ++m_seqFront;
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

    static const int MAXBATCH = 8;    // frames per GPU batch
    static const int MAXINFLIGHT = 4; // batches in flight

    // a frame being collected / released: the first 128 super blocks as they arrived, headers included
    struct Collect {
        int index;
        int count, nrec; // all blocks seen; recovery blocks among the first 128
        bool meta;       // block 0 is among them
        std::vector<unsigned char> rx;
        Collect() : index(-1), count(0), nrec(0), meta(false), rx((size_t)NB * UDPSOURCEFEC_UDPSIZE) {}
        void reset(int idx) { index = idx; count = 0; nrec = 0; meta = false; }
    };
    // a frame ready to be handed out: 128 x 508 bytes (block 0 = meta), holes zero
    struct Ready {
        std::vector<unsigned char> frame;
        int count, nrec;
        bool meta, decoded;
        bool pending; // its recovered blocks are still on the GPU (a submitted batch)
        Ready() : frame((size_t)NB * BLOCK, 0), count(0), nrec(0), meta(false), decoded(false), pending(false) {}
    };

    // one super block into the collector (SDRdaemonFECBuffer.cpp:112-170 without the decode)
    void feed(const unsigned char *sb)
    {
        const int frameIndex = sb[0] | (sb[1] << 8);
        if (m_open.index != frameIndex) { // first datagram of another frame: release the slot as it is
            m_closed.push_back(m_open);
            m_open.reset(frameIndex);
        }
        if (m_open.count < NB) { // still collecting the first 128
            std::memcpy(&m_open.rx[(size_t)m_open.count * UDPSOURCEFEC_UDPSIZE], sb, UDPSOURCEFEC_UDPSIZE);
            if (sb[2] == 0) m_open.meta = true;
            if (sb[2] >= NB) ++m_open.nrec;
        }
        ++m_open.count;
    }

    // the released frames -> m_ready, in order; those that used recovery blocks as ONE asynchronous GPU batch
    void decodeClosed()
    {
        std::vector<unsigned long> seqs; // sequence numbers (position in the hand-out order) of the frames that go to the GPU
        std::vector<unsigned char> rxb;
        int maxrec = 0; // most recovery blocks any frame of this batch holds: the collector counted them, so this is no guess
        for (std::deque<Collect>::iterator it = m_closed.begin(); it != m_closed.end(); ++it) {
            m_ready.push_back(Ready());
            Ready &r = m_ready.back();
            const unsigned long seq = m_seqNext++;
            r.count = it->count; r.nrec = it->nrec; r.meta = it->meta; r.decoded = it->count >= NB;
            const int have = it->count < NB ? it->count : NB;
            for (int p = 0; p < have; ++p) { // received originals in place
                const unsigned char *sb = &it->rx[(size_t)p * UDPSOURCEFEC_UDPSIZE];
                if (sb[2] < NB) std::memcpy(&r.frame[(size_t)sb[2] * BLOCK], sb + 4, BLOCK);
            }
            if (!r.decoded) {
                if (it->index >= 0) // (the initial empty slot is released silently)
                    std::cerr << "SDRdaemonFECBuffer::getSlotData: incomplete frame: m_blockCount: " << it->count
                              << " m_recoveryCount: " << it->nrec << std::endl;
            } else if (it->nrec > 0 && m_tx) {
                if (it->nrec > maxrec) maxrec = it->nrec;
                seqs.push_back(seq);
                rxb.insert(rxb.end(), it->rx.begin(), it->rx.end());
            }
        }
        m_closed.clear();
        if (seqs.empty()) return;
        while ((int)m_batches.size() >= MAXINFLIGHT) collectBatch(); // (the ring is full: the oldest batch first)
        {
            // tell the library how many recovery blocks a frame of this batch can hold (the option "dec_max_rows", sdrhip.h): with
            // <= 32 -- every sender up to fecblk 32 -- the decoder plans each frame inside its own launch (one kernel instead of two)
            char v[16];
            std::snprintf(v, sizeof(v), "%d", maxrec < 1 ? 1 : maxrec);
            (void)sdrhip_ctx_set_option(m_ctx, "dec_max_rows", v);
        }
        if (sdrhip_tx_submit(m_tx, &rxb[0], 0, seqs.size(), 0) != SDRHIP_OK) {
            std::cerr << "SDRdaemonFECBuffer::writeAndRead: CM256 decode error (" << sdrhip_last_error() << ")" << std::endl;
            return; // the frames keep what was received
        }
        for (size_t g = 0; g < seqs.size(); ++g) m_ready[seqs[g] - m_seqFront].pending = true;
        m_batches.push_back(seqs);
    }

    // the oldest batch back from the GPU: its frames' payload (blocks 1..127) and meta blocks into their Ready entries
    void collectBatch()
    {
        if (m_batches.empty()) { // (cannot happen: a pending frame belongs to a batch) -- never spin on it
            for (std::deque<Ready>::iterator it = m_ready.begin(); it != m_ready.end(); ++it) it->pending = false;
            return;
        }
        const std::vector<unsigned long> seqs = m_batches.front(); // (popped only once the library has consumed ITS oldest batch)
        const size_t per = (size_t)(NB - 1) * BLOCK; // bytes of payload per frame = 16129 samples
        std::vector<unsigned char> payload(seqs.size() * per), b0(seqs.size() * (size_t)BLOCK);
        size_t n_out = 0, nf = 0;
        const int rc = sdrhip_tx_collect(m_tx, reinterpret_cast<int16_t *>(&payload[0]), 0, payload.size() / 4, &b0[0], &n_out, &nf, 1);
        if (rc != SDRHIP_OK) {
            // the library's batch is still at the head of ITS ring (a collect that fails consumes nothing): dropping only our entry would
            // pair every later collect with the wrong frames.  Every batch in flight is given up (the frames keep what was received)
            // and the handle is replaced, so that the two rings start over in step.
            std::cerr << "SDRdaemonFECBuffer::writeAndRead: CM256 decode error (" << sdrhip_last_error() << ")" << std::endl;
            resetTx();
            return;
        }
        m_batches.pop_front();
        const bool ok = nf == seqs.size() && n_out * 4 == payload.size();
        if (!ok) std::cerr << "SDRdaemonFECBuffer::writeAndRead: CM256 decode error (batch of " << nf << " frames for " << seqs.size() << ")" << std::endl;
        for (size_t g = 0; g < seqs.size(); ++g) {
            if (seqs[g] < m_seqFront) continue; // (handed out already: cannot happen, frames wait for their batch)
            Ready &r = m_ready[seqs[g] - m_seqFront];
            r.pending = false;
            if (!ok) continue; // the frame keeps what was received
            std::memcpy(&r.frame[0], &b0[g * BLOCK], BLOCK);
            std::memcpy(&r.frame[BLOCK], &payload[g * per], per);
        }
    }

    // give up every batch in flight and replace the library handle (collectBatch: a collect that failed without consuming its batch)
    void resetTx()
    {
        for (std::deque<Ready>::iterator it = m_ready.begin(); it != m_ready.end(); ++it) it->pending = false;
        m_batches.clear();
        if (m_tx) sdrhip_tx_destroy(m_tx);
        m_tx = 0;
        if (m_ctx && (sdrhip_tx_create(m_ctx, 1, 0, &m_tx) != SDRHIP_OK || sdrhip_tx_set_async(m_tx, MAXINFLIGHT) != SDRHIP_OK)) {
            std::cerr << "UDPSourceFEC: cannot re-create the decoder (" << sdrhip_last_error() << "): frames are delivered as received" << std::endl;
            if (m_tx) sdrhip_tx_destroy(m_tx);
            m_tx = 0;
        }
    }

    sdrhip_ctx *m_ctx;
    sdrhip_tx *m_tx;
    Collect m_open;
    std::deque<Collect> m_closed;
    std::deque<Ready> m_ready;
    std::deque<std::vector<unsigned long> > m_batches; // frames (by sequence number) of the batches in flight, oldest first
    unsigned long m_seqNext, m_seqFront;               // sequence number of the next released frame / of m_ready.front()
    int m_curNbBlocks, m_curNbRecovery, m_minNbBlocks, m_maxNbRecovery;
    MetaDataFEC m_currentMeta, m_outputMeta;
};

#endif
