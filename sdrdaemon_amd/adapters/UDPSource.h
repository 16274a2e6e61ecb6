// UDPSource.h -- drop-in replacement of the reference's UDPSource base class (UDPSource.h:83-99):
// same constructor, virtuals, getters and error() / operator bool contract (sdrdaemontx.cpp:381-498).
// Self-contained: the socket helper lives in UDPSink.h of this directory.
#ifndef SDRHIP_UDPSOURCE_ADAPTER_H
#define SDRHIP_UDPSOURCE_ADAPTER_H

#include "UDPSink.h" // sdrhip_adapter::UdpSocket, crc32

class UDPSource
{
public:
    UDPSource(const std::string &address, unsigned int port, unsigned int udpSize)
        : m_address(address), m_port((unsigned short)port), m_udpSize(udpSize), m_sampleBytes(1), m_sampleBits(8), m_nbSamples(0)
    {
    }
    virtual ~UDPSource() {}

    /** Read IQ samples from UDP port (UDPSource.h:92) */
    virtual void read(IQSampleVector &samples_out) = 0;
    /** Append a status message to the C string in messageBuffer (UDPSource.h:97) */
    virtual void getStatusMessage(char *messageBuffer) = 0;

    std::string error()
    {
        std::string ret(m_error);
        m_error.clear();
        return ret;
    }

    std::uint8_t getSampleBytes() const { return m_sampleBytes; }
    std::uint8_t getSampleBits() { return m_sampleBits; }

    operator bool() const { return m_error.empty(); }

protected:
    std::string m_address;
    unsigned short m_port;
    unsigned int m_udpSize;
    std::string m_error;
    std::uint8_t m_sampleBytes;
    std::uint8_t m_sampleBits;
    std::uint32_t m_nbSamples;
    sdrhip_adapter::UdpSocket m_socket;
};

#endif
