// UDPSink.h -- drop-in replacement of the reference's UDPSink base class (UDPSink.h:74-102): same
// constructor, virtuals, setters, error() / operator bool contract, so that sdrdaemonrx.cpp:480-490,
// 640-655 compiles unchanged against it.  Self-contained (POSIX sockets, no UDPSocket / CRC64 classes).
#ifndef SDRHIP_UDPSINK_ADAPTER_H
#define SDRHIP_UDPSINK_ADAPTER_H

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <string>

#include "sdrhip_adapter_common.h"

namespace sdrhip_adapter {

// one IPv4 UDP socket; send side resolves its peer once
class UdpSocket
{
public:
    UdpSocket() : m_fd(-1), m_havePeer(false) { std::memset(&m_peer, 0, sizeof(m_peer)); }
    ~UdpSocket() { if (m_fd >= 0) ::close(m_fd); }

    bool open(std::string &err)
    {
        if (m_fd >= 0) return true;
        m_fd = ::socket(AF_INET, SOCK_DGRAM, IPPROTO_UDP);
        if (m_fd < 0) { err = std::string("socket: ") + std::strerror(errno); return false; }
        return true;
    }

    static bool resolve(const std::string &host, unsigned port, sockaddr_in &sa, std::string &err)
    {
        std::memset(&sa, 0, sizeof(sa));
        sa.sin_family = AF_INET;
        sa.sin_port = htons((unsigned short)port);
        if (host.empty() || host == "0.0.0.0") { sa.sin_addr.s_addr = htonl(INADDR_ANY); return true; }
        if (::inet_pton(AF_INET, host.c_str(), &sa.sin_addr) == 1) return true;
        addrinfo hints, *res = 0;
        std::memset(&hints, 0, sizeof(hints));
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_DGRAM;
        if (::getaddrinfo(host.c_str(), 0, &hints, &res) != 0 || !res) { err = "cannot resolve " + host; return false; }
        sa.sin_addr = reinterpret_cast<sockaddr_in *>(res->ai_addr)->sin_addr;
        ::freeaddrinfo(res);
        return true;
    }

    bool setPeer(const std::string &host, unsigned port, std::string &err)
    {
        m_havePeer = open(err) && resolve(host, port, m_peer, err);
        return m_havePeer;
    }

    bool bindLocal(const std::string &host, unsigned port, std::string &err)
    {
        if (!open(err)) return false;
        int one = 1, big = 8 << 20;
        ::setsockopt(m_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        ::setsockopt(m_fd, SOL_SOCKET, SO_RCVBUF, &big, sizeof(big));
        sockaddr_in sa;
        if (!resolve(host, port, sa, err)) return false;
        if (::bind(m_fd, reinterpret_cast<sockaddr *>(&sa), sizeof(sa)) != 0) { err = std::string("bind: ") + std::strerror(errno); return false; }
        return true;
    }

    bool send(const void *p, size_t n)
    {
        return m_havePeer && ::sendto(m_fd, p, n, 0, reinterpret_cast<const sockaddr *>(&m_peer), sizeof(m_peer)) == (ssize_t)n;
    }

    // blocking receive with a timeout; -1 on error, 0 on timeout, else the datagram length
    int recv(void *p, size_t n, int timeout_ms)
    {
        timeval tv;
        tv.tv_sec = timeout_ms / 1000;
        tv.tv_usec = (timeout_ms % 1000) * 1000;
        ::setsockopt(m_fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        ssize_t r = ::recvfrom(m_fd, p, n, 0, 0, 0);
        if (r < 0) return (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) ? 0 : -1;
        return (int)r;
    }

    // non-blocking receive: the datagram length, 0 when nothing is queued, -1 on error
    int recv_nowait(void *p, size_t n)
    {
        ssize_t r = ::recvfrom(m_fd, p, n, MSG_DONTWAIT, 0, 0);
        if (r < 0) return (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) ? 0 : -1;
        return (int)r;
    }

private:
    UdpSocket(const UdpSocket &);
    UdpSocket &operator=(const UdpSocket &);
    int m_fd;
    bool m_havePeer;
    sockaddr_in m_peer;
};

// CRC-32 (IEEE 802.3, reflected, init / xorout 0xFFFFFFFF) = boost::crc_32_type of UDPSinkFEC.cpp:106-109
inline std::uint32_t crc32(const void *data, size_t n)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    std::uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) {
        crc ^= p[i];
        for (int k = 0; k < 8; ++k) crc = (crc & 1u) ? 0xEDB88320u ^ (crc >> 1) : crc >> 1;
    }
    return crc ^ 0xFFFFFFFFu;
}

} // namespace sdrhip_adapter

class UDPSink
{
public:
    UDPSink(const std::string &address, unsigned int port, unsigned int udpSize)
        : m_address(address), m_port(port), m_udpSize(udpSize), m_centerFrequency(100000), m_sampleRate(48000), m_sampleBytes(1),
          m_sampleBits(8), m_nbSamples(0)
    {
        m_socket.setPeer(address, port, m_error);
    }
    virtual ~UDPSink() {}

    /** Write IQ samples to UDP port (UDPSink.h:84) */
    virtual void write(const IQSampleVector &samples_in) = 0;

    /** Return the last error, or an empty string; reading clears it (UDPSink.h:87-92) */
    std::string error()
    {
        std::string ret(m_error);
        m_error.clear();
        return ret;
    }

    void setCenterFrequency(std::uint64_t centerFrequency) { m_centerFrequency = (std::uint32_t)(centerFrequency / 1000); } // Hz in, kHz on the wire
    void setSampleRate(std::uint32_t sampleRate) { m_sampleRate = sampleRate; }
    void setSampleBytes(std::uint8_t sampleBytes) { m_sampleBytes = (std::uint8_t)((sampleBytes & 0x0F) + (m_sampleBytes & 0xF0)); }
    void setSampleBits(std::uint8_t sampleBits) { m_sampleBits = sampleBits; }
    virtual void setNbBlocksFEC(int) {}
    virtual void setTxDelay(int) {}

    /** true if the stream is OK (UDPSink.h:104-108) */
    operator bool() const { return m_error.empty(); }

protected:
    std::string m_address;
    unsigned int m_port;
    unsigned int m_udpSize;
    std::string m_error;
    std::uint32_t m_centerFrequency; // kHz
    std::uint32_t m_sampleRate;      // Hz
    std::uint8_t m_sampleBytes;
    std::uint8_t m_sampleBits;
    std::uint32_t m_nbSamples;
    sdrhip_adapter::UdpSocket m_socket;
};

#endif
