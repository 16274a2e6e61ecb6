// cm256.h -- drop-in replacement of cm256cc's header for the two call sites of the reference:
// UDPSinkFEC.cpp:38,195-246 (isInitialized, cm256_encode) and SDRdaemonFECBuffer.cpp:32-34,42,
// 148-163,197 (cm256_decode).  Same class name, nested types and return conventions (0 = OK);
// the block arithmetic runs in libsdrhip.so on the GPU.
// Every CM256 object owns its sdrhip context: the reference calls cm256_encode on UDPSinkFEC's transmit thread
// (UDPSinkFEC.cpp:193-288) while the main thread decimates, and a context serialises its calls (and shares its
// staging buffers between them), so the codec must not share the decimators' context.
#ifndef SDRHIP_CM256_ADAPTER_H
#define SDRHIP_CM256_ADAPTER_H

#include <string.h> // SDRdaemonFECBuffer.h relies on cm256.h for memcmp/memset

#include "sdrhip_adapter_common.h"

class CM256
{
public:
    // Encoder parameters (cm256cc: cm256_encoder_params)
    typedef struct cm256_encoder_params_t {
        int OriginalCount; // number of original blocks, < 256
        int RecoveryCount; // number of recovery blocks, OriginalCount + RecoveryCount <= 256
        int BlockBytes;    // bytes per block
    } cm256_encoder_params;

    // Block descriptor (cm256cc: cm256_block)
    typedef struct cm256_block_t {
        void *Block;
        unsigned char Index; // 0..OriginalCount-1 original, OriginalCount.. recovery row
    } cm256_block;

    CM256() : m_ctx(nullptr), m_initialized(false)
    {
        try { m_ctx = sdrhip_adapter::new_context(); m_initialized = true; } catch (...) { m_ctx = nullptr; m_initialized = false; }
    }
    ~CM256() { if (m_ctx) sdrhip_ctx_destroy(m_ctx); }
    CM256(const CM256&) = delete;
    CM256& operator=(const CM256&) = delete;
    bool isInitialized() const { return m_initialized; }

    int cm256_encode(cm256_encoder_params params, cm256_block *originals, void *recoveryBlocks)
    {
        if (!m_initialized) return -4;
        sdrhip_cm256_params p = {params.OriginalCount, params.RecoveryCount, params.BlockBytes};
        return sdrhip_cm256_encode(m_ctx, p, reinterpret_cast<const sdrhip_cm256_block *>(originals), recoveryBlocks);
    }

    int cm256_decode(cm256_encoder_params params, cm256_block *blocks)
    {
        if (!m_initialized) return -4;
        sdrhip_cm256_params p = {params.OriginalCount, params.RecoveryCount, params.BlockBytes};
        return sdrhip_cm256_decode(m_ctx, p, reinterpret_cast<sdrhip_cm256_block *>(blocks));
    }

private:
    sdrhip_ctx *m_ctx;
    bool m_initialized;
    static_assert(sizeof(cm256_block) == sizeof(sdrhip_cm256_block), "descriptor layouts must match");
};

#endif
