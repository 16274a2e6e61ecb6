// cm256.h (test-only) -- a `CM256` class backed by the ORACLE's C restatement, so that the
// reference's own SDRdaemonFECBuffer.cpp can run on the CPU here (cm256cc itself is absent) and
// pin the oracle's restatement of the buffer logic (tests/test_oracle_vs_ref.py).
// TEST INFRASTRUCTURE ONLY; the product's adapter is sdrdaemon_amd/adapters/cm256.h.
#ifndef ORACLE_REF_CM256_H
#define ORACLE_REF_CM256_H
#include <string.h> // the reference header relies on cm256.h for memcmp/memset
#include "../sdr_oracle.h"

class CM256
{
public:
    typedef struct cm256_encoder_params_t { int OriginalCount; int RecoveryCount; int BlockBytes; } cm256_encoder_params;
    typedef struct cm256_block_t { void *Block; unsigned char Index; } cm256_block;
    bool isInitialized() const { return true; }
    int cm256_encode(cm256_encoder_params p, cm256_block *originals, void *recoveryBlocks)
    {
        orc_cm256_params q = {p.OriginalCount, p.RecoveryCount, p.BlockBytes};
        return orc_cm256_encode(q, reinterpret_cast<const orc_cm256_block *>(originals), recoveryBlocks);
    }
    int cm256_decode(cm256_encoder_params p, cm256_block *blocks)
    {
        orc_cm256_params q = {p.OriginalCount, p.RecoveryCount, p.BlockBytes};
        return orc_cm256_decode(q, reinterpret_cast<orc_cm256_block *>(blocks));
    }
};
#endif
