// ref_cm256_shim.cpp -- C entry points around the REAL cm256cc library (f4exb/cm256cc: cm256.cpp + gf256.cpp), the
// third-party dependency behind the reference's FEC call sites (UDPSinkFEC.cpp:38,195-246; SDRdaemonFECBuffer.cpp:
// 32-34,148-213).  TEST INFRASTRUCTURE ONLY.  The library is NOT part of the reference tree (cm256cc/CMakeLists.txt:
// 12-20 points at an external checkout, ${LIBCM256CCSRC}) and is absent from this machine: oracle/Makefile builds this
// file into _ref/libsdrref_cm256.so only where $(LIBCM256CCSRC)/cm256.cpp exists, compiling the library's sources
// where they lie, unchanged, with its own header.  Until then the FEC half of the oracle stays "parity unpinned".
// Uses exactly the interface the reference's call sites use.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "cm256.h" // from $(LIBCM256CCSRC)

extern "C" {
int sdrref_cm256_initialized(void)
{
    CM256 cm;
    return cm.isInitialized() ? 1 : 0;
}

// originals: k blocks of block_bytes each, contiguous; recovery: m blocks.  Returns cm256_encode's code.
int sdrref_cm256_encode(int k, int m, int block_bytes, const uint8_t *originals, uint8_t *recovery)
{
    CM256 cm;
    CM256::cm256_encoder_params params;
    params.OriginalCount = k; params.RecoveryCount = m; params.BlockBytes = block_bytes;
    std::vector<CM256::cm256_block> blocks((size_t)k);
    for (int i = 0; i < k; ++i) {
        blocks[(size_t)i].Block = const_cast<uint8_t *>(originals) + (size_t)i * block_bytes;
        blocks[(size_t)i].Index = (unsigned char)i;
    }
    return cm.cm256_encode(params, blocks.data(), recovery);
}

// data: k received blocks (in place), indices: their Index fields (in / out).  Returns cm256_decode's code.
int sdrref_cm256_decode(int k, int m, int block_bytes, uint8_t *data, uint8_t *indices)
{
    CM256 cm;
    CM256::cm256_encoder_params params;
    params.OriginalCount = k; params.RecoveryCount = m; params.BlockBytes = block_bytes;
    std::vector<CM256::cm256_block> blocks((size_t)k);
    for (int i = 0; i < k; ++i) {
        blocks[(size_t)i].Block = data + (size_t)i * block_bytes;
        blocks[(size_t)i].Index = indices[i];
    }
    const int rc = cm.cm256_decode(params, blocks.data());
    for (int i = 0; i < k; ++i) indices[i] = blocks[(size_t)i].Index;
    return rc;
}
}
