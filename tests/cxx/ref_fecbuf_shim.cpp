// ref_fecbuf_shim.cpp -- C entry points around the REAL reference class SDRdaemonFECBuffer
// (include/SDRdaemonFECBuffer.h, sdmnbase/SDRdaemonFECBuffer.cpp), compiled where it lies by tests/cxx/Makefile
// against the product's cm256.h adapter (CM256 = libsdrhip.so on the GPU): the reference's own decoder call site
// running on the product.  An integration test of the drop-in boundary, not an oracle.
#include <cstddef>
#include <cstdint>

#include "SDRdaemonFECBuffer.h"

extern "C" {
void *sdrref_fecbuf_new(void) { return new SDRdaemonFECBuffer(); }
void sdrref_fecbuf_free(void *p) { delete static_cast<SDRdaemonFECBuffer *>(p); }
// SDRdaemonFECBuffer::writeAndRead(array, data, dataLength): returns 1 when data was produced
int sdrref_fecbuf_write_and_read(void *p, const uint8_t *superblock, uint8_t *data, size_t *len)
{
    std::size_t n = 0;
    bool avail = static_cast<SDRdaemonFECBuffer *>(p)->writeAndRead(const_cast<uint8_t *>(superblock), data, n);
    *len = n;
    return avail ? 1 : 0;
}
int sdrref_fecbuf_cur_nb_blocks(void *p) { return static_cast<SDRdaemonFECBuffer *>(p)->getCurNbBlocks(); }
int sdrref_fecbuf_cur_nb_recovery(void *p) { return static_cast<SDRdaemonFECBuffer *>(p)->getCurNbRecovery(); }
int sdrref_fecbuf_min_nb_blocks(void *p) { return static_cast<SDRdaemonFECBuffer *>(p)->getMinNbBlocks(); }
int sdrref_fecbuf_max_nb_recovery(void *p) { return static_cast<SDRdaemonFECBuffer *>(p)->getMaxNbRecovery(); }
}
