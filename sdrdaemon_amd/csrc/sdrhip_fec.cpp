// sdrhip_fec.cpp -- host side of the CM256 entry points (include/sdrhip.h): planning on the
// host (gf256.cpp), block arithmetic on the GPU (gf_kernels.hip).
#include "gf256.h"
#include "sdrhip_host.h"
#include <iterator>

#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace sdrhip;

namespace sdrhip {

// below this many recovery blocks the generic kernel (rows x 128 products) is cheaper than a full
// 32-row Karatsuba tile (13.7 k lane-ops per column vs 36 k x rows / 32): ENC128_MIN_ROWS, sdrhip_internal.h

int fec_encode128_launch(sdrhip_ctx *c, const Enc128Args &k, hipStream_t on)
{
    hipError_t e;
    {
        KTimer kt(c, SDRHIP_K_FEC_ENCODE, on);
        Enc128Args ks = k;
        ks.stagger = c->opt.fec_stagger; ks.stagger_div = c->opt.fec_stagger_mod ? -c->opt.fec_stagger_mod : c->n_cu;
        ks.half_units = c->opt.enc_half;
        e = launch_gf_encode128(ks, on ? on : c->stream);
    }
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec encode launch: %s", hipGetErrorString(e));
    return SDRHIP_OK;
}

int fec_encode_device(sdrhip_ctx *c, const uint8_t *frames, size_t frame_bytes, size_t nframes, int nb_fec, uint8_t *rec,
                      size_t rec_frame_bytes, const int32_t *frame_list_dev, int ngroups, const EncodeLin *lin)
{
    if (nframes == 0 || nb_fec <= 0) return SDRHIP_OK;
    hipError_t e;
    if (nb_fec >= enc128_min_rows(c) && rec_frame_bytes % 4 == 0 && frame_bytes % 4 == 0) {
        // structured encoder (Karatsuba over the XOR-convolution form of the Cauchy rows); frame lists of
        // the generic kernel come in groups of GF_FRAMES_PER_GROUP, this kernel takes them flat
        Enc128Args k;
        memset(&k, 0, sizeof(k));
        k.in = frames; k.out = rec; k.tab = c->gf_tab; k.leaf_tables = c->enc_leaves; k.fft_tables = c->enc_fft; k.use_fft = c->opt.enc_fft;
        k.in_frame_bytes = frame_bytes; k.out_frame_bytes = rec_frame_bytes;
        k.rows = nb_fec; k.nframes = (int)nframes;
        k.frame_list = frame_list_dev; k.nlist = frame_list_dev ? ngroups * GF_FRAMES_PER_GROUP : (int)nframes;
        if (lin) { k.lin = lin->lin; k.lin_stride = lin->stride; k.lin_cap = lin->cap; k.lin_first = lin->first; k.lin_pending = lin->pending; }
        k.stagger = c->opt.fec_stagger; k.stagger_div = c->opt.fec_stagger_mod ? -c->opt.fec_stagger_mod : c->n_cu;
        k.half_units = c->opt.enc_half;
        {
            KTimer kt(c, SDRHIP_K_FEC_ENCODE);
            e = launch_gf_encode128(k, c->stream);
        }
        if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec encode launch: %s", hipGetErrorString(e));
        return SDRHIP_OK; // (the kernel writes the recovery block headers as well)
    }
    GfArgs a;
    memset(&a, 0, sizeof(a));
    a.in = frames; a.out = rec; a.coef = c->enc_matrix; a.tab = c->gf_tab;
    a.in_frame_bytes = frame_bytes; a.out_frame_bytes = rec_frame_bytes;
    a.in_pitch = SDRHIP_UDPSIZE; a.out_pitch = SDRHIP_UDPSIZE; a.in_off = 4; a.out_off = 4;
    a.rows = nb_fec; a.cols = SDRHIP_NB_ORIGINAL; a.coef_per_frame = 0;
    a.nframes = (int)nframes;
    a.frame_list = frame_list_dev; a.ngroups = ngroups;
    {
        KTimer kt(c, SDRHIP_K_FEC_ENCODE);
        e = launch_gf_apply(a, c->stream);
    }
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec encode launch: %s", hipGetErrorString(e));
    // headers {frameIndex, 128 + r, filler 0} of the recovery super blocks (UDPSinkFEC.cpp:239-243)
    e = launch_fec_headers(frames, frame_bytes, rec, rec_frame_bytes, nb_fec, SDRHIP_NB_ORIGINAL, (int)nframes, frame_list_dev,
                           ngroups * GF_FRAMES_PER_GROUP, c->stream);
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec header launch: %s", hipGetErrorString(e));
    return SDRHIP_OK;
}

// Decode section of SDRdaemonFECBuffer::writeAndRead (.cpp:143-213) + getSlotData (:72-75) for a batch of frames,
// entirely on the device: a planning kernel per frame (closed-form inverse of the Cauchy block, gf_kernels.hip), the
// scatter of the received originals, the matrix apply.  No host synchronisation, no limit on the number of distinct
// erasure patterns in a batch.  indices: optional HOST array (nframes x 128 block indices in arrival order); NULL = the
// kernels read header.blockIndex of the super blocks themselves (:147).
bool fec_decode_gather_ok(const sdrhip_ctx *c)
{
    // (the conditions under which launch_fec_decode_device_plan takes its one-launch branch)
    return c->opt.dec_syndrome && c->opt.enc_fft && c->enc_fft && c->opt.dec_fused_plan && c->opt.dec_max_rows <= 32;
}

int fec_decode_device(sdrhip_ctx *c, const uint8_t *rx, size_t rx_frame_bytes, const uint8_t *indices, size_t nframes,
                      uint8_t *payload_out, size_t payload_frame_bytes, uint8_t *block0_out, const DecodeSide *side, const DecodeGather *gather)
{
    if (gather && !fec_decode_gather_ok(c)) return fail(SDRHIP_EINVAL, "internal: no-copy decode without the fused-plan decoder");
    // (side: a pipelined Tx pipe decodes on the context's second stream with work buffers of its own, so that the context's
    // shared ones stay free for whatever runs on the first stream meanwhile)
    hipStream_t st = side ? side->stream : c->stream;
    DevBuf &planbuf = side ? *side->plan : c->dec_plan;
    DevBuf &idxbuf = side ? *side->idx : c->aux;
    PinnedBuf &pinbuf = side ? *side->pin : c->pin;
    if (nframes == 0) return SDRHIP_OK;
    if (nframes > 0x3fffffffu) return fail(SDRHIP_EINVAL, "fec decode: too many frames in one call");
    // (the dense path's scatter pass puts the frame index in gridDim.y: 65535 at most; the default syndrome path has no such pass)
    if (!c->opt.dec_syndrome && nframes > 65535) return fail(SDRHIP_EINVAL, "fec decode (dec_path = dense): at most 65535 frames per call");
    int rc;
    if ((rc = planbuf.reserve(DecodeBuffers::bytes(nframes)))) return rc;
    DecodeBuffers d;
    uint8_t *base = planbuf.as<uint8_t>();
    d.coef = base; base += nframes * (size_t)128 * 128;
    d.pmap = reinterpret_cast<int16_t *>(base); base += nframes * 128 * sizeof(int16_t);
    d.zmap = reinterpret_cast<int16_t *>(base); base += nframes * 128 * sizeof(int16_t);
    d.pdst = reinterpret_cast<int16_t *>(base); base += nframes * 128 * sizeof(int16_t);
    d.zdst = reinterpret_cast<int16_t *>(base); base += nframes * 128 * sizeof(int16_t);
    d.nrec = reinterpret_cast<int32_t *>(base); base += (nframes * 2 * sizeof(int32_t) + 15) & ~(size_t)15;
    // syndrome decoder (default) or the dense matrix kernel alone (ctx option dec_path = dense: A / B and fallback)
    d.plan2 = c->opt.dec_syndrome ? base : nullptr;
    d.leaf_tables = c->enc_leaves;
    d.fft_tables = c->enc_fft; d.use_fft = c->opt.enc_fft;
    d.stagger = c->opt.fec_stagger; d.stagger_div = c->opt.fec_stagger_mod ? -c->opt.fec_stagger_mod : c->n_cu;
    d.fused_plan = c->opt.dec_fused_plan;
    d.srcmap = gather ? gather->srcmap : nullptr; d.restored = gather ? gather->restored : nullptr; d.restored_rows = gather ? gather->rows : 0;
    const uint8_t *idx_dev = nullptr;
    if (indices) {
        const size_t nb = nframes * (size_t)SDRHIP_NB_ORIGINAL;
        if ((rc = pinbuf.reserve(nb))) return rc;
        if ((rc = idxbuf.reserve(nb))) return rc;
        memcpy(pinbuf.p, indices, nb);
        HIP_TRY(hipMemcpyAsync(idxbuf.p, pinbuf.p, nb, hipMemcpyHostToDevice, st));
        pinbuf.mark(st);
        idx_dev = idxbuf.as<uint8_t>();
    }
    hipError_t e;
    {
        KTimer kt(c, SDRHIP_K_FEC_DECODE, side ? st : nullptr);
        e = launch_fec_decode_device_plan(d, rx, rx_frame_bytes, idx_dev, c->gf_explog, c->gf_tab, (int)nframes, payload_out,
                                          payload_frame_bytes, block0_out, c->opt.dec_max_rows, c->opt.dec_strict, c->dec_stats, st);
    }
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec decode launch: %s", hipGetErrorString(e));
    return SDRHIP_OK;
}

} // namespace sdrhip

// ---------------------------------------------------------------------------------- C ABI
extern "C" int sdrhip_fec_encode_frames(sdrhip_ctx *c, const uint8_t *frames, size_t nframes, int nb_fec, uint8_t *recovery_out,
                                        int mem)
{
    if (!c) return fail(SDRHIP_EINVAL, "ctx is NULL");
    sdrhip::CtxLock lock_(c);
    if (nb_fec < 0 || nb_fec > 128) return fail(SDRHIP_EINVAL, "nb_fec must be 0..128 (OriginalCount + RecoveryCount <= 256)");
    if (nframes == 0 || nb_fec == 0) return SDRHIP_OK;
    if (!frames || !recovery_out) return fail(SDRHIP_EINVAL, "fec_encode_frames: NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t fb = (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE, rb = (size_t)nb_fec * SDRHIP_UDPSIZE;
    if (mem == SDRHIP_MEM_DEVICE) return fec_encode_device(c, frames, fb, nframes, nb_fec, recovery_out, rb);
    if (mem != SDRHIP_MEM_HOST) return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    int rc;
    if ((rc = c->in.reserve(nframes * fb))) return rc;
    if ((rc = c->out.reserve(nframes * rb))) return rc;
    HIP_TRY(hipMemcpyAsync(c->in.p, frames, nframes * fb, hipMemcpyHostToDevice, c->stream));
    if ((rc = fec_encode_device(c, c->in.as<uint8_t>(), fb, nframes, nb_fec, c->out.as<uint8_t>(), rb))) return rc;
    HIP_TRY(hipMemcpyAsync(recovery_out, c->out.p, nframes * rb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDRHIP_OK;
}

extern "C" int sdrhip_fec_decode_frames(sdrhip_ctx *c, const uint8_t *rx, const uint8_t *indices, size_t nframes, uint8_t *payload_out,
                                        uint8_t *block0_out, int mem)
{
    if (!c) return fail(SDRHIP_EINVAL, "ctx is NULL");
    sdrhip::CtxLock lock_(c);
    if (nframes == 0) return SDRHIP_OK;
    if (!rx || !payload_out) return fail(SDRHIP_EINVAL, "fec_decode_frames: NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t fb = (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE, pb = (size_t)127 * SDRHIP_BLOCK_BYTES;
    if (mem == SDRHIP_MEM_DEVICE) return fec_decode_device(c, rx, fb, indices, nframes, payload_out, pb, block0_out);
    if (mem != SDRHIP_MEM_HOST) return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    int rc;
    if ((rc = c->in.reserve(nframes * fb))) return rc;
    if ((rc = c->out.reserve(nframes * pb))) return rc;
    if (block0_out && (rc = c->aux3.reserve(nframes * SDRHIP_BLOCK_BYTES))) return rc;
    HIP_TRY(hipMemcpyAsync(c->in.p, rx, nframes * fb, hipMemcpyHostToDevice, c->stream));
    if ((rc = fec_decode_device(c, c->in.as<uint8_t>(), fb, indices, nframes, c->out.as<uint8_t>(), pb, block0_out ? c->aux3.as<uint8_t>() : nullptr)))
        return rc;
    HIP_TRY(hipMemcpyAsync(payload_out, c->out.p, nframes * pb, hipMemcpyDeviceToHost, c->stream));
    if (block0_out) HIP_TRY(hipMemcpyAsync(block0_out, c->aux3.p, nframes * SDRHIP_BLOCK_BYTES, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDRHIP_OK;
}

// CM256::cm256_encode on host pointers (UDPSinkFEC.cpp:246).  Any geometry the library accepts.
extern "C" int sdrhip_cm256_encode(sdrhip_ctx *c, sdrhip_cm256_params p, const sdrhip_cm256_block *originals, void *recoveryBlocks)
{
    if (!c) return fail(SDRHIP_EINVAL, "ctx is NULL");
    sdrhip::CtxLock lock_(c);
    if (p.OriginalCount <= 0 || p.RecoveryCount <= 0 || p.BlockBytes <= 0) return fail(-1, "cm256_encode: invalid params"); // upstream -1
    if (p.OriginalCount + p.RecoveryCount > 256) return fail(-2, "cm256_encode: OriginalCount + RecoveryCount > 256");      // upstream -2
    if (!originals || !recoveryBlocks) return fail(-3, "cm256_encode: NULL pointer");                                         // upstream -3
    HIP_TRY(hipSetDevice(c->device));
    const int k = p.OriginalCount, m = p.RecoveryCount;
    if (k == 1) { // upstream: copies of the single original
        for (int r = 0; r < m; ++r) memcpy(static_cast<uint8_t *>(recoveryBlocks) + (size_t)r * p.BlockBytes, originals[0].Block, (size_t)p.BlockBytes);
        return SDRHIP_OK;
    }
    // blocks are processed in 508-byte column slabs of a pitch-512 staging layout
    const int bb = p.BlockBytes;
    const int nslab = (bb + SDRHIP_BLOCK_BYTES - 1) / SDRHIP_BLOCK_BYTES; // slabs act as independent "frames"
    const size_t in_frame = (size_t)k * SDRHIP_UDPSIZE, out_frame = (size_t)m * SDRHIP_UDPSIZE;
    std::vector<uint8_t> hin(in_frame * nslab, 0), hout(out_frame * nslab), mat((size_t)m * k);
    for (int s = 0; s < nslab; ++s) {
        const int off = s * SDRHIP_BLOCK_BYTES, len = (bb - off) < SDRHIP_BLOCK_BYTES ? (bb - off) : SDRHIP_BLOCK_BYTES;
        for (int j = 0; j < k; ++j) memcpy(&hin[s * in_frame + (size_t)j * SDRHIP_UDPSIZE + 4], static_cast<const uint8_t *>(originals[j].Block) + off, (size_t)len);
    }
    cm256_encode_matrix(k, m, mat.data());
    int rc;
    if ((rc = c->in.reserve(hin.size()))) return rc;
    if ((rc = c->out.reserve(hout.size()))) return rc;
    if ((rc = c->aux.reserve(mat.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(c->in.p, hin.data(), hin.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->aux.p, mat.data(), mat.size(), hipMemcpyHostToDevice, c->stream));
    GfArgs a;
    memset(&a, 0, sizeof(a));
    a.in = c->in.as<uint8_t>(); a.out = c->out.as<uint8_t>(); a.coef = c->aux.as<uint8_t>(); a.tab = c->gf_tab;
    a.in_frame_bytes = in_frame; a.out_frame_bytes = out_frame;
    a.in_pitch = SDRHIP_UDPSIZE; a.out_pitch = SDRHIP_UDPSIZE; a.in_off = 4; a.out_off = 4;
    a.rows = m; a.cols = k; a.nframes = nslab;
    hipError_t e = launch_gf_apply(a, c->stream);
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "cm256 encode launch: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(hout.data(), c->out.p, hout.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int s = 0; s < nslab; ++s) {
        const int off = s * SDRHIP_BLOCK_BYTES, len = (bb - off) < SDRHIP_BLOCK_BYTES ? (bb - off) : SDRHIP_BLOCK_BYTES;
        for (int r = 0; r < m; ++r) memcpy(static_cast<uint8_t *>(recoveryBlocks) + (size_t)r * bb + off, &hout[s * out_frame + (size_t)r * SDRHIP_UDPSIZE + 4], (size_t)len);
    }
    return SDRHIP_OK;
}

// CM256::cm256_decode on host pointers (SDRdaemonFECBuffer.cpp:197), in-place contract.
extern "C" int sdrhip_cm256_decode(sdrhip_ctx *c, sdrhip_cm256_params p, sdrhip_cm256_block *blocks)
{
    if (!c) return fail(SDRHIP_EINVAL, "ctx is NULL");
    sdrhip::CtxLock lock_(c);
    if (p.OriginalCount <= 0 || p.RecoveryCount <= 0 || p.BlockBytes <= 0) return fail(-1, "cm256_decode: invalid params");
    if (p.OriginalCount + p.RecoveryCount > 256) return fail(-2, "cm256_decode: OriginalCount + RecoveryCount > 256");
    if (!blocks) return fail(-3, "cm256_decode: NULL pointer");
    const int k = p.OriginalCount;
    if (k == 1) { blocks[0].Index = 0; return SDRHIP_OK; } // upstream: the same block repeated
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint8_t> idx(k), rec_pos(k), erased(256), coef((size_t)k * k);
    for (int i = 0; i < k; ++i) idx[i] = blocks[i].Index;
    int n_rec = 0;
    if (cm256_decode_plan(k, p.RecoveryCount, idx.data(), &n_rec, rec_pos.data(), erased.data(), coef.data()))
        return fail(SDRHIP_EDECODE, "cm256_decode: duplicate original index or singular system"); // upstream Initialize() false -> -5
    if (n_rec == 0) return SDRHIP_OK;
    const int bb = p.BlockBytes;
    const int nslab = (bb + SDRHIP_BLOCK_BYTES - 1) / SDRHIP_BLOCK_BYTES;
    const size_t in_frame = (size_t)k * SDRHIP_UDPSIZE, out_frame = (size_t)n_rec * SDRHIP_UDPSIZE;
    std::vector<uint8_t> hin(in_frame * nslab, 0), hout(out_frame * nslab);
    for (int s = 0; s < nslab; ++s) {
        const int off = s * SDRHIP_BLOCK_BYTES, len = (bb - off) < SDRHIP_BLOCK_BYTES ? (bb - off) : SDRHIP_BLOCK_BYTES;
        for (int j = 0; j < k; ++j) memcpy(&hin[s * in_frame + (size_t)j * SDRHIP_UDPSIZE + 4], static_cast<const uint8_t *>(blocks[j].Block) + off, (size_t)len);
    }
    int rc;
    if ((rc = c->in.reserve(hin.size()))) return rc;
    if ((rc = c->out.reserve(hout.size()))) return rc;
    if ((rc = c->aux.reserve((size_t)n_rec * k))) return rc;
    HIP_TRY(hipMemcpyAsync(c->in.p, hin.data(), hin.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->aux.p, coef.data(), (size_t)n_rec * k, hipMemcpyHostToDevice, c->stream));
    GfArgs a;
    memset(&a, 0, sizeof(a));
    a.in = c->in.as<uint8_t>(); a.out = c->out.as<uint8_t>(); a.coef = c->aux.as<uint8_t>(); a.tab = c->gf_tab;
    a.in_frame_bytes = in_frame; a.out_frame_bytes = out_frame;
    a.in_pitch = SDRHIP_UDPSIZE; a.out_pitch = SDRHIP_UDPSIZE; a.in_off = 4; a.out_off = 4;
    a.rows = n_rec; a.cols = k; a.nframes = nslab;
    hipError_t e = launch_gf_apply(a, c->stream);
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "cm256 decode launch: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(hout.data(), c->out.p, hout.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n_rec; ++i) {
        sdrhip_cm256_block &b = blocks[rec_pos[i]];
        for (int s = 0; s < nslab; ++s) {
            const int off = s * SDRHIP_BLOCK_BYTES, len = (bb - off) < SDRHIP_BLOCK_BYTES ? (bb - off) : SDRHIP_BLOCK_BYTES;
            memcpy(static_cast<uint8_t *>(b.Block) + off, &hout[s * out_frame + (size_t)i * SDRHIP_UDPSIZE + 4], (size_t)len);
        }
        b.Index = erased[i];
    }
    return SDRHIP_OK;
}
