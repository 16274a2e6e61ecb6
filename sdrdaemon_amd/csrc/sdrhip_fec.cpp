// sdrhip_fec.cpp -- host side of the CM256 entry points (include/sdrhip.h): planning on the
// host (gf256.cpp), block arithmetic on the GPU (gf_kernels.hip).
#include "gf256.h"
#include "sdrhip_host.h"

#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace sdrhip;

namespace sdrhip {

int fec_encode_device(sdrhip_ctx *c, const uint8_t *frames, size_t frame_bytes, size_t nframes, int nb_fec, uint8_t *rec,
                      size_t rec_frame_bytes, const int32_t *frame_list_dev, int ngroups)
{
    if (nframes == 0 || nb_fec <= 0) return SDRHIP_OK;
    GfArgs a;
    memset(&a, 0, sizeof(a));
    a.in = frames; a.out = rec; a.coef = c->enc_matrix; a.tab = c->gf_tab;
    a.in_frame_bytes = frame_bytes; a.out_frame_bytes = rec_frame_bytes;
    a.in_pitch = SDRHIP_UDPSIZE; a.out_pitch = SDRHIP_UDPSIZE; a.in_off = 4; a.out_off = 4;
    a.rows = nb_fec; a.cols = SDRHIP_NB_ORIGINAL; a.coef_per_frame = 0;
    a.nframes = (int)nframes;
    a.frame_list = frame_list_dev; a.ngroups = ngroups;
    hipError_t e;
    {
        KTimer kt(c, SDRHIP_K_FEC_ENCODE);
        e = launch_gf_apply(a, c->stream);
    }
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec encode launch: %s", hipGetErrorString(e));
    // headers {frameIndex, 128 + r, filler 0} of the recovery super blocks (UDPSinkFEC.cpp:239-243)
    e = launch_fec_headers(frames, frame_bytes, rec, rec_frame_bytes, nb_fec, SDRHIP_NB_ORIGINAL, (int)nframes, c->stream);
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec header launch: %s", hipGetErrorString(e));
    return SDRHIP_OK;
}

int fec_decode_device(sdrhip_ctx *c, const uint8_t *rx, size_t rx_frame_bytes, const uint8_t *indices, size_t nframes,
                      uint8_t *payload_out, size_t payload_frame_bytes, uint8_t *block0_out)
{
    if (nframes == 0) return SDRHIP_OK;
    const int K = SDRHIP_NB_ORIGINAL;
    // ---- host planning: scatter map for received originals + one decode matrix per distinct
    // erasure pattern (frames with the same 128 indices share it)
    std::vector<int16_t> map(nframes * (size_t)K);
    std::map<std::string, int> pattern_of;
    std::vector<std::vector<int32_t>> members;     // frames per pattern
    std::vector<std::vector<uint8_t>> coefs;       // n_rec x K per pattern
    std::vector<std::vector<int16_t>> dsts;        // n_rec destination block (-1 skip)
    std::vector<uint8_t> coef((size_t)K * K), rec_pos(K), erased(256);
    int max_rows = 0;
    for (size_t f = 0; f < nframes; ++f) {
        const uint8_t *idx = indices + f * K;
        int n_recovery = 0;
        for (int p = 0; p < K; ++p) {
            map[f * K + p] = idx[p] < K ? (int16_t)idx[p] : (int16_t)-1;
            if (idx[p] >= K) ++n_recovery;
        }
        if (n_recovery == 0) continue; // SDRdaemonFECBuffer.cpp:174: decode only if recovery blocks were used
        std::string key(reinterpret_cast<const char *>(idx), K);
        auto it = pattern_of.find(key);
        if (it == pattern_of.end()) {
            int n_rec = 0;
            // the reference passes the number of RECEIVED recovery blocks as RecoveryCount (:176)
            int rc = cm256_decode_plan(K, n_recovery, idx, &n_rec, rec_pos.data(), erased.data(), coef.data());
            if (rc) {
                // "CM256 decode error" (:199): the frame keeps what was received
                pattern_of[key] = -1;
                continue;
            }
            std::vector<int16_t> d(n_rec);
            for (int i = 0; i < n_rec; ++i) d[i] = (int16_t)erased[i];
            it = pattern_of.insert(std::make_pair(key, (int)coefs.size())).first;
            coefs.push_back(std::vector<uint8_t>(coef.begin(), coef.begin() + (size_t)n_rec * K));
            dsts.push_back(d);
            members.push_back(std::vector<int32_t>());
            if (n_rec > max_rows) max_rows = n_rec;
        }
        if (it->second >= 0) members[it->second].push_back((int32_t)f);
    }
    // ---- payload = zeros (initDecodeSlot memset, :109) + received originals in place
    // destination frame layout: block b (0..127) at dst_base + b * 508 where block 0 goes to
    // block0_out (or nowhere) and blocks 1.. to payload_out: use two scatters via a map with -1
    int rc;
    const size_t map_bytes = map.size() * sizeof(int16_t);
    if ((rc = c->aux.reserve(map_bytes))) return rc;
    HIP_TRY(hipMemsetAsync(payload_out, 0, nframes * payload_frame_bytes, c->stream));
    // payload scatter: block index i >= 1 -> slot i - 1
    std::vector<int16_t> pmap(map.size()), zmap(map.size());
    for (size_t i = 0; i < map.size(); ++i) {
        pmap[i] = map[i] >= 1 ? (int16_t)(map[i] - 1) : (int16_t)-1;
        zmap[i] = map[i] == 0 ? (int16_t)0 : (int16_t)-1;
    }
    HIP_TRY(hipMemcpyAsync(c->aux.p, pmap.data(), map_bytes, hipMemcpyHostToDevice, c->stream));
    hipError_t e = launch_block_scatter(rx, rx_frame_bytes, SDRHIP_UDPSIZE, 4, payload_out, payload_frame_bytes, SDRHIP_BLOCK_BYTES, 0,
                                        c->aux.as<int16_t>(), K, (int)nframes, c->stream);
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "scatter launch: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(c->stream)); // pmap is a host temporary
    if (block0_out) {
        HIP_TRY(hipMemsetAsync(block0_out, 0, nframes * (size_t)SDRHIP_BLOCK_BYTES, c->stream));
        HIP_TRY(hipMemcpyAsync(c->aux.p, zmap.data(), map_bytes, hipMemcpyHostToDevice, c->stream));
        e = launch_block_scatter(rx, rx_frame_bytes, SDRHIP_UDPSIZE, 4, block0_out, SDRHIP_BLOCK_BYTES, SDRHIP_BLOCK_BYTES, 0,
                                 c->aux.as<int16_t>(), K, (int)nframes, c->stream);
        if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "scatter launch: %s", hipGetErrorString(e));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (coefs.empty()) return SDRHIP_OK;

    // ---- groups of GF_FRAMES_PER_GROUP frames sharing a pattern
    std::vector<int32_t> frame_list;
    std::vector<int> group_pattern;
    for (size_t pt = 0; pt < members.size(); ++pt)
        for (size_t i = 0; i < members[pt].size(); i += GF_FRAMES_PER_GROUP) {
            for (size_t u = 0; u < (size_t)GF_FRAMES_PER_GROUP; ++u) frame_list.push_back(i + u < members[pt].size() ? members[pt][i + u] : -1);
            group_pattern.push_back((int)pt);
        }
    const int ngroups = (int)group_pattern.size();
    const int rows = max_rows;
    std::vector<uint8_t> gcoef((size_t)ngroups * rows * K, 0);
    std::vector<int16_t> gdst_payload((size_t)ngroups * rows, -1), gdst_b0((size_t)ngroups * rows, -1);
    bool any_b0 = false;
    for (int g = 0; g < ngroups; ++g) {
        const std::vector<uint8_t> &cf = coefs[group_pattern[g]];
        const std::vector<int16_t> &d = dsts[group_pattern[g]];
        memcpy(&gcoef[(size_t)g * rows * K], cf.data(), cf.size());
        for (size_t i = 0; i < d.size(); ++i) {
            if (d[i] >= 1 && d[i] < K) gdst_payload[(size_t)g * rows + i] = (int16_t)(d[i] - 1);
            if (d[i] == 0) { gdst_b0[(size_t)g * rows + i] = 0; any_b0 = true; }
        }
    }
    const size_t o_coef = 0, o_list = (gcoef.size() + 15) & ~(size_t)15, o_dst = o_list + ((frame_list.size() * 4 + 15) & ~(size_t)15);
    const size_t o_dst0 = o_dst + ((gdst_payload.size() * 2 + 15) & ~(size_t)15);
    const size_t total = o_dst0 + gdst_b0.size() * 2 + 16;
    if ((rc = c->aux2.reserve(total))) return rc;
    uint8_t *base = c->aux2.as<uint8_t>();
    HIP_TRY(hipMemcpyAsync(base + o_coef, gcoef.data(), gcoef.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(base + o_list, frame_list.data(), frame_list.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(base + o_dst, gdst_payload.data(), gdst_payload.size() * 2, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(base + o_dst0, gdst_b0.data(), gdst_b0.size() * 2, hipMemcpyHostToDevice, c->stream));
    GfArgs a;
    memset(&a, 0, sizeof(a));
    a.in = rx; a.out = payload_out; a.coef = base + o_coef; a.tab = c->gf_tab;
    a.in_frame_bytes = rx_frame_bytes; a.out_frame_bytes = payload_frame_bytes;
    a.in_pitch = SDRHIP_UDPSIZE; a.in_off = 4; a.out_pitch = SDRHIP_BLOCK_BYTES; a.out_off = 0;
    a.rows = rows; a.cols = K; a.coef_per_frame = 1;
    a.row_dst = reinterpret_cast<const int16_t *>(base + o_dst);
    a.nframes = (int)nframes;
    a.frame_list = reinterpret_cast<const int32_t *>(base + o_list);
    a.ngroups = ngroups;
    {
        KTimer kt(c, SDRHIP_K_FEC_DECODE);
        e = launch_gf_apply(a, c->stream);
    }
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec decode launch: %s", hipGetErrorString(e));
    if (block0_out && any_b0) {
        a.out = block0_out; a.out_frame_bytes = SDRHIP_BLOCK_BYTES;
        a.row_dst = reinterpret_cast<const int16_t *>(base + o_dst0);
        e = launch_gf_apply(a, c->stream);
        if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec decode launch: %s", hipGetErrorString(e));
    }
    HIP_TRY(hipStreamSynchronize(c->stream)); // host temporaries were the copy sources
    return SDRHIP_OK;
}

} // namespace sdrhip

// ---------------------------------------------------------------------------------- C ABI
extern "C" int sdrhip_fec_encode_frames(sdrhip_ctx *c, const uint8_t *frames, size_t nframes, int nb_fec, uint8_t *recovery_out,
                                        int mem)
{
    if (!c) return fail(SDRHIP_EINVAL, "ctx is NULL");
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
    if (nframes == 0) return SDRHIP_OK;
    if (!rx || !payload_out) return fail(SDRHIP_EINVAL, "fec_decode_frames: NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t fb = (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE, pb = (size_t)127 * SDRHIP_BLOCK_BYTES;
    if (mem == SDRHIP_MEM_DEVICE) {
        if (!indices) return fail(SDRHIP_EINVAL, "fec_decode_frames: device mode needs the host `indices` array");
        return fec_decode_device(c, rx, fb, indices, nframes, payload_out, pb, block0_out);
    }
    if (mem != SDRHIP_MEM_HOST) return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    std::vector<uint8_t> idx;
    if (!indices) { // header.blockIndex of every received super block (SDRdaemonFECBuffer.cpp:147)
        idx.resize(nframes * SDRHIP_NB_ORIGINAL);
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = rx[i * SDRHIP_UDPSIZE + 2];
        indices = idx.data();
    }
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
