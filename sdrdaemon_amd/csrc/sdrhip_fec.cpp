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
// 32-row Karatsuba tile (13.7 k lane-ops per column vs 36 k x rows / 32)
constexpr int ENC128_MIN_ROWS = 13;

int fec_encode_device(sdrhip_ctx *c, const uint8_t *frames, size_t frame_bytes, size_t nframes, int nb_fec, uint8_t *rec,
                      size_t rec_frame_bytes, const int32_t *frame_list_dev, int ngroups)
{
    if (nframes == 0 || nb_fec <= 0) return SDRHIP_OK;
    hipError_t e;
    if (nb_fec >= ENC128_MIN_ROWS && rec_frame_bytes % 4 == 0 && frame_bytes % 4 == 0) {
        // structured encoder (Karatsuba over the XOR-convolution form of the Cauchy rows); frame lists of
        // the generic kernel come in groups of GF_FRAMES_PER_GROUP, this kernel takes them flat
        Enc128Args k;
        memset(&k, 0, sizeof(k));
        k.in = frames; k.out = rec; k.tab = c->gf_tab; k.leaf_tables = c->enc_leaves;
        k.in_frame_bytes = frame_bytes; k.out_frame_bytes = rec_frame_bytes;
        k.rows = nb_fec; k.nframes = (int)nframes;
        k.frame_list = frame_list_dev; k.nlist = frame_list_dev ? ngroups * GF_FRAMES_PER_GROUP : (int)nframes;
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

// one batch with at most DEC_SLOTS distinct erasure patterns (fec_decode_device below splits longer ones)
static int fec_decode_chunk(sdrhip_ctx *c, const uint8_t *rx, size_t rx_frame_bytes, const uint8_t *indices, size_t nframes,
                            uint8_t *payload_out, size_t payload_frame_bytes, uint8_t *block0_out)
{
    if (nframes == 0) return SDRHIP_OK;
    const int K = SDRHIP_NB_ORIGINAL;
    constexpr int SLOTS = sdrhip_ctx::DEC_SLOTS;
    int rc;
    if (!c->dec_coef) {
        if (hipMalloc(reinterpret_cast<void **>(&c->dec_coef), (size_t)SLOTS * K * K) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&c->dec_dst), (size_t)2 * SLOTS * K * sizeof(int16_t)) != hipSuccess)
            return fail(SDRHIP_ENOMEM, "hipMalloc decode-plan cache");
    }
    // ---- host planning.  Per frame: scatter maps for the received originals; frames that used
    // recovery blocks (SDRdaemonFECBuffer.cpp:174) get the matrix slot of their erasure pattern, derived
    // once per distinct pattern (restating CM256Decoder::Initialize/Decode) and cached on the device.
    // staging layout: pmap[nframes*K] i16 | zmap[nframes*K] i16 | flist[2*nframes] i32 | gcm[nframes] i32 | new slots
    const size_t nmap = nframes * (size_t)K;
    const size_t o_pmap = 0, o_zmap = o_pmap + nmap * 2, o_flist = (o_zmap + nmap * 2 + 15) & ~(size_t)15;
    const size_t o_gcm = o_flist + nframes * GF_FRAMES_PER_GROUP * 4, o_new = (o_gcm + nframes * 4 + 15) & ~(size_t)15;
    const size_t slot_bytes = (size_t)K * K + 2 * K * sizeof(int16_t);
    if ((rc = c->pin.reserve(o_new + (size_t)SLOTS * slot_bytes))) return rc;
    if ((rc = c->aux.reserve(o_new))) return rc;
    uint8_t *hp = c->pin.as<uint8_t>();
    int16_t *pmap = reinterpret_cast<int16_t *>(hp + o_pmap), *zmap = reinterpret_cast<int16_t *>(hp + o_zmap);
    int32_t *flist = reinterpret_cast<int32_t *>(hp + o_flist), *gcm = reinterpret_cast<int32_t *>(hp + o_gcm);

    std::vector<std::vector<int32_t>> members; // frames per slot used by this call
    std::vector<int> used_slots;
    std::map<int, int> member_of;              // slot -> index in members
    std::vector<int> new_slots;
    std::vector<uint8_t> coef((size_t)K * K), rec_pos(K), erased(256);
    int max_rows = 0;
    bool any_b0 = false;
    std::vector<size_t> holes; // frames that will not be written completely: their output is zeroed first
    for (size_t f = 0; f < nframes; ++f) {
        const uint8_t *idx = indices + f * K;
        int n_recovery = 0;
        uint64_t have[2] = {0, 0};
        for (int p = 0; p < K; ++p) {
            const int b = idx[p];
            pmap[f * K + p] = (b >= 1 && b < K) ? (int16_t)(b - 1) : (int16_t)-1; // payload slot of block b
            zmap[f * K + p] = b == 0 ? (int16_t)0 : (int16_t)-1;
            if (b >= K) ++n_recovery;
            else have[b >> 6] |= (uint64_t)1 << (b & 63);
        }
        if (n_recovery == 0) {
            if (~have[0] || ~have[1]) holes.push_back(f); // 128 originals with repeats: some block never arrived
            continue;
        }
        std::string key(reinterpret_cast<const char *>(idx), K);
        auto it = c->dec_slot_of.find(key);
        if (it == c->dec_slot_of.end()) {
            // a slot for the new pattern: a free one, a fresh one, or evict the patterns this call does not use
            int slot = -1;
            if (!c->dec_free.empty()) {
                slot = c->dec_free.back();
                c->dec_free.pop_back();
            } else if ((int)c->dec_nrec.size() < SLOTS) {
                slot = (int)c->dec_nrec.size();
                c->dec_nrec.push_back(0);
                c->dec_b0.push_back(0);
            } else {
                std::vector<char> busy(SLOTS, 0);
                for (size_t u = 0; u < used_slots.size(); ++u) busy[used_slots[u]] = 1;
                for (size_t u = 0; u < new_slots.size(); ++u) busy[new_slots[u]] = 1;
                for (auto jt = c->dec_slot_of.begin(); jt != c->dec_slot_of.end();) {
                    if (jt->second >= 0 && !busy[jt->second]) { c->dec_free.push_back(jt->second); jt = c->dec_slot_of.erase(jt); }
                    else ++jt;
                }
                if (c->dec_free.empty()) return fail(SDRHIP_EINVAL, "internal: decode batch not split at %d erasure patterns", SLOTS);
                slot = c->dec_free.back();
                c->dec_free.pop_back();
            }
            int n_rec = 0;
            // the reference passes the number of RECEIVED recovery blocks as RecoveryCount (:176)
            if (cm256_decode_plan(K, n_recovery, idx, &n_rec, rec_pos.data(), erased.data(), coef.data())) {
                c->dec_free.push_back(slot);
                // undecodable patterns (e.g. duplicated datagrams) are remembered too, but not for ever: a receiver fed
                // from the network must not grow by one key per bad pattern
                if (c->dec_slot_of.size() > 4 * (size_t)SLOTS)
                    for (auto jt = c->dec_slot_of.begin(); jt != c->dec_slot_of.end();) jt = jt->second < 0 ? c->dec_slot_of.erase(jt) : std::next(jt);
                c->dec_slot_of[key] = -1; // "CM256 decode error" (:199): the frame keeps what was received
                holes.push_back(f);
                continue;
            }
            c->dec_nrec[slot] = n_rec;
            c->dec_b0[slot] = 0;
            it = c->dec_slot_of.insert(std::make_pair(key, slot)).first;
            uint8_t *sp = hp + o_new + new_slots.size() * slot_bytes;
            memset(sp, 0, slot_bytes);
            memcpy(sp, coef.data(), (size_t)n_rec * K);
            int16_t *dp = reinterpret_cast<int16_t *>(sp + (size_t)K * K), *dz = dp + K;
            for (int i = 0; i < K; ++i) { dp[i] = -1; dz[i] = -1; }
            for (int i = 0; i < n_rec; ++i) {
                if (erased[i] >= 1 && erased[i] < K) dp[i] = (int16_t)(erased[i] - 1);
                if (erased[i] == 0) { dz[i] = 0; c->dec_b0[slot] = 1; }
            }
            new_slots.push_back(slot);
        }
        const int slot = it->second;
        if (slot < 0) { if (holes.empty() || holes.back() != f) holes.push_back(f); continue; }
        auto mo = member_of.find(slot);
        if (mo == member_of.end()) {
            mo = member_of.insert(std::make_pair(slot, (int)members.size())).first;
            members.push_back(std::vector<int32_t>());
            used_slots.push_back(slot);
            if (c->dec_nrec[slot] > max_rows) max_rows = c->dec_nrec[slot];
            any_b0 |= c->dec_b0[slot] != 0;
        }
        members[mo->second].push_back((int32_t)f);
    }
    int ngroups = 0;
    for (size_t u = 0; u < members.size(); ++u)
        for (size_t i = 0; i < members[u].size(); i += GF_FRAMES_PER_GROUP) {
            for (size_t v = 0; v < (size_t)GF_FRAMES_PER_GROUP; ++v) flist[ngroups * GF_FRAMES_PER_GROUP + v] = i + v < members[u].size() ? members[u][i + v] : -1;
            gcm[ngroups++] = used_slots[u];
        }
    // ---- uploads (pinned, asynchronous): maps + lists in one copy, new matrices into their slots
    uint8_t *dv = c->aux.as<uint8_t>();
    HIP_TRY(hipMemcpyAsync(dv, hp, o_new, hipMemcpyHostToDevice, c->stream));
    for (size_t i = 0; i < new_slots.size(); ++i) {
        const uint8_t *sp = hp + o_new + i * slot_bytes;
        const int slot = new_slots[i];
        HIP_TRY(hipMemcpyAsync(c->dec_coef + (size_t)slot * K * K, sp, (size_t)K * K, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->dec_dst + (size_t)slot * K, sp + (size_t)K * K, K * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->dec_dst + (size_t)(SLOTS + slot) * K, sp + (size_t)K * K + K * sizeof(int16_t), K * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
    }
    c->pin.mark(c->stream);

    // ---- payload = zeros (initDecodeSlot memset, :109) + received originals in place.  A frame whose 127
    // payload blocks all get written (received or recovered) needs no zero fill: only the undecodable ones do.
    if (holes.size() > 32) {
        HIP_TRY(hipMemsetAsync(payload_out, 0, nframes * payload_frame_bytes, c->stream));
    } else {
        for (size_t h = 0; h < holes.size(); ++h)
            HIP_TRY(hipMemsetAsync(payload_out + holes[h] * payload_frame_bytes, 0, (size_t)(K - 1) * SDRHIP_BLOCK_BYTES, c->stream));
    }
    hipError_t e = launch_block_scatter(rx, rx_frame_bytes, SDRHIP_UDPSIZE, 4, payload_out, payload_frame_bytes, SDRHIP_BLOCK_BYTES, 0,
                                        reinterpret_cast<const int16_t *>(dv + o_pmap), K, (int)nframes, c->stream);
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "scatter launch: %s", hipGetErrorString(e));
    if (block0_out) {
        if (holes.size() > 32) {
            HIP_TRY(hipMemsetAsync(block0_out, 0, nframes * (size_t)SDRHIP_BLOCK_BYTES, c->stream));
        } else {
            for (size_t h = 0; h < holes.size(); ++h)
                HIP_TRY(hipMemsetAsync(block0_out + holes[h] * (size_t)SDRHIP_BLOCK_BYTES, 0, SDRHIP_BLOCK_BYTES, c->stream));
        }
        e = launch_block_scatter(rx, rx_frame_bytes, SDRHIP_UDPSIZE, 4, block0_out, SDRHIP_BLOCK_BYTES, SDRHIP_BLOCK_BYTES, 0,
                                 reinterpret_cast<const int16_t *>(dv + o_zmap), K, (int)nframes, c->stream);
        if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "scatter launch: %s", hipGetErrorString(e));
    }
    if (ngroups == 0) return SDRHIP_OK;

    GfArgs a;
    memset(&a, 0, sizeof(a));
    a.in = rx; a.out = payload_out; a.coef = c->dec_coef; a.tab = c->gf_tab;
    a.in_frame_bytes = rx_frame_bytes; a.out_frame_bytes = payload_frame_bytes;
    a.in_pitch = SDRHIP_UDPSIZE; a.in_off = 4; a.out_pitch = SDRHIP_BLOCK_BYTES; a.out_off = 0;
    a.rows = max_rows; a.cols = K; a.matrix_rows = K;
    a.row_dst = c->dec_dst;
    a.nframes = (int)nframes;
    a.frame_list = reinterpret_cast<const int32_t *>(dv + o_flist);
    a.group_cm = reinterpret_cast<const int32_t *>(dv + o_gcm);
    a.ngroups = ngroups;
    {
        KTimer kt(c, SDRHIP_K_FEC_DECODE);
        e = launch_gf_apply(a, c->stream);
    }
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec decode launch: %s", hipGetErrorString(e));
    if (block0_out && any_b0) {
        a.out = block0_out; a.out_frame_bytes = SDRHIP_BLOCK_BYTES;
        a.row_dst = c->dec_dst + (size_t)SLOTS * K;
        e = launch_gf_apply(a, c->stream);
        if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "fec decode launch: %s", hipGetErrorString(e));
    }
    return SDRHIP_OK;
}

int fec_decode_device(sdrhip_ctx *c, const uint8_t *rx, size_t rx_frame_bytes, const uint8_t *indices, size_t nframes,
                      uint8_t *payload_out, size_t payload_frame_bytes, uint8_t *block0_out)
{
    // The plan cache holds DEC_SLOTS matrices: a batch with more distinct erasure patterns than that goes
    // through in consecutive chunks (stream order keeps a slot's matrix alive until the chunk that used it
    // has run).  The common case -- a handful of loss patterns per batch -- is one chunk.
    const int K = SDRHIP_NB_ORIGINAL;
    size_t f0 = 0;
    while (f0 < nframes) {
        std::map<std::string, char> seen;
        size_t f1 = f0;
        for (; f1 < nframes; ++f1) {
            const uint8_t *idx = indices + f1 * K;
            bool uses_recovery = false;
            for (int p = 0; p < K && !uses_recovery; ++p) uses_recovery = idx[p] >= K;
            if (!uses_recovery) continue;
            std::string key(reinterpret_cast<const char *>(idx), K);
            if (seen.count(key)) continue;
            if ((int)seen.size() == sdrhip_ctx::DEC_SLOTS) break;
            seen[key] = 1;
        }
        const int rc = fec_decode_chunk(c, rx + f0 * rx_frame_bytes, rx_frame_bytes, indices + f0 * K, f1 - f0,
                                        payload_out + f0 * payload_frame_bytes, payload_frame_bytes,
                                        block0_out ? block0_out + f0 * (size_t)SDRHIP_BLOCK_BYTES : nullptr);
        if (rc) return rc;
        f0 = f1;
    }
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
