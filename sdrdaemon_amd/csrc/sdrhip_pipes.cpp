// sdrhip_pipes.cpp -- interpolator bank and the fused Rx / Tx pipes of include/sdrhip.h.
#include "gf256.h"
#include "sdrhip_host.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace sdrhip;

// --------------------------------------------------------------------------- interpolators
struct sdrhip_interpolators {
    sdrhip_ctx *ctx;
    int nstreams;
    int32_t *state[2]; // [nstreams][INT_STATE_WORDS]
    int cur;
};

extern "C" int sdrhip_interpolators_create(sdrhip_ctx *ctx, int nstreams, sdrhip_interpolators **out)
{
    if (!ctx || !out || nstreams <= 0 || nstreams > 65535) return fail(SDRHIP_EINVAL, "interpolators_create: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    sdrhip_interpolators *p = new (std::nothrow) sdrhip_interpolators();
    if (!p) return fail(SDRHIP_ENOMEM, "out of host memory");
    p->ctx = ctx; p->nstreams = nstreams; p->cur = 0;
    size_t bytes = (size_t)nstreams * INT_STATE_WORDS * sizeof(int32_t);
    p->state[0] = p->state[1] = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p->state[0]), bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&p->state[1]), bytes) != hipSuccess) {
        if (p->state[0]) (void)hipFree(p->state[0]);
        delete p;
        return fail(SDRHIP_ENOMEM, "hipMalloc interpolator state");
    }
    ctx_retain(ctx);
    *out = p;
    return sdrhip_interpolators_reset(p);
}

extern "C" void sdrhip_interpolators_destroy(sdrhip_interpolators *p)
{
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    (void)hipFree(p->state[0]);
    (void)hipFree(p->state[1]);
    ctx_release(p->ctx);
    delete p;
}

extern "C" int sdrhip_interpolators_reset(sdrhip_interpolators *p)
{
    if (!p) return fail(SDRHIP_EINVAL, "interpolators is NULL");
    sdrhip::CtxLock lock_(p->ctx);
    size_t bytes = (size_t)p->nstreams * INT_STATE_WORDS * sizeof(int32_t);
    HIP_TRY(hipMemsetAsync(p->state[0], 0, bytes, p->ctx->stream));
    HIP_TRY(hipMemsetAsync(p->state[1], 0, bytes, p->ctx->stream));
    p->cur = 0;
    return SDRHIP_OK;
}

namespace sdrhip {
bool interpolate_gather_ok(const sdrhip_ctx *c, int log2interp) { return c->opt.interp_wave && log2interp >= 2; }

int interpolate_device(sdrhip_interpolators *p, int log2interp, const int16_t *in, size_t n_in, size_t in_stride, int16_t *out,
                       size_t out_stride, size_t *n_out, const InterpGather *gather)
{
    sdrhip_ctx *c = p->ctx;
    if (n_out) *n_out = n_in << log2interp;
    if (n_in == 0) return SDRHIP_OK;
    if (gather && !interpolate_gather_ok(p->ctx, log2interp)) return fail(SDRHIP_EINVAL, "internal: gathered input needs the wave interpolator");
    if (log2interp == 0) { // Upsampler::process m_interp == 0: samples_out = samples_in (Upsampler.cpp:54-57)
        HIP_TRY(hipMemcpy2DAsync(out, out_stride * 4, in, in_stride * 4, n_in * 4, p->nstreams, hipMemcpyDeviceToDevice, c->stream));
        return SDRHIP_OK;
    }
    InterpArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.in_stride = in_stride; a.out_stride = out_stride; a.n_in = n_in;
    a.state_cur = p->state[p->cur]; a.state_next = p->state[p->cur ^ 1];
    a.nstreams = p->nstreams;
    if (gather) { a.gmap = gather->map; a.grx = gather->rx; a.grest = gather->restored; a.gframes = gather->frames; }
    // SDRHIP_INTERP_PATH = wave (K5w, default) | valu (K5); SDRHIP_INTERP_SPAN = segment length in inputs (tests)
    const bool use_wave = c->opt.interp_wave && log2interp >= 2;
    if (use_wave) plan_interpolate_wave(log2interp, n_in, p->nstreams, c->n_cu, c->opt.interp_span, &a.nsub_per_seg, &a.nseg);
    else plan_interpolate(log2interp, n_in, p->nstreams, &a.nsub_per_seg, &a.nseg);
    hipError_t e;
    {
        KTimer kt(c, SDRHIP_K_INTERPOLATE);
        e = use_wave ? launch_interpolate_wave(log2interp, a, c->stream) : launch_interpolate(log2interp, a, c->stream);
    }
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "interpolate launch: %s", hipGetErrorString(e));
    p->cur ^= 1;
    return SDRHIP_OK;
}
} // namespace sdrhip

extern "C" int sdrhip_interpolate(sdrhip_interpolators *p, int log2interp, const int16_t *iq_in, size_t n_in, size_t in_stride,
                                  int16_t *iq_out, size_t out_stride, size_t *n_out, int mem)
{
    if (!p) return fail(SDRHIP_EINVAL, "interpolate: NULL handle");
    sdrhip::CtxLock lock_(p->ctx);
    if (log2interp < 0 || log2interp > 6) return fail(SDRHIP_EINVAL, "Invalid log2 interpolation factor"); // Upsampler.cpp:38-42
    if (n_in && (!iq_in || !iq_out)) return fail(SDRHIP_EINVAL, "interpolate: NULL buffer");
    sdrhip_ctx *c = p->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int S = p->nstreams;
    const size_t n_res = n_in << log2interp;
    if (S == 1) { in_stride = n_in; out_stride = n_res; }
    if (S > 1 && (in_stride < n_in || out_stride < n_res)) return fail(SDRHIP_EINVAL, "interpolate: stride smaller than the per-stream length");
    if (mem == SDRHIP_MEM_DEVICE) {
        if (n_in && (!aligned16(iq_in) || !aligned16(iq_out) || (S > 1 && ((in_stride & 3) || (out_stride & 3)))))
            return fail(SDRHIP_EALIGN, "interpolate: device pointers must be 16-byte aligned and strides multiples of 4 samples");
        return interpolate_device(p, log2interp, iq_in, n_in, in_stride, iq_out, out_stride, n_out);
    }
    if (mem != SDRHIP_MEM_HOST) return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    if (n_in == 0) { if (n_out) *n_out = 0; return SDRHIP_OK; }
    const size_t dis = (n_in + 3) & ~(size_t)3, dos = (n_res + 3) & ~(size_t)3;
    int rc;
    if ((rc = c->in.reserve((size_t)S * dis * 4 + 16))) return rc;
    if ((rc = c->out.reserve((size_t)S * dos * 4 + 16))) return rc;
    HIP_TRY(hipMemcpy2DAsync(c->in.p, dis * 4, iq_in, in_stride * 4, n_in * 4, S, hipMemcpyHostToDevice, c->stream));
    if ((rc = interpolate_device(p, log2interp, c->in.as<int16_t>(), n_in, dis, c->out.as<int16_t>(), dos, n_out))) return rc;
    HIP_TRY(hipMemcpy2DAsync(iq_out, out_stride * 4, c->out.p, dos * 4, n_res * 4, S, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SDRHIP_OK;
}

// --------------------------------------------------------------------------- fused Rx pipe
struct sdrhip_rx {
    sdrhip_ctx *ctx;
    int nstreams;
    sdrhip_rx_config cfg;
    sdrhip_decimators *dec;
    // [nstreams][cap_frames][128 + nb_fec][512].  A call fills slots base_slot .. base_slot + done of
    // every stream; slot base_slot + done (the frame still being filled) is slot base_slot of the next
    // call, so the window slides and nothing is copied until it reaches the end of the area.
    DevBuf work;
    size_t cap_frames;        // frame slots per stream in `work`
    size_t base_slot;         // slot of the frame being filled
    uint64_t pending_samples; // decimated samples sitting in that slot (the partial frame)
    bool frame_open;          // it has its meta block (a frame was started)
    uint16_t frame_count;     // its m_frameCount
    // what sdrhip_rx_frames_view shows: the frames the last call DELIVERED
    const uint8_t *view_base = nullptr; // slot 0 of the delivered window of stream 0
    size_t view_stride = 0;             // bytes between streams
    size_t view_frames = 0;
    DevBuf lin[2];            // stream-order decimator output of a call that is framed by K2 (two: pipelined mode)
    int lin_sel = 0;
    DevBuf flist;             // frame list of the generic encode launch (device), relative to the window
    std::vector<int32_t> flist_host;
    size_t flist_done = 0, flist_cap = 0;
    // ---- pipelined mode (sdrhip_rx_set_pipelined): a call delivers the frames the PREVIOUS call completed; their
    // recovery blocks are computed by encoder workgroups inside this call's decimator launch (rx_fused_kernel)
    int pipelined = 0;
    struct Late {
        bool have = false;          // frames completed by the previous call wait for delivery
        bool encode = false;        // ... and still have to be encoded (k)
        Enc128Args k;
        const uint8_t *base = nullptr;
        size_t stride = 0, frames = 0, frame_bytes = 0;
        size_t slot0 = 0;           // window position inside `work` (overlap check of the sliding window), SIZE_MAX = other area
    } late;
    DevBuf old_work;          // the previous frame area after a re-allocation, kept while `late` points into it
    // overlap mode (option rx_fused = 3): the waiting encode runs on the context's second stream beside the next call's decimator.
    // ev_framed: recorded on the first stream when a call has written everything its deferred encode reads (decimator + K2);
    // ev_enc: recorded on the second stream behind the encode, the first stream waits for it before the frames are delivered
    hipEvent_t ev_framed = nullptr, ev_enc = nullptr;
    // ---- asynchronous host-pointer entry (sdrhip_rx_submit / sdrhip_rx_collect): a ring of batches
    struct Batch {
        PinnedBuf in;             // the submitted blocks, appended: [block][stream][n] (unless the caller's memory is pinned by us)
        DevBuf din;               // [stream][dstride] on the device
        PinnedBuf out;            // the batch's finished frames [stream][frames][128 + R][512]
        hipEvent_t done = nullptr;
        std::vector<std::pair<const int16_t *, size_t> > blocks; // source of each block (host address, samples per stream) and
        std::vector<size_t> strides;                             // its stream stride in samples
        size_t n_in = 0;          // samples per stream so far
        size_t in_cap = 0;        // row length of `in` in samples: staged blocks lie stream-major, [stream][in_cap], at their batch offset
        uint32_t tv_sec = 0, tv_usec = 0;
        size_t frames = 0, frame_bytes = 0;
        int state = 0;            // 0 free, 1 filling, 2 in flight
    };
    std::vector<Batch> abatch;
    int a_blocks = 1;             // blocks per launch
    bool consumed = false;        // set by sdrhip_rx_process once the decimator launch of the call went out (the filter state advanced)
    size_t a_head = 0, a_tail = 0; // next batch to collect / batch being filled
};

static int rx_check_config(const sdrhip_rx_config *cfg);

extern "C" int sdrhip_rx_create(sdrhip_ctx *ctx, int nstreams, const sdrhip_rx_config *cfg, sdrhip_rx **out)
{
    if (!ctx || !cfg || !out || nstreams <= 0) return fail(SDRHIP_EINVAL, "rx_create: bad argument");
    {
        const int rcc = rx_check_config(cfg);
        if (rcc) return rcc;
    }
    sdrhip_rx *rx = new (std::nothrow) sdrhip_rx();
    if (!rx) return fail(SDRHIP_ENOMEM, "out of host memory");
    rx->ctx = ctx; rx->nstreams = nstreams; rx->cfg = *cfg; rx->dec = nullptr;
    rx->cap_frames = 0; rx->base_slot = 0; rx->pending_samples = 0; rx->frame_open = false; rx->frame_count = 0;
    int rc = sdrhip_decimators_create(ctx, nstreams, cfg->hb_variant, &rx->dec);
    if (rc) { delete rx; return rc; }
    *out = rx;
    return SDRHIP_OK;
}

static int rx_check_config(const sdrhip_rx_config *cfg)
{
    if (cfg->log2decim < 0 || cfg->log2decim > 6) return fail(SDRHIP_EINVAL, "Invalid log2 decimation factor");
    if (cfg->fcpos < 0 || cfg->fcpos > 2) return fail(SDRHIP_EINVAL, "Invalid Fc position index");
    if (cfg->nb_fec < 0 || cfg->nb_fec > 128) return fail(SDRHIP_EINVAL, "nb_fec must be 0..128");
    if (cfg->sample_bits < 1 || cfg->sample_bits > 16) return fail(SDRHIP_EINVAL, "sample_bits must be 1..16");
    return SDRHIP_OK;
}

// the encode that a pipelined call left for the next launch, now (flush, reconfiguration, a call that cannot fuse it)
static int rx_settle(sdrhip_rx *rx)
{
    if (!rx->late.encode) return SDRHIP_OK;
    rx->late.encode = false;
    sdrhip_ctx *c = rx->ctx;
    if (c->opt.rx_fused != 3 || !rx->ev_framed) return fec_encode128_launch(c, rx->late.k);
    // overlap mode: the encode goes to the second stream -- behind everything the call that left it had enqueued (ev_framed), beside
    // whatever the first stream runs now (the decimator of the current call, enqueued just before) -- and the first stream picks
    // up behind it: the frames are delivered, and the buffers reused, in first-stream order
    hipStream_t s2 = nullptr;
    int rc = ctx_stream2(c, &s2);
    if (rc) return rc;
    if (!rx->ev_enc) HIP_TRY(hipEventCreateWithFlags(&rx->ev_enc, hipEventDisableTiming));
    HIP_TRY(hipStreamWaitEvent(s2, rx->ev_framed, 0));
    if ((rc = fec_encode128_launch(c, rx->late.k, s2))) return rc;
    HIP_TRY(hipEventRecord(rx->ev_enc, s2));
    HIP_TRY(hipStreamWaitEvent(c->stream, rx->ev_enc, 0));
    return SDRHIP_OK;
}

extern "C" int sdrhip_rx_set_pipelined(sdrhip_rx *rx, int on)
{
    if (!rx) return fail(SDRHIP_EINVAL, "rx is NULL");
    sdrhip::CtxLock lock_(rx->ctx);
    if (!on && rx->late.have) return fail(SDRHIP_EINVAL, "rx_set_pipelined: sdrhip_rx_flush the waiting frames first");
    rx->pipelined = on ? 1 : 0;
    return SDRHIP_OK;
}

extern "C" int sdrhip_rx_reconfigure(sdrhip_rx *rx, const sdrhip_rx_config *cfg)
{
    if (!rx || !cfg) return fail(SDRHIP_EINVAL, "rx_reconfigure: NULL argument");
    sdrhip::CtxLock lock_(rx->ctx);
    int rc = rx_check_config(cfg);
    if (rc) return rc;
    if (cfg->hb_variant != rx->cfg.hb_variant) return fail(SDRHIP_EINVAL, "rx_reconfigure: hb_variant is fixed at creation");
    sdrhip_ctx *c = rx->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (cfg->nb_fec != rx->cfg.nb_fec && rx->late.have)
        return fail(SDRHIP_EINVAL, "rx_reconfigure: frames of the previous call wait for delivery (pipelined mode): sdrhip_rx_flush them "
                                   "before changing fecblk (they carry the old frame size)");
    if (cfg->nb_fec != rx->cfg.nb_fec && rx->cap_frames) {
        // the slots change size: the frame being filled (its 128 original super blocks) moves to slot 0 of a new area
        if ((rc = rx_settle(rx))) return rc;
        const int S = rx->nstreams;
        const size_t old_fb = (size_t)(SDRHIP_NB_ORIGINAL + rx->cfg.nb_fec) * SDRHIP_UDPSIZE;
        const size_t new_fb = (size_t)(SDRHIP_NB_ORIGINAL + cfg->nb_fec) * SDRHIP_UDPSIZE;
        DevBuf fresh;
        if ((rc = fresh.reserve((size_t)S * rx->cap_frames * new_fb))) return rc;
        if (rx->frame_open)
            HIP_TRY(hipMemcpy2DAsync(fresh.p, rx->cap_frames * new_fb, rx->work.as<uint8_t>() + rx->base_slot * old_fb,
                                     rx->cap_frames * old_fb, (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE, S, hipMemcpyDeviceToDevice,
                                     c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream)); // earlier launches may still use the old area
        // (no frames wait for delivery here: a fecblk change with late.have set was refused above)
        rx->old_work.release();
        rx->work.release();
        rx->work = fresh;
        rx->base_slot = 0;
        rx->view_base = nullptr;
        rx->view_frames = 0;
    }
    rx->cfg = *cfg;
    return SDRHIP_OK;
}

extern "C" void sdrhip_rx_destroy(sdrhip_rx *rx)
{
    if (!rx) return;
    sdrhip_decimators_destroy(rx->dec);
    rx->work.release();
    rx->old_work.release();
    rx->lin[0].release();
    rx->lin[1].release();
    rx->flist.release();
    if (rx->ev_framed) (void)hipEventDestroy(rx->ev_framed);
    if (rx->ev_enc) (void)hipEventDestroy(rx->ev_enc);
    for (auto &b : rx->abatch) {
        if (b.done) { (void)hipEventSynchronize(b.done); (void)hipEventDestroy(b.done); }
        b.in.release(); b.din.release(); b.out.release();
    }
    delete rx;
}

extern "C" int sdrhip_rx_frames_view(const sdrhip_rx *rx, const uint8_t **base, size_t *stream_stride_bytes, size_t *n_frames)
{
    if (!rx || !base || !stream_stride_bytes || !n_frames) return fail(SDRHIP_EINVAL, "rx_frames_view: NULL argument");
    sdrhip::CtxLock lock_(rx->ctx);
    *base = rx->view_base;
    *stream_stride_bytes = rx->view_stride;
    *n_frames = rx->view_frames;
    return SDRHIP_OK;
}

extern "C" int sdrhip_rx_last_plan(const sdrhip_rx *rx, sdrhip_decim_plan *out)
{
    if (!rx) return fail(SDRHIP_EINVAL, "rx is NULL");
    return sdrhip_decimators_last_plan(rx->dec, out);
}

extern "C" size_t sdrhip_rx_max_frames(const sdrhip_rx *rx, size_t n_in)
{
    if (!rx) return 0;
    sdrhip::CtxLock lock_(rx->ctx);
    const size_t now = (size_t)((rx->pending_samples + (n_in >> rx->cfg.log2decim)) / SDRHIP_SAMPLES_PER_FRAME);
    if (!rx->pipelined) return now;
    return rx->late.have && rx->late.frames > now ? rx->late.frames : now; // (a pipelined call delivers the previous call's frames)
}

// delivery of a finished window: optional copy to the caller's buffer, and the zero-copy view
static int rx_deliver(sdrhip_rx *rx, const uint8_t *base, size_t stride, size_t frames, size_t frame_bytes, uint8_t *frames_out,
                      size_t frame_stride_bytes, size_t *n_frames, int mem)
{
    sdrhip_ctx *c = rx->ctx;
    const int S = rx->nstreams;
    if (frames && frames_out) {
        if (S > 1 && frame_stride_bytes < frames * frame_bytes) return fail(SDRHIP_EINVAL, "rx_process: frame stride too small");
        HIP_TRY(hipMemcpy2DAsync(frames_out, S > 1 ? frame_stride_bytes : frames * frame_bytes, base, stride, frames * frame_bytes, S,
                                 mem == SDRHIP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, c->stream));
    } else if (frames && mem != SDRHIP_MEM_DEVICE) {
        return fail(SDRHIP_EINVAL, "rx_process: NULL frames_out");
    }
    rx->view_base = base; rx->view_stride = stride; rx->view_frames = frames;
    if (n_frames) *n_frames = frames;
    return SDRHIP_OK;
}

extern "C" int sdrhip_rx_flush(sdrhip_rx *rx, uint8_t *frames_out, size_t frame_stride_bytes, size_t *n_frames, int mem)
{
    if (!rx) return fail(SDRHIP_EINVAL, "rx is NULL");
    sdrhip::CtxLock lock_(rx->ctx);
    if (n_frames) *n_frames = 0;
    if (mem != SDRHIP_MEM_HOST && mem != SDRHIP_MEM_DEVICE) return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    if (!rx->late.have) { rx->view_frames = 0; return SDRHIP_OK; }
    HIP_TRY(hipSetDevice(rx->ctx->device));
    int rc;
    if ((rc = rx_settle(rx))) return rc;
    if ((rc = rx_deliver(rx, rx->late.base, rx->late.stride, rx->late.frames, rx->late.frame_bytes, frames_out, frame_stride_bytes, n_frames, mem))) return rc;
    rx->late.have = false;
    if (mem == SDRHIP_MEM_HOST) HIP_TRY(hipStreamSynchronize(rx->ctx->stream));
    return SDRHIP_OK;
}

extern "C" int sdrhip_rx_process(sdrhip_rx *rx, const int16_t *iq_in, size_t n_in, size_t in_stride, uint32_t tv_sec, uint32_t tv_usec,
                                 uint8_t *frames_out, size_t frame_stride_bytes, size_t *n_frames, int mem)
{
    if (!rx) return fail(SDRHIP_EINVAL, "rx is NULL");
    sdrhip::CtxLock lock_(rx->ctx);
    if (n_frames) *n_frames = 0;
    if (n_in == 0) {
        // an empty call completes nothing; in pipelined mode it still DELIVERS what the previous call completed (the header's
        // contract: every call delivers the frames of the one before it)
        if (rx->pipelined && rx->late.have) return sdrhip_rx_flush(rx, frames_out, frame_stride_bytes, n_frames, mem);
        rx->view_frames = 0;
        return SDRHIP_OK;
    }
    if (!iq_in) return fail(SDRHIP_EINVAL, "rx_process: NULL input");
    sdrhip_ctx *c = rx->ctx;
    rx->consumed = false;
    HIP_TRY(hipSetDevice(c->device));
    const int S = rx->nstreams, L = rx->cfg.log2decim, R = rx->cfg.nb_fec;
    const int FB = SDRHIP_NB_ORIGINAL + R;
    const size_t frame_bytes = (size_t)FB * SDRHIP_UDPSIZE;
    const size_t n_dec = n_in >> L;
    const uint64_t total = rx->pending_samples + n_dec;
    const size_t done = (size_t)(total / SDRHIP_SAMPLES_PER_FRAME);
    const uint64_t rest = total - (uint64_t)done * SDRHIP_SAMPLES_PER_FRAME;
    if (S == 1) in_stride = n_in;
    const size_t deliver_now = rx->pipelined ? (rx->late.have ? rx->late.frames : 0) : done;
    const size_t deliver_fb = rx->pipelined && rx->late.have ? rx->late.frame_bytes : frame_bytes;
    if (deliver_now && !frames_out && mem != SDRHIP_MEM_DEVICE) return fail(SDRHIP_EINVAL, "rx_process: NULL frames_out");
    if (frames_out && S > 1 && deliver_now && frame_stride_bytes < deliver_now * deliver_fb) return fail(SDRHIP_EINVAL, "rx_process: frame stride too small");

    const int16_t *din = iq_in;
    size_t dstride = in_stride;
    int rc;
    if (mem == SDRHIP_MEM_HOST) {
        dstride = (n_in + 3) & ~(size_t)3;
        if ((size_t)S * dstride * 4 <= SDRHIP_ZEROCOPY_MAX) {
            // small call: the decimator reads pinned host memory itself (no copy engine in front of the launch); the buffer is
            // free again when this call returns (host-pointer calls end with a stream synchronisation)
            if ((rc = c->zin.reserve((size_t)S * dstride * 4 + 16))) return rc;
            for (int s = 0; s < S; ++s) memcpy(c->zin.as<int16_t>() + (size_t)s * dstride * 2, iq_in + (size_t)s * in_stride * 2, n_in * 4);
            din = c->zin.as<int16_t>();
        } else {
            if ((rc = c->in.reserve((size_t)S * dstride * 4 + 16))) return rc;
            HIP_TRY(hipMemcpy2DAsync(c->in.p, dstride * 4, iq_in, in_stride * 4, n_in * 4, S, hipMemcpyHostToDevice, c->stream));
            din = c->in.as<int16_t>();
        }
    } else if (mem == SDRHIP_MEM_DEVICE) {
        if (!aligned16(iq_in) || (S > 1 && (in_stride & 3))) return fail(SDRHIP_EALIGN, "rx_process: device input must be 16-byte aligned");
    } else {
        return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    }

    // ---- work area [stream][slot][128 + R][512]: this call fills slots base_slot .. base_slot + done.  The
    // finished frames stay readable in place until the next call (sdrhip_rx_frames_view); the frame still
    // being filled is the first slot of the next call.  At the end of the area the window wraps: the open
    // frame moves to slot 0 (one strided copy every few calls instead of a save + restore per call).  Frames that
    // wait for delivery (pipelined mode) are never overwritten: a window that would reach them gets a new area.
    const size_t need = done + 1;
    if (rx->old_work.p && !(rx->late.have && rx->late.slot0 == SIZE_MAX)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        rx->old_work.release();
    }
    if (rx->base_slot + need > rx->cap_frames) {
        const bool late_here = rx->late.have && rx->late.slot0 != SIZE_MAX;
        const bool wrap_hits_late = late_here && need > rx->late.slot0; // the wrapped window [0, need) against [slot0, slot0 + frames)
        if (need > rx->cap_frames || wrap_hits_late) {
            DevBuf bigger;
            // the window: TWO calls' frames (round 5; four until then).  The step rewrites what it wrote two calls ago: 2 x 84 MB per
            // 8-stream bank stay in the 256 MB of Infinity Cache, and on this memory system a write stream that stays there
            // costs the read stream beside it less (profiles/r05_rx_direct.txt: decimator launch 0.2435 -> 0.2335 ms, encoder
            // launch 0.052 -> 0.049 ms; a window of one call wraps -- a copy of the open frames -- on every call).  option rx_window (SDRHIP_RX_WINDOW at context creation, 1..8) = A / B
            // (pipelined pipes keep the previous call's frames until they are delivered: four calls, as before)
            const size_t wmul = c->opt.rx_window ? (size_t)c->opt.rx_window : rx->pipelined ? 4 : 2;
            const size_t ncap = need > rx->cap_frames ? wmul * need : rx->cap_frames;
            if ((rc = bigger.reserve((size_t)S * ncap * frame_bytes))) return rc;
            if (rx->frame_open)
                HIP_TRY(hipMemcpy2DAsync(bigger.p, ncap * frame_bytes, rx->work.as<uint8_t>() + rx->base_slot * frame_bytes,
                                         rx->cap_frames * frame_bytes, frame_bytes, S, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream)); // earlier launches may still use the old area
            if (late_here) { rx->old_work.release(); rx->old_work = rx->work; rx->late.slot0 = SIZE_MAX; }
            else rx->work.release();
            rx->work = bigger;
            rx->cap_frames = ncap;
        } else if (rx->frame_open) {
            uint8_t *w0 = rx->work.as<uint8_t>();
            HIP_TRY(hipMemcpy2DAsync(w0, rx->cap_frames * frame_bytes, w0 + rx->base_slot * frame_bytes, rx->cap_frames * frame_bytes,
                                     frame_bytes, S, hipMemcpyDeviceToDevice, c->stream));
        }
        rx->base_slot = 0;
    }
    const size_t stream_bytes = rx->cap_frames * frame_bytes;
    uint8_t *work = rx->work.as<uint8_t>() + rx->base_slot * frame_bytes; // slot 0 of the window

    // ---- meta record of the frames started by this call (UDPSinkFEC.cpp:87-132); the decimator kernel writes
    // their meta blocks and super block headers on its way
    unsigned ss = rx->cfg.sample_bits;
    const int first_new = rx->frame_open ? 1 : 0;
    const int started = (int)(done + (rest > 0 ? 1 : 0)) - first_new; // frames whose first sample arrives now
    RxMeta meta;
    memset(&meta, 0, sizeof(meta));
    if (started > 0) {
        const unsigned ssd = decimated_sample_size((unsigned)L, ss);
        uint8_t m[24];
        const uint32_t fc = rx->cfg.center_frequency_khz, sr = rx->cfg.sample_rate;
        memcpy(m + 0, &fc, 4); memcpy(m + 4, &sr, 4);
        m[8] = (uint8_t)((ssd - 1) / 8 + 1); // setSampleBytes((sampleSize - 1) / 8 + 1), sdrdaemonrx.cpp:643
        m[9] = (uint8_t)ssd;                 // setSampleBits(sampleSize), :642
        m[10] = SDRHIP_NB_ORIGINAL; m[11] = (uint8_t)R;
        // tv_sec / tv_usec = the stamp of the call's first sample; a frame's own stamp (the reference calls gettimeofday when it
        // opens the frame, UDPSinkFEC.cpp:90-104) is that plus its first sample's offset on the sample clock, and the
        // boost::crc_32_type over the first 20 bytes (:106-109) follows from it: both per frame on the device (frame_meta_words)
        // (m[20..23] = the CRC of the record with a ZERO stamp: the affine part of the per-frame CRC, see frame_meta_words)
        memset(m + 12, 0, 8);
        uint32_t crc = 0xFFFFFFFFu;
        for (int i = 0; i < 20; ++i) {
            crc ^= m[i];
            for (int k = 0; k < 8; ++k) crc = (crc & 1) ? 0xEDB88320u ^ (crc >> 1) : crc >> 1;
        }
        crc ^= 0xFFFFFFFFu;
        memcpy(m + 12, &tv_sec, 4); memcpy(m + 16, &tv_usec, 4); memcpy(m + 20, &crc, 4);
        meta.first = first_new; meta.count = started; meta.frame_count0 = (unsigned)rx->frame_count + first_new;
        memcpy(meta.w, m, 24);
        meta.idx0 = first_new ? (uint64_t)SDRHIP_SAMPLES_PER_FRAME - rx->pending_samples : 0;
        meta.rate = sr;
    }

    size_t n_out = 0;
    EncodeLin elin;
    bool use_lin = false, pack_with_encoder = false;
    FrameArgs fa;
    memset(&fa, 0, sizeof(fa));
    const bool structured = R >= enc128_min_rows(c) && frame_bytes % 4 == 0; // gf_encode128_kernel serves this setting
    const bool filterless = L == 0 || (rx->cfg.fcpos != SDRHIP_FC_CEN && L <= 2); // Decimators.cpp:22-91,127-170: no cascade kernel
    const Enc128Args *fuse = rx->late.encode && (c->opt.rx_fused == 1 || c->opt.rx_fused == 2) ? &rx->late.k : nullptr;
    // overlap mode: the waiting encode will run on the second stream BESIDE this call's decimator (rx_settle below), which therefore
    // leaves room on its CUs (ring depth 3) and raises its waves' priority
    const bool coresident = rx->late.encode && c->opt.rx_fused == 3 && rx->ev_framed;
    bool fused = false;
    // (matrix-core decimator: stream order + K2 + the encoder's fused copy, unless its waves frame their output themselves)
    const bool direct = c->opt.rx_direct && !rx->pipelined && stream_bytes < 0x3fffffffu;
    if (filterless || (!direct && decimate_mfma_applies(rx->dec, L, rx->cfg.fcpos, n_in))) {
        // ---- decimate in stream order, then K2 lays the samples out as super blocks (+ meta blocks and headers)
        const size_t lstride = (n_dec + 3) & ~(size_t)3;
        if (rx->pipelined) rx->lin_sel ^= 1; // (the deferred encoder of the previous call still reads the other one)
        DevBuf &lin = rx->lin[rx->lin_sel];
        if (lin.cap < (size_t)S * lstride * 4 + 16 && rx->late.encode) { if ((rc = rx_settle(rx))) return rc; fuse = nullptr; HIP_TRY(hipStreamSynchronize(c->stream)); }
        if ((rc = lin.reserve((size_t)S * lstride * 4 + 16))) return rc;
        rc = decimate_device(rx->dec, L, rx->cfg.fcpos, &ss, din, n_in, dstride, lin.as<int16_t>(), lstride, &n_out, 0, 0, 0, nullptr, fuse, &fused, coresident);
        if (rc) return rc;
        rx->consumed = true;
        if (fused) rx->late.encode = false;
        // the frames that lie entirely inside this call's samples are laid out by the encoder (fused copy); K2 does
        // the frame that was open when the call began, the one left open at its end, meta blocks and headers
        if (fec_encode_fuses_framing(c, R)) {
            const size_t first = rx->pending_samples ? 1 : 0;
            if (done > first && frame_bytes % 4 == 0) {
                elin.lin = lin.as<unsigned>(); elin.stride = lstride; elin.cap = (int)rx->cap_frames;
                elin.first = (int)first; elin.pending = (int)rx->pending_samples;
                use_lin = true;
            }
        }
        if (use_lin) {
            fa.skip_from = (size_t)elin.first * SDRHIP_SAMPLES_PER_FRAME - (size_t)elin.pending;
            fa.skip_to = done * SDRHIP_SAMPLES_PER_FRAME - (size_t)elin.pending;
        }
        fa.in = lin.as<unsigned>(); fa.out = reinterpret_cast<unsigned *>(work);
        fa.in_stride = lstride; fa.out_stride = stream_bytes / 4;
        fa.n = n_dec; fa.frame_sample_base = rx->pending_samples; fa.frame_blocks = FB;
        fa.meta_first = meta.first; fa.meta_count = meta.count; fa.meta_frame_count0 = meta.frame_count0;
        memcpy(fa.meta_w, meta.w, sizeof(fa.meta_w));
        fa.meta_idx0 = meta.idx0; fa.meta_rate = meta.rate;
        // K2 rides in the encoder's launch when this call's frames are encoded right away by the structured encoder (one launch
        // less per step); otherwise it goes out now
        pack_with_encoder = !rx->pipelined && c->opt.rx_fused && structured && R > 0 && use_lin;
        if (!pack_with_encoder) {
            hipError_t e = launch_frame_pack(fa, S, c->stream);
            if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "frame pack launch: %s", hipGetErrorString(e));
        }
    } else {
        // ---- decimate straight into the frame layout (VALU cascade kernel with the framing epilogue, or the matrix-core kernel
        // with its frame-layout stores and the VALU pieces' epilogue for meta blocks and headers)
        rc = decimate_device(rx->dec, L, rx->cfg.fcpos, &ss, din, n_in, dstride, reinterpret_cast<int16_t *>(work), stream_bytes / 4, &n_out, 1,
                             FB, rx->pending_samples, &meta);
        if (rc) return rc;
        rx->consumed = true;
    }
    if ((rc = rx_settle(rx))) return rc; // (a waiting encode that this call's launch could not take along)

    // ---- FEC over the completed frames of every stream, recovery blocks land behind block 127
    bool encode_later = false;
    Enc128Args k;
    memset(&k, 0, sizeof(k));
    if (done && R > 0) {
        if (structured) {
            // structured encoder, one workgroup per (frame, half block); frame (s, f) of the window is frame s * cap_frames + f
            k.in = work; k.out = work + (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE; k.tab = c->gf_tab; k.leaf_tables = c->enc_leaves; k.fft_tables = c->enc_fft; k.use_fft = c->opt.enc_fft;
            k.in_frame_bytes = frame_bytes; k.out_frame_bytes = frame_bytes;
            k.rows = R; k.nframes = (int)((size_t)S * rx->cap_frames);
            k.nlist = (int)((size_t)S * done); k.gen_done = (int)done; k.gen_cap = (int)rx->cap_frames;
            if (use_lin) { k.lin = elin.lin; k.lin_stride = elin.stride; k.lin_cap = elin.cap; k.lin_first = elin.first; k.lin_pending = elin.pending; }
            if (rx->pipelined) {
                encode_later = true; // rides in the next call's decimator launch (or sdrhip_rx_flush)
            } else if (pack_with_encoder) {
                // encoder + K2 in one launch: the encoder derives the meta blocks of the frames this call starts itself and
                // completes the frame that was open (its tail comes from the stream-order buffer), K2 leaves both alone
                k.meta_first = meta.first; k.meta_count = meta.count; k.meta_frame_count0 = meta.frame_count0;
                memcpy(k.meta_w, meta.w, sizeof(k.meta_w));
                k.meta_idx0 = meta.idx0; k.meta_rate = meta.rate;
                if (elin.first == 1) { k.lin_straddle = 1; fa.skip_from = 0; }
                hipError_t e;
                {
                    KTimer kt(c, SDRHIP_K_FEC_ENCODE);
                    e = launch_gf_encode128_pack(k, fa, S, c->stream);
                }
                if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "encode + frame pack launch: %s", hipGetErrorString(e));
            } else if ((rc = fec_encode128_launch(c, k))) return rc;
        } else {
            // generic matrix kernel: one launch for every stream, frame list in groups of GF_FRAMES_PER_GROUP
            if (rx->flist_done != done || rx->flist_cap != rx->cap_frames) {
                HIP_TRY(hipStreamSynchronize(c->stream)); // a previous upload may still read flist_host
                rx->flist_host.clear();
                for (int s = 0; s < S; ++s)
                    for (size_t f = 0; f < done; ++f) rx->flist_host.push_back((int32_t)(s * rx->cap_frames + f));
                while (rx->flist_host.size() % GF_FRAMES_PER_GROUP) rx->flist_host.push_back(-1);
                if ((rc = rx->flist.reserve(rx->flist_host.size() * 4))) return rc;
                HIP_TRY(hipMemcpyAsync(rx->flist.p, rx->flist_host.data(), rx->flist_host.size() * 4, hipMemcpyHostToDevice, c->stream));
                rx->flist_done = done; rx->flist_cap = rx->cap_frames;
            }
            if ((rc = fec_encode_device(c, work, frame_bytes, (size_t)S * rx->cap_frames, R, work + (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE,
                                        frame_bytes, rx->flist.as<int32_t>(), (int)(rx->flist_host.size() / GF_FRAMES_PER_GROUP), nullptr)))
                return rc;
        }
    }
    // ---- delivery: this call's frames, or (pipelined) the previous call's, whose encode went out above
    if (rx->pipelined) {
        if (rx->late.have) {
            if ((rc = rx_deliver(rx, rx->late.base, rx->late.stride, rx->late.frames, rx->late.frame_bytes, frames_out, frame_stride_bytes, n_frames, mem))) return rc;
        } else {
            rx->view_base = nullptr; rx->view_frames = 0;
        }
        rx->late.have = done > 0;
        rx->late.encode = encode_later;
        if (encode_later && c->opt.rx_fused == 3) { // (everything the deferred encode reads has been enqueued on the first stream by now)
            if (!rx->ev_framed) HIP_TRY(hipEventCreateWithFlags(&rx->ev_framed, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(rx->ev_framed, c->stream));
        }
        rx->late.k = k;
        rx->late.base = work; rx->late.stride = stream_bytes; rx->late.frames = done; rx->late.frame_bytes = frame_bytes;
        rx->late.slot0 = rx->base_slot;
    } else {
        if ((rc = rx_deliver(rx, work, stream_bytes, done, frame_bytes, frames_out, frame_stride_bytes, n_frames, mem))) return rc;
    }
    rx->base_slot += done; // the frame still being filled opens the next call's window
    rx->pending_samples = rest;
    rx->frame_open = rest > 0;
    rx->frame_count = (uint16_t)(rx->frame_count + done);
    if (mem == SDRHIP_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    return SDRHIP_OK;
}

// --------------------------------------------------------------------------- asynchronous host-pointer Rx entry
// sdrdaemonrx's chain is asynchronous end to end (source thread -> source_buffer -> main loop -> output_buffer -> writer ->
// transmit thread, sdrdaemonrx.cpp:555-663): a block's frames leave the process long after Downsampler::process returned.
// sdrhip_rx_process on host pointers is one synchronous launch per block (38 us for a 65 536-sample TestSource block, of which
// the GPU works ~10); submit / collect give the host-pointer path the same asynchrony: blocks are appended to a pinned
// staging buffer (or taken in place from sdrhip_host_alloc memory), every `blocks` of them go out as ONE upload + launch +
// download on the context's stream, and the frames are collected later, batch by batch, in order.
namespace {
// memory handed out by sdrhip_host_alloc: pinned, usable in place
struct HostRange { const char *p; size_t n; };
std::mutex g_host_mtx;
std::vector<HostRange> g_host_ranges;
bool host_is_pinned(const void *p, size_t n)
{
    std::lock_guard<std::mutex> g(g_host_mtx);
    const char *c = static_cast<const char *>(p);
    for (const HostRange &r : g_host_ranges)
        if (c >= r.p && c + n <= r.p + r.n) return true;
    return false;
}
int rx_launch_batch(sdrhip_rx *rx, sdrhip_rx::Batch &b)
{
    sdrhip_ctx *c = rx->ctx;
    const int S = rx->nstreams;
    const size_t dstride = (b.n_in + 3) & ~(size_t)3;
    int rc;
    if ((rc = b.din.reserve((size_t)S * dstride * 4 + 16))) return rc;
    // uploads: runs of blocks that are adjacent in host memory go out as ONE 2-D copy (a run of staged blocks -- stream-major in
    // the pinned arena -- or of in-place blocks cut from one buffer)
    size_t off = 0;
    for (size_t i = 0; i < b.blocks.size();) {
        const int16_t *src = b.blocks[i].first;
        size_t sstride = b.strides[i], n = b.blocks[i].second, j = i + 1;
        if (!src) { // staged: [stream][in_cap] at sample offset `off` of every row (all staged blocks of a batch are one run)
            src = b.in.as<int16_t>() + off * 2;
            sstride = b.in_cap;
            while (j < b.blocks.size() && !b.blocks[j].first) n += b.blocks[j++].second;
        } else {
            while (j < b.blocks.size() && b.blocks[j].first == src + n * 2 && b.strides[j] == sstride) n += b.blocks[j++].second;
        }
        if (S == 1) HIP_TRY(hipMemcpyAsync(b.din.as<char>() + off * 4, src, n * 4, hipMemcpyHostToDevice, c->stream)); // (no pitch limits)
        else HIP_TRY(hipMemcpy2DAsync(b.din.as<char>() + off * 4, dstride * 4, src, sstride * 4, n * 4, S, hipMemcpyHostToDevice, c->stream));
        off += n;
        i = j;
    }
    b.in.mark(c->stream);
    // everything that can fail for want of memory happens BEFORE the samples are consumed: a batch that failed here, or that
    // sdrhip_rx_process refused before its decimator launch went out, is launched again by the next submit / collect; one that fails
    // behind the decimator launch (rx->consumed) is dropped: its samples are in the filter state already, replaying them would
    // duplicate samples and shift every later stamp
    b.frame_bytes = (size_t)(SDRHIP_NB_ORIGINAL + rx->cfg.nb_fec) * SDRHIP_UDPSIZE;
    const size_t nf_max = sdrhip_rx_max_frames(rx, b.n_in);
    if (nf_max && (rc = b.out.reserve((size_t)S * nf_max * b.frame_bytes))) return rc;
    if (!b.done && hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) { b.done = nullptr; return fail(SDRHIP_EDEVICE, "hipEventCreate"); }
    size_t nf = 0;
    rc = sdrhip_rx_process(rx, b.din.as<int16_t>(), b.n_in, dstride, b.tv_sec, b.tv_usec, nullptr, 0, &nf, SDRHIP_MEM_DEVICE);
    if (rc) {
        if (rx->consumed) b.state = 0; // consumed and lost: never replayed (the pipe's own error stands)
        return rc;
    }
    b.frames = nf;
    hipError_t e = hipSuccess;
    if (nf > nf_max) e = hipErrorInvalidValue; // (cannot happen: rx_max_frames is the pipe's own bound)
    else if (nf && S == 1) e = hipMemcpyAsync(b.out.p, rx->view_base, nf * b.frame_bytes, hipMemcpyDeviceToHost, c->stream);
    else if (nf) e = hipMemcpy2DAsync(b.out.p, nf * b.frame_bytes, rx->view_base, rx->view_stride, nf * b.frame_bytes, S, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipEventRecord(b.done, c->stream);
    if (e != hipSuccess) {
        b.state = 0; // consumed and lost: never replayed
        return fail(SDRHIP_EDEVICE, "rx batch download: %s (the batch's %zu frames per stream are lost)", hipGetErrorString(e), nf);
    }
    b.state = 2;
    return SDRHIP_OK;
}
} // namespace

extern "C" void *sdrhip_host_alloc(sdrhip_ctx *c, size_t bytes)
{
    if (!c || bytes == 0) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)fail(SDRHIP_ENOMEM, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    std::lock_guard<std::mutex> g(g_host_mtx);
    g_host_ranges.push_back(HostRange{static_cast<const char *>(p), bytes});
    return p;
}

extern "C" void sdrhip_host_free(sdrhip_ctx *c, void *p)
{
    if (!p) return;
    (void)c;
    {
        std::lock_guard<std::mutex> g(g_host_mtx);
        for (size_t i = 0; i < g_host_ranges.size(); ++i)
            if (g_host_ranges[i].p == p) { g_host_ranges.erase(g_host_ranges.begin() + (long)i); break; }
    }
    (void)hipHostFree(p);
}

extern "C" int sdrhip_rx_set_async(sdrhip_rx *rx, int depth, int blocks)
{
    if (!rx) return fail(SDRHIP_EINVAL, "rx is NULL");
    sdrhip::CtxLock lock_(rx->ctx);
    if (depth < 1 || depth > 64 || blocks < 1 || blocks > 1024) return fail(SDRHIP_EINVAL, "rx_set_async: depth 1..64, blocks 1..1024");
    for (auto &b : rx->abatch)
        if (b.state != 0) return fail(SDRHIP_EINVAL, "rx_set_async: batches are in flight: collect them first");
    for (auto &b : rx->abatch) { if (b.done) (void)hipEventDestroy(b.done); b.in.release(); b.din.release(); b.out.release(); }
    rx->abatch.assign((size_t)depth, sdrhip_rx::Batch());
    rx->a_blocks = blocks; rx->a_head = rx->a_tail = 0;
    return SDRHIP_OK;
}

extern "C" int sdrhip_rx_submit(sdrhip_rx *rx, const int16_t *iq_in, size_t n_in, size_t in_stride, uint32_t tv_sec, uint32_t tv_usec)
{
    if (!rx) return fail(SDRHIP_EINVAL, "rx is NULL");
    sdrhip::CtxLock lock_(rx->ctx);
    if (n_in == 0) return SDRHIP_OK;
    if (!iq_in) return fail(SDRHIP_EINVAL, "rx_submit: NULL input");
    if (rx->abatch.empty()) { rx->abatch.assign(4, sdrhip_rx::Batch()); rx->a_blocks = 1; }
    HIP_TRY(hipSetDevice(rx->ctx->device));
    const int S = rx->nstreams;
    if (S == 1) in_stride = n_in;
    sdrhip_rx::Batch &b = rx->abatch[rx->a_tail % rx->abatch.size()];
    if (b.state == 2) return fail(SDRHIP_EBUSY, "rx_submit: every batch of the ring is in flight: sdrhip_rx_collect first");
    if (b.state == 0) {
        b.blocks.clear(); b.strides.clear(); b.n_in = 0; b.tv_sec = tv_sec; b.tv_usec = tv_usec; b.state = 1;
        b.in_cap = 0; // no staged rows yet: the first pageable block of this batch (re)claims the arena
    }
    if (host_is_pinned(iq_in, ((size_t)(S - 1) * in_stride + n_in) * 4)) {
        b.blocks.push_back(std::make_pair(iq_in, n_in)); // in place: the caller keeps it untouched until the batch is collected
        b.strides.push_back(in_stride);
    } else {
        // staged: row s of the pinned arena holds stream s, the block at the batch's current sample offset
        // (in-place blocks in front of it leave their part of the rows unused: a block always sits at its batch offset)
        const size_t need = b.n_in + n_in;
        if (b.in_cap == 0) { // first staged block of the batch, whatever came before it in place
            const size_t cap = (size_t)rx->a_blocks * n_in > need ? (size_t)rx->a_blocks * n_in : need;
            int rc = b.in.reserve((size_t)S * cap * 4); // (waits for the upload of the batch that used this buffer last)
            if (rc) return rc;
            b.in_cap = cap;
        } else if (need > b.in_cap) { // blocks longer than the first one: re-lay the rows out in a bigger arena
            PinnedBuf bigger;
            const size_t ncap = 2 * need;
            int rc = bigger.reserve((size_t)S * ncap * 4);
            if (rc) return rc;
            for (int s = 0; s < S; ++s) memcpy(bigger.as<char>() + (size_t)s * ncap * 4, b.in.as<char>() + (size_t)s * b.in_cap * 4, b.n_in * 4);
            b.in.release();
            b.in = bigger;
            b.in_cap = ncap;
        }
        for (int s = 0; s < S; ++s) memcpy(b.in.as<char>() + ((size_t)s * b.in_cap + b.n_in) * 4, iq_in + (size_t)s * in_stride * 2, n_in * 4);
        b.blocks.push_back(std::make_pair((const int16_t *)nullptr, n_in));
        b.strides.push_back(n_in);
    }
    b.n_in += n_in;
    if ((int)b.blocks.size() >= rx->a_blocks) {
        int rc = rx_launch_batch(rx, b);
        if (rc) return rc;
        ++rx->a_tail;
    }
    return SDRHIP_OK;
}

extern "C" int sdrhip_rx_collect(sdrhip_rx *rx, uint8_t *frames_out, size_t frame_stride_bytes, size_t max_frames, size_t *n_frames, int wait)
{
    if (!rx || !n_frames) return fail(SDRHIP_EINVAL, "rx_collect: NULL argument");
    // (the wait for a batch happens OUTSIDE the context lock: the submitting thread -- the reference's source / main thread -- keeps
    // feeding the ring while the collecting thread -- its transmit thread -- sleeps on the oldest batch's event)
    std::unique_lock<std::recursive_mutex> lock_(rx->ctx->mtx);
    *n_frames = 0;
    if (rx->abatch.empty()) return fail(SDRHIP_EBUSY, "rx_collect: nothing was submitted");
    HIP_TRY(hipSetDevice(rx->ctx->device));
    sdrhip_rx::Batch *bp = nullptr;
    for (;;) {
        sdrhip_rx::Batch &h = rx->abatch[rx->a_head % rx->abatch.size()];
        if (h.state == 0) return fail(SDRHIP_EBUSY, "rx_collect: nothing was submitted"); // (SDRHIP_OK always means: one batch collected, *n_frames of it -- possibly 0)
        if (h.state == 1) {
            if (!wait) return fail(SDRHIP_EBUSY, "rx_collect: the oldest batch is still being filled (wait = 1 launches it as it is)");
            int rc = rx_launch_batch(rx, h); // a partly filled batch goes out as it is (end of stream)
            if (rc) return rc;
            ++rx->a_tail;
        }
        const hipError_t q = hipEventQuery(h.done);
        if (q == hipSuccess) { bp = &h; break; }
        if (q != hipErrorNotReady) return fail(SDRHIP_EDEVICE, "hipEventQuery: %s", hipGetErrorString(q));
        if (!wait) return fail(SDRHIP_EBUSY, "rx_collect: the oldest batch is still in flight");
        const size_t head = rx->a_head;
        hipEvent_t ev = h.done;
        lock_.unlock();
        const hipError_t w = hipEventSynchronize(ev);
        lock_.lock();
        if (w != hipSuccess) return fail(SDRHIP_EDEVICE, "hipEventSynchronize: %s", hipGetErrorString(w));
        if (rx->a_head == head) { bp = &rx->abatch[head % rx->abatch.size()]; break; }
        // (another thread collected that batch meanwhile: look at the new head)
    }
    sdrhip_rx::Batch &b = *bp;
    const int S = rx->nstreams;
    if (b.frames > max_frames) { // (the batch stays where it is: call again with room for *n_frames frames per stream)
        *n_frames = b.frames;
        return fail(SDRHIP_EINVAL, "rx_collect: the batch holds %zu frames per stream, frames_out has room for %zu", b.frames, max_frames);
    }
    if (b.frames) {
        if (!frames_out) return fail(SDRHIP_EINVAL, "rx_collect: NULL frames_out");
        const size_t row = b.frames * b.frame_bytes;
        if (S > 1 && frame_stride_bytes < row) return fail(SDRHIP_EINVAL, "rx_collect: frame stride too small");
        for (int s = 0; s < S; ++s) memcpy(frames_out + (size_t)s * (S > 1 ? frame_stride_bytes : row), b.out.as<char>() + (size_t)s * row, row);
    }
    *n_frames = b.frames;
    b.state = 0;
    ++rx->a_head;
    return SDRHIP_OK;
}

// --------------------------------------------------------------------------- fused Tx pipe
struct sdrhip_tx {
    sdrhip_ctx *ctx;
    int nstreams;
    int log2interp;
    sdrhip_interpolators *itp;
    DevBuf rxbuf, payload[2], outbuf;
    DevBuf srcmap, restored;  // no-copy mode (tx_gather): the decoder's position map and restored blocks, read by K5w's gather variant
    size_t restored_slots = 0; // slots `restored` was zero-terminated for (its last slot must read zero)
    // ---- pipelined mode (sdrhip_tx_set_pipelined): a call decodes ITS batch into payload[psel] -- on the context's second stream,
    // with work buffers of its own -- while the first stream interpolates the batch the PREVIOUS call decoded (payload[psel ^ 1]);
    // the samples are delivered one call late, like SDRdaemonFECBuffer delivers a frame when the next one begins (.cpp:133-139)
    int pipelined = 0;
    int psel = 0;
    struct Late {
        bool have = false;
        size_t n_payload = 0, pstride = 0;
        int log2interp = 0; // the factor in force when the batch was handed in
    } late;
    DevBuf plan_own, idx_own;
    PinnedBuf pin_own;
    hipEvent_t ev_in = nullptr;              // first stream: the caller's device rx buffer is ready
    hipEvent_t ev_up = nullptr;              // second stream: the upload of the caller's HOST rx buffer has read it (the call returns behind it)
    hipEvent_t ev_dec = nullptr;             // second stream: the waiting batch is decoded
    hipEvent_t ev_itp[2] = {nullptr, nullptr}; // first stream: the interpolator has read payload[i]
    bool itp_pending[2] = {false, false};
    // ---- asynchronous host-pointer entry (sdrhip_tx_submit / sdrhip_tx_collect): a ring of batches of received frames
    struct ABatch {
        PinnedBuf in;             // the batch's received super blocks [stream][frame][128][512] (staged; sdrhip_host_alloc memory is used in place)
        DevBuf din, dout, db0;    // ... on the device; its samples [stream][dos]; its meta blocks [stream * nframes][508]
        PinnedBuf out;            // samples, then meta blocks, downloaded
        hipEvent_t done = nullptr;
        size_t nframes = 0, n_res = 0, dos = 0;
        int state = 0;            // 0 free, 2 in flight
    };
    std::vector<ABatch> abatch;
    size_t a_head = 0, a_tail = 0;
};

extern "C" int sdrhip_tx_create(sdrhip_ctx *ctx, int nstreams, int log2interp, sdrhip_tx **out)
{
    if (!ctx || !out || nstreams <= 0) return fail(SDRHIP_EINVAL, "tx_create: bad argument");
    if (log2interp < 0 || log2interp > 6) return fail(SDRHIP_EINVAL, "Invalid log2 interpolation factor");
    sdrhip_tx *tx = new (std::nothrow) sdrhip_tx();
    if (!tx) return fail(SDRHIP_ENOMEM, "out of host memory");
    tx->ctx = ctx; tx->nstreams = nstreams; tx->log2interp = log2interp; tx->itp = nullptr;
    int rc = sdrhip_interpolators_create(ctx, nstreams, &tx->itp);
    if (rc) { delete tx; return rc; }
    *out = tx;
    return SDRHIP_OK;
}

// Upsampler::configure (Upsampler.cpp:31-50), applied between two batches like sdrdaemontx does with a control message
// (sdrdaemontx.cpp:381): the six interpolator instances are shared by every interpolateN entry point
// (Interpolators.h:47-52), so their histories carry over.  (Pipelined mode: a batch that waits for delivery keeps the factor
// it was handed in with.)
extern "C" int sdrhip_tx_reconfigure(sdrhip_tx *tx, int log2interp)
{
    if (!tx) return fail(SDRHIP_EINVAL, "tx is NULL");
    sdrhip::CtxLock lock_(tx->ctx);
    if (log2interp < 0 || log2interp > 6) return fail(SDRHIP_EINVAL, "Invalid log2 interpolation factor"); // Upsampler.cpp:38-42
    tx->log2interp = log2interp;
    return SDRHIP_OK;
}

extern "C" void sdrhip_tx_destroy(sdrhip_tx *tx)
{
    if (!tx) return;
    (void)hipSetDevice(tx->ctx->device);
    if (tx->ctx->stream2) (void)hipStreamSynchronize(tx->ctx->stream2); // (a decode of the pipelined mode may still run there)
    sdrhip_interpolators_destroy(tx->itp); // (synchronises the first stream)
    tx->rxbuf.release(); tx->payload[0].release(); tx->payload[1].release(); tx->outbuf.release();
    tx->srcmap.release(); tx->restored.release();
    tx->plan_own.release(); tx->idx_own.release(); tx->pin_own.release();
    for (auto &b : tx->abatch) {
        if (b.done) { (void)hipEventSynchronize(b.done); (void)hipEventDestroy(b.done); }
        b.in.release(); b.din.release(); b.dout.release(); b.db0.release(); b.out.release();
    }
    if (tx->ev_in) (void)hipEventDestroy(tx->ev_in);
    if (tx->ev_up) (void)hipEventDestroy(tx->ev_up);
    if (tx->ev_dec) (void)hipEventDestroy(tx->ev_dec);
    for (int i = 0; i < 2; ++i) if (tx->ev_itp[i]) (void)hipEventDestroy(tx->ev_itp[i]);
    delete tx;
}

extern "C" int sdrhip_tx_set_pipelined(sdrhip_tx *tx, int on)
{
    if (!tx) return fail(SDRHIP_EINVAL, "tx is NULL");
    sdrhip::CtxLock lock_(tx->ctx);
    if (!on && tx->late.have) return fail(SDRHIP_EINVAL, "tx_set_pipelined: sdrhip_tx_flush the waiting batch first");
    if (on) {
        HIP_TRY(hipSetDevice(tx->ctx->device));
        if (!tx->ev_in) HIP_TRY(hipEventCreateWithFlags(&tx->ev_in, hipEventDisableTiming));
        if (!tx->ev_up) HIP_TRY(hipEventCreateWithFlags(&tx->ev_up, hipEventDisableTiming));
        if (!tx->ev_dec) HIP_TRY(hipEventCreateWithFlags(&tx->ev_dec, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) if (!tx->ev_itp[i]) HIP_TRY(hipEventCreateWithFlags(&tx->ev_itp[i], hipEventDisableTiming));
    }
    tx->pipelined = on ? 1 : 0;
    return SDRHIP_OK;
}

namespace {
// decode S x nframes frames into `pay` ([S][pstride] samples): one batch, or one call per stream when the rows are padded
int tx_decode(sdrhip_tx *tx, const uint8_t *drx, const uint8_t *indices, size_t nframes, DevBuf &pay, size_t pstride, const DecodeSide *side,
              uint8_t *block0 = nullptr) // block0 (optional, device): [stream * nframes][508], the frames' meta blocks
{
    sdrhip_ctx *c = tx->ctx;
    const int S = tx->nstreams;
    const size_t fb = (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE, n_payload = nframes * SDRHIP_SAMPLES_PER_FRAME;
    int rc;
    if (pstride == n_payload)
        return fec_decode_device(c, drx, fb, indices, (size_t)S * nframes, pay.as<uint8_t>(), (size_t)127 * SDRHIP_BLOCK_BYTES, block0, side);
    for (int s = 0; s < S; ++s)
        if ((rc = fec_decode_device(c, drx + (size_t)s * nframes * fb, fb, indices ? indices + (size_t)s * nframes * SDRHIP_NB_ORIGINAL : nullptr, nframes,
                                    pay.as<uint8_t>() + (size_t)s * pstride * 4, (size_t)127 * SDRHIP_BLOCK_BYTES,
                                    block0 ? block0 + (size_t)s * nframes * SDRHIP_BLOCK_BYTES : nullptr, side)))
            return rc;
    return SDRHIP_OK;
}

// no-copy mode of a batch (round 6): the decoder leaves the received originals in drx and writes only the restored blocks + a map;
// the wave interpolator gathers through the map.  Applies to immediate calls and the asynchronous entry (the received frames must
// outlive the interpolator: a pipelined call interpolates one call LATE, when a device caller may have reused its buffer).
bool tx_gather_applies(const sdrhip_tx *tx, int log2interp)
{
    const sdrhip_ctx *c = tx->ctx;
    return c->opt.tx_gather && !tx->pipelined && interpolate_gather_ok(c, log2interp) && fec_decode_gather_ok(c);
}
int tx_decode_gather(sdrhip_tx *tx, const uint8_t *drx, const uint8_t *indices, size_t nframes, InterpGather *g, uint8_t *block0 = nullptr)
{
    sdrhip_ctx *c = tx->ctx;
    const size_t F = (size_t)tx->nstreams * nframes, rows = (size_t)c->opt.dec_max_rows, slots = F * rows + 1;
    if (F * 128 >= 0x7fffffffu || slots >= 0x7fffffffu) return fail(SDRHIP_EINVAL, "tx: too many frames in one call");
    int rc;
    if ((rc = tx->srcmap.reserve(F * 128 * sizeof(unsigned)))) return rc;
    if ((rc = tx->restored.reserve(slots * SDRHIP_BLOCK_BYTES))) return rc;
    if (tx->restored_slots != slots) { // (the all-zero slot behind the last frame's: wherever it lies for this batch size)
        HIP_TRY(hipMemsetAsync(tx->restored.as<uint8_t>() + (slots - 1) * SDRHIP_BLOCK_BYTES, 0, SDRHIP_BLOCK_BYTES, c->stream));
        tx->restored_slots = slots;
    }
    DecodeGather dg;
    dg.srcmap = tx->srcmap.as<unsigned>(); dg.restored = tx->restored.as<uint8_t>(); dg.rows = (int)rows;
    const size_t fb = (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE;
    if ((rc = fec_decode_device(c, drx, fb, indices, F, nullptr, 0, block0, nullptr, &dg))) return rc;
    g->map = dg.srcmap; g->rx = drx; g->restored = dg.restored; g->frames = (int)nframes;
    return SDRHIP_OK;
}

// interpolate a decoded batch on the first stream into the caller's buffer (host: through outbuf + a download)
int tx_interpolate(sdrhip_tx *tx, int log2interp, const DevBuf &pay, size_t n_payload, size_t pstride, int16_t *iq_out, size_t out_stride, int mem,
                   const InterpGather *gather = nullptr)
{
    sdrhip_ctx *c = tx->ctx;
    const int S = tx->nstreams;
    const size_t n_res = n_payload << log2interp;
    int16_t *dout = iq_out;
    size_t dos = out_stride;
    int rc;
    if (mem == SDRHIP_MEM_HOST) {
        dos = (n_res + 3) & ~(size_t)3;
        if ((rc = tx->outbuf.reserve((size_t)S * dos * 4 + 16))) return rc;
        dout = tx->outbuf.as<int16_t>();
    }
    if ((rc = interpolate_device(tx->itp, log2interp, gather ? nullptr : pay.as<int16_t>(), n_payload, pstride, dout, dos, nullptr, gather))) return rc;
    if (mem == SDRHIP_MEM_HOST)
        HIP_TRY(hipMemcpy2DAsync(iq_out, out_stride * 4, dout, dos * 4, n_res * 4, S, hipMemcpyDeviceToHost, c->stream));
    return SDRHIP_OK;
}

// delivery half of a pipelined call (and of sdrhip_tx_flush): the waiting batch through the interpolator on the first stream
int tx_deliver_late(sdrhip_tx *tx, int16_t *iq_out, size_t out_stride, size_t *n_out, int mem)
{
    sdrhip_ctx *c = tx->ctx;
    const int S = tx->nstreams;
    const size_t n_res = tx->late.n_payload << tx->late.log2interp;
    if (S == 1) out_stride = n_res;
    if (!iq_out) return fail(SDRHIP_EINVAL, "tx: NULL output buffer for the waiting batch");
    if (out_stride < n_res) return fail(SDRHIP_EINVAL, "tx: out_stride smaller than the waiting batch (%zu samples per stream)", n_res);
    if (mem == SDRHIP_MEM_DEVICE && (!aligned16(iq_out) || (S > 1 && (out_stride & 3)))) return fail(SDRHIP_EALIGN, "tx_process: device output must be 16-byte aligned");
    const int sel = tx->psel ^ 1; // (psel already points at the buffer the NEXT decode fills)
    HIP_TRY(hipStreamWaitEvent(c->stream, tx->ev_dec, 0));
    int rc = tx_interpolate(tx, tx->late.log2interp, tx->payload[sel], tx->late.n_payload, tx->late.pstride, iq_out, out_stride, mem);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(tx->ev_itp[sel], c->stream));
    tx->itp_pending[sel] = true;
    tx->late.have = false;
    if (n_out) *n_out = n_res;
    return SDRHIP_OK;
}
} // namespace

extern "C" int sdrhip_tx_flush(sdrhip_tx *tx, int16_t *iq_out, size_t out_stride, size_t *n_out, int mem)
{
    if (!tx) return fail(SDRHIP_EINVAL, "tx is NULL");
    sdrhip::CtxLock lock_(tx->ctx);
    if (n_out) *n_out = 0;
    if (mem != SDRHIP_MEM_HOST && mem != SDRHIP_MEM_DEVICE) return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    if (!tx->late.have) return SDRHIP_OK;
    HIP_TRY(hipSetDevice(tx->ctx->device));
    int rc = tx_deliver_late(tx, iq_out, out_stride, n_out, mem);
    if (rc) return rc;
    if (mem == SDRHIP_MEM_HOST) HIP_TRY(hipStreamSynchronize(tx->ctx->stream));
    return SDRHIP_OK;
}

extern "C" size_t sdrhip_tx_pending_samples(const sdrhip_tx *tx)
{
    if (!tx) return 0;
    sdrhip::CtxLock lock_(tx->ctx);
    return tx->late.have ? tx->late.n_payload << tx->late.log2interp : 0;
}

extern "C" int sdrhip_tx_process(sdrhip_tx *tx, const uint8_t *rx, const uint8_t *indices, size_t nframes, size_t rx_stride_bytes,
                                 int16_t *iq_out, size_t out_stride, size_t *n_out, int mem)
{
    if (!tx) return fail(SDRHIP_EINVAL, "tx is NULL");
    sdrhip::CtxLock lock_(tx->ctx);
    const size_t n_payload = nframes * SDRHIP_SAMPLES_PER_FRAME;
    const size_t n_res = n_payload << tx->log2interp;
    if (n_out) *n_out = tx->pipelined ? 0 : n_res;
    if (mem != SDRHIP_MEM_HOST && mem != SDRHIP_MEM_DEVICE) return fail(SDRHIP_EINVAL, "mem must be SDRHIP_MEM_HOST or SDRHIP_MEM_DEVICE");
    if (nframes == 0) {
        // an empty call decodes nothing; in pipelined mode it still delivers the batch that waits
        if (tx->pipelined && tx->late.have) return sdrhip_tx_flush(tx, iq_out, out_stride, n_out, mem);
        return SDRHIP_OK;
    }
    if (!rx || (!iq_out && !tx->pipelined)) return fail(SDRHIP_EINVAL, "tx_process: NULL buffer");
    sdrhip_ctx *c = tx->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int S = tx->nstreams;
    const size_t fb = (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE;
    if (S == 1) { rx_stride_bytes = nframes * fb; if (!tx->pipelined) out_stride = n_res; }
    if (rx_stride_bytes < nframes * fb || (!tx->pipelined && out_stride < n_res)) return fail(SDRHIP_EINVAL, "tx_process: stride too small");
    if (mem == SDRHIP_MEM_DEVICE && S > 1 && rx_stride_bytes != nframes * fb) return fail(SDRHIP_EINVAL, "tx_process: device rx must be contiguous per stream");
    const size_t pstride = (n_payload + 3) & ~(size_t)3; // samples
    int rc;

    if (tx->pipelined) {
        // ---- first stream: the batch that waits goes through the interpolator (enqueued FIRST: the long kernel takes the CUs, the
        // decoder's workgroups fill in as its waves retire); second stream: this call's batch is decoded beside it
        const bool overlap = c->opt.tx_overlap != 0;
        hipStream_t s2 = c->stream;
        if (overlap && (rc = ctx_stream2(c, &s2))) return rc;
        if (mem == SDRHIP_MEM_DEVICE && overlap) HIP_TRY(hipEventRecord(tx->ev_in, c->stream)); // (whatever produced rx on the caller's stream)
        if (tx->late.have) {
            if ((rc = tx_deliver_late(tx, iq_out, out_stride, n_out, mem))) return rc;
        }
        const int sel = tx->psel;
        DevBuf &pay = tx->payload[sel];
        if (pay.cap < (size_t)S * pstride * 4 + 16) {
            // (growing the buffer frees it: nothing may still read it)
            if (tx->itp_pending[sel]) { HIP_TRY(hipEventSynchronize(tx->ev_itp[sel])); tx->itp_pending[sel] = false; }
            if ((rc = pay.reserve((size_t)S * pstride * 4 + 16))) return rc;
        }
        const uint8_t *drx = rx;
        if (mem == SDRHIP_MEM_HOST) {
            if (overlap) HIP_TRY(hipStreamSynchronize(s2)); // (the previous decode may still read rxbuf; it ran beside the previous call's interpolator)
            if ((rc = tx->rxbuf.reserve((size_t)S * nframes * fb))) return rc;
            HIP_TRY(hipMemcpy2DAsync(tx->rxbuf.p, nframes * fb, rx, rx_stride_bytes, nframes * fb, S, hipMemcpyHostToDevice, s2));
            // (pinned caller memory makes this copy truly asynchronous, and it runs on the SECOND stream: the call must not return
            // before it has read `rx` -- the host-pointer contract is "the buffer is yours again when the call returns")
            if (overlap) HIP_TRY(hipEventRecord(tx->ev_up, s2));
            drx = tx->rxbuf.as<uint8_t>();
        } else if (overlap) {
            HIP_TRY(hipStreamWaitEvent(s2, tx->ev_in, 0));
        }
        if (overlap && tx->itp_pending[sel]) { HIP_TRY(hipStreamWaitEvent(s2, tx->ev_itp[sel], 0)); tx->itp_pending[sel] = false; }
        DecodeSide side;
        side.stream = s2; side.plan = &tx->plan_own; side.idx = &tx->idx_own; side.pin = &tx->pin_own;
        if ((rc = tx_decode(tx, drx, indices, nframes, pay, pstride, overlap ? &side : nullptr))) return rc;
        HIP_TRY(hipEventRecord(tx->ev_dec, s2));
        tx->late.have = true; tx->late.n_payload = n_payload; tx->late.pstride = pstride; tx->late.log2interp = tx->log2interp;
        tx->psel ^= 1;
        if (mem == SDRHIP_MEM_HOST) {
            HIP_TRY(hipStreamSynchronize(c->stream)); // (the delivered samples; the decode goes on)
            if (overlap) HIP_TRY(hipEventSynchronize(tx->ev_up)); // (... but the caller's rx has been read)
        }
        return SDRHIP_OK;
    }

    const uint8_t *drx = rx;
    if (mem == SDRHIP_MEM_HOST) {
        if ((rc = tx->rxbuf.reserve((size_t)S * nframes * fb))) return rc;
        HIP_TRY(hipMemcpy2DAsync(tx->rxbuf.p, nframes * fb, rx, rx_stride_bytes, nframes * fb, S, hipMemcpyHostToDevice, c->stream));
        drx = tx->rxbuf.as<uint8_t>();
    } else {
        if (!aligned16(iq_out) || (S > 1 && (out_stride & 3))) return fail(SDRHIP_EALIGN, "tx_process: device output must be 16-byte aligned");
    }
    if (tx_gather_applies(tx, tx->log2interp)) {
        // no-copy: decode (restored blocks + map only), then the interpolator reads the received frames through the map
        InterpGather g;
        if ((rc = tx_decode_gather(tx, drx, indices, nframes, &g))) return rc;
        if ((rc = tx_interpolate(tx, tx->log2interp, tx->payload[0], n_payload, pstride, iq_out, out_stride, mem, &g))) return rc;
        if (mem == SDRHIP_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
        return SDRHIP_OK;
    }
    // decode all S * nframes frames in one batch: payload [S][nframes][127 * 508] = [S][n_payload] samples
    if ((rc = tx->payload[0].reserve((size_t)S * pstride * 4 + 16))) return rc;
    if ((rc = tx_decode(tx, drx, indices, nframes, tx->payload[0], pstride, nullptr))) return rc;
    if ((rc = tx_interpolate(tx, tx->log2interp, tx->payload[0], n_payload, pstride, iq_out, out_stride, mem))) return rc;
    if (mem == SDRHIP_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    return SDRHIP_OK;
}

// --------------------------------------------------------------------------- asynchronous host-pointer Tx entry
// sdrdaemontx's chain is asynchronous as well: a reader thread keeps receiving super blocks while the main loop interpolates
// (sdrdaemontx.cpp:449-498), and SDRdaemonFECBuffer hands a frame out one frame late (SDRdaemonFECBuffer.cpp:133-139).
// sdrhip_tx_process on host pointers is upload + three launches + download + a synchronisation per call; submit / collect give
// the host-pointer path the reference's asynchrony: a batch of received frames goes out as ONE upload + decode + interpolate +
// download on the context's stream and the call returns; the samples (and the frames' meta blocks) are collected later, in order.
extern "C" int sdrhip_tx_set_async(sdrhip_tx *tx, int depth)
{
    if (!tx) return fail(SDRHIP_EINVAL, "tx is NULL");
    sdrhip::CtxLock lock_(tx->ctx);
    if (depth < 1 || depth > 64) return fail(SDRHIP_EINVAL, "tx_set_async: depth 1..64");
    for (auto &b : tx->abatch)
        if (b.state != 0) return fail(SDRHIP_EINVAL, "tx_set_async: batches are in flight: collect them first");
    for (auto &b : tx->abatch) { if (b.done) (void)hipEventDestroy(b.done); b.in.release(); b.din.release(); b.dout.release(); b.db0.release(); b.out.release(); }
    tx->abatch.assign((size_t)depth, sdrhip_tx::ABatch());
    tx->a_head = tx->a_tail = 0;
    return SDRHIP_OK;
}

extern "C" int sdrhip_tx_submit(sdrhip_tx *tx, const uint8_t *rx, const uint8_t *indices, size_t nframes, size_t rx_stride_bytes)
{
    if (!tx) return fail(SDRHIP_EINVAL, "tx is NULL");
    sdrhip::CtxLock lock_(tx->ctx);
    if (nframes == 0) return SDRHIP_OK;
    if (!rx) return fail(SDRHIP_EINVAL, "tx_submit: NULL input");
    if (tx->pipelined) return fail(SDRHIP_EINVAL, "tx_submit: the handle is in pipelined mode (sdrhip_tx_process delivers one call late there); use one or the other");
    if (tx->abatch.empty()) tx->abatch.assign(4, sdrhip_tx::ABatch());
    sdrhip_ctx *c = tx->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int S = tx->nstreams;
    const size_t fb = (size_t)SDRHIP_NB_ORIGINAL * SDRHIP_UDPSIZE, row = nframes * fb;
    if (S == 1) rx_stride_bytes = row;
    if (rx_stride_bytes < row) return fail(SDRHIP_EINVAL, "tx_submit: stride too small");
    sdrhip_tx::ABatch &b = tx->abatch[tx->a_tail % tx->abatch.size()];
    if (b.state == 2) return fail(SDRHIP_EBUSY, "tx_submit: every batch of the ring is in flight: sdrhip_tx_collect first");
    const size_t n_payload = nframes * SDRHIP_SAMPLES_PER_FRAME, n_res = n_payload << tx->log2interp;
    const size_t pstride = (n_payload + 3) & ~(size_t)3, dos = (n_res + 3) & ~(size_t)3;
    const size_t b0_bytes = (size_t)S * nframes * SDRHIP_BLOCK_BYTES;
    int rc;
    // everything that can fail for want of memory comes first
    if ((rc = b.din.reserve((size_t)S * row))) return rc;
    if ((rc = b.dout.reserve((size_t)S * dos * 4 + 16))) return rc;
    if ((rc = b.db0.reserve(b0_bytes))) return rc;
    if ((rc = b.out.reserve((size_t)S * dos * 4 + b0_bytes))) return rc;
    const bool gather = tx_gather_applies(tx, tx->log2interp);
    if (!gather && (rc = tx->payload[0].reserve((size_t)S * pstride * 4 + 16))) return rc;
    if (!b.done && hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) { b.done = nullptr; return fail(SDRHIP_EDEVICE, "hipEventCreate"); }
    const uint8_t *src = rx;
    size_t sstride = rx_stride_bytes;
    if (!host_is_pinned(rx, (size_t)(S - 1) * rx_stride_bytes + row)) {
        if ((rc = b.in.reserve((size_t)S * row))) return rc; // (waits for the upload of the batch that used this buffer last)
        for (int s = 0; s < S; ++s) memcpy(b.in.as<char>() + (size_t)s * row, rx + (size_t)s * rx_stride_bytes, row);
        src = b.in.as<uint8_t>(); sstride = row;
    }
    if (S == 1) HIP_TRY(hipMemcpyAsync(b.din.p, src, row, hipMemcpyHostToDevice, c->stream));
    else HIP_TRY(hipMemcpy2DAsync(b.din.p, row, src, sstride, row, S, hipMemcpyHostToDevice, c->stream));
    if (src != rx) b.in.mark(c->stream);
    if (gather) {
        // (the batch's received frames live in b.din until it is collected: the interpolator reads them in place)
        InterpGather g;
        if ((rc = tx_decode_gather(tx, b.din.as<uint8_t>(), indices, nframes, &g, b.db0.as<uint8_t>()))) return rc;
        if ((rc = interpolate_device(tx->itp, tx->log2interp, nullptr, n_payload, pstride, b.dout.as<int16_t>(), dos, nullptr, &g))) return rc;
    } else {
        if ((rc = tx_decode(tx, b.din.as<uint8_t>(), indices, nframes, tx->payload[0], pstride, nullptr, b.db0.as<uint8_t>()))) return rc;
        if ((rc = interpolate_device(tx->itp, tx->log2interp, tx->payload[0].as<int16_t>(), n_payload, pstride, b.dout.as<int16_t>(), dos, nullptr))) return rc;
    }
    // (from here on the interpolator's state has advanced: a failure loses the batch, it is never replayed)
    hipError_t e = hipMemcpyAsync(b.out.p, b.dout.p, (size_t)S * dos * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b.out.as<char>() + (size_t)S * dos * 4, b.db0.p, b0_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipEventRecord(b.done, c->stream);
    if (e != hipSuccess) return fail(SDRHIP_EDEVICE, "tx batch download: %s (the batch's %zu frames per stream are lost)", hipGetErrorString(e), nframes);
    b.nframes = nframes; b.n_res = n_res; b.dos = dos;
    b.state = 2;
    ++tx->a_tail;
    return SDRHIP_OK;
}

extern "C" int sdrhip_tx_collect(sdrhip_tx *tx, int16_t *iq_out, size_t out_stride, size_t max_samples, uint8_t *block0_out, size_t *n_out, size_t *n_frames, int wait)
{
    if (!tx || !n_out) return fail(SDRHIP_EINVAL, "tx_collect: NULL argument");
    // (the wait happens outside the context lock: the submitting thread -- the reference's reader thread -- keeps feeding the ring)
    std::unique_lock<std::recursive_mutex> lock_(tx->ctx->mtx);
    *n_out = 0;
    if (n_frames) *n_frames = 0;
    if (tx->abatch.empty()) return fail(SDRHIP_EBUSY, "tx_collect: nothing was submitted");
    HIP_TRY(hipSetDevice(tx->ctx->device));
    sdrhip_tx::ABatch *bp = nullptr;
    for (;;) {
        sdrhip_tx::ABatch &h = tx->abatch[tx->a_head % tx->abatch.size()];
        if (h.state == 0) return fail(SDRHIP_EBUSY, "tx_collect: nothing was submitted");
        const hipError_t q = hipEventQuery(h.done);
        if (q == hipSuccess) { bp = &h; break; }
        if (q != hipErrorNotReady) return fail(SDRHIP_EDEVICE, "hipEventQuery: %s", hipGetErrorString(q));
        if (!wait) return fail(SDRHIP_EBUSY, "tx_collect: the oldest batch is still in flight");
        const size_t head = tx->a_head;
        hipEvent_t ev = h.done;
        lock_.unlock();
        const hipError_t w = hipEventSynchronize(ev);
        lock_.lock();
        if (w != hipSuccess) return fail(SDRHIP_EDEVICE, "hipEventSynchronize: %s", hipGetErrorString(w));
        if (tx->a_head == head) { bp = &tx->abatch[head % tx->abatch.size()]; break; }
    }
    sdrhip_tx::ABatch &b = *bp;
    const int S = tx->nstreams;
    if (b.n_res > max_samples) { // (the batch stays where it is: call again with room for *n_out samples per stream)
        *n_out = b.n_res;
        if (n_frames) *n_frames = b.nframes;
        return fail(SDRHIP_EINVAL, "tx_collect: the batch holds %zu samples per stream, iq_out has room for %zu", b.n_res, max_samples);
    }
    if (b.n_res) {
        if (!iq_out) return fail(SDRHIP_EINVAL, "tx_collect: NULL iq_out");
        if (S == 1) out_stride = b.n_res;
        if (out_stride < b.n_res) return fail(SDRHIP_EINVAL, "tx_collect: out_stride too small");
        for (int s = 0; s < S; ++s) memcpy(iq_out + (size_t)s * out_stride * 2, b.out.as<char>() + (size_t)s * b.dos * 4, b.n_res * 4);
    }
    if (block0_out) memcpy(block0_out, b.out.as<char>() + (size_t)S * b.dos * 4, (size_t)S * b.nframes * SDRHIP_BLOCK_BYTES);
    *n_out = b.n_res;
    if (n_frames) *n_frames = b.nframes;
    b.state = 0;
    ++tx->a_head;
    return SDRHIP_OK;
}
