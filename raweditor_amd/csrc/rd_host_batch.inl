// rd_host_batch.inl -- part of librawdev.so's host side: included by rawdev.hip (one translation unit; the kernels are
// templates in rd_kernels.h).  rd_batch (multi-frame launches), rd_node_batch (one process, N devices, RCCL histogram all-reduce), rd_exporter
// (pinned ring): the batch export the north-star adds (BASELINE.json configs 3-5; no reference counterpart).

// ------------------------------------------------------------------------------------------------
// rd_batch
// ------------------------------------------------------------------------------------------------
struct rd_batch {
    int device = 0;
    uint32_t w = 0, h = 0, fmt = 0;
    bool hist = false;
    bool identity_ok = false;
    uint32_t math_mode = RD_MATH_STRICT;
    rd_launch_cfg cfg;
    uint32_t blocks = 0;                       // fixed grid: slab rows stay aligned across launches
    // RD_BATCH_STREAMS=2: the launches of a call alternate between the caller's stream and an internal one (forked from and
    // joined back into the caller's stream inside rd_batch_develop).  Concurrent launches need their own slab rows and ticket
    // counters.  Round 1 (one launch per 24 MP frame, two full-size grids): +2.4 % strict / -2 % contracted, not the default.
    // Round 6: with two lanes every launch takes ONE of the CU's two workgroup slots -- the lanes' launches then run side by
    // side instead of queueing for slots the other one holds, a lane's next launch moves into the slots its previous one
    // frees, and there is no chip-wide drain between launches: BASELINE config 5 as worded (8 row-band launches per 100 MP
    // frame) 248.5 -> 234.1 us per frame (profiles/r06_c5_tiled_ab.txt).
    uint32_t n_streams = 1;
    hipStream_t aux[1] = { nullptr };
    hipEvent_t ev_fork = nullptr, ev_join[1] = { nullptr };
    unsigned long long *slab64 = nullptr;      // n_streams x blocks x 768
    rd_scratch scratch;                        // per stream: ticket counters
    // Multi-frame launches (the default; RD_BATCH_PERSISTENT=0 falls back to one launch per frame / row band): the
    // frames of a call reach the kernel as an array of descriptors in HBM.  Two arrays with pinned staging; an array is
    // rewritten only when the caller's frames differ from what it holds (bench.py re-submits the same batch every
    // step), and only after the launches that read it have finished (`done`).
    bool persistent = true;
    uint32_t max_frames = 8;                   // RD_BATCH_MAX_FRAMES: frames per launch (default 8 for f32, 32 otherwise)
    struct desc_buf {
        rd_frame_desc *dev = nullptr, *host = nullptr;
        size_t cap = 0, n = 0;
        hipEvent_t done = nullptr;             // after the last launch that reads the array
        hipStream_t done_on = nullptr;         // ... and the stream that recorded it (another stream's call chains behind it, below)
        bool done_set = false;
        hipEvent_t uploaded = nullptr;         // after the copy that filled it (a later call may come on another stream)
        bool valid = false;
    } db[2];
    hipStream_t up = nullptr;                  // the descriptor uploads' own stream (round 6): a copy need not queue behind the launches
                                               // already on the caller's stream -- it only has to land before THIS call's first launch
    int db_last = 1;
    uint32_t last_launches = 0;                // fused launches enqueued by the last rd_batch_develop call
    // Launch timing (measurement aid, rd_batch_set_launch_timing): a HIP event pair around every fused launch of the last
    // `timing_keep` develop calls, oldest first.  Off (0) by default: the pairs put two barrier packets between launches.
    struct timed_launch { hipEvent_t start = nullptr, end = nullptr; uint32_t call = 0; };
    uint32_t timing_keep = 0, timing_call = 0;
    std::vector<timed_launch> timeline;
    std::vector<hipEvent_t> ev_free;
};

// One timing event: a recycled one, or a new one (nullptr when the runtime refuses).
static hipEvent_t rd_batch_timing_event(rd_batch *b)
{
    if (!b->ev_free.empty()) { hipEvent_t e = b->ev_free.back(); b->ev_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

// A timed develop call starts: forget the launches of calls that fell out of the window (their events are reused).
static void rd_batch_timing_begin(rd_batch *b)
{
    if (!b->timing_keep) return;
    b->timing_call += 1u;
    size_t drop = 0;
    while (drop < b->timeline.size() && b->timeline[drop].call + b->timing_keep <= b->timing_call) ++drop;
    for (size_t i = 0; i < drop; ++i) { b->ev_free.push_back(b->timeline[i].start); b->ev_free.push_back(b->timeline[i].end); }
    b->timeline.erase(b->timeline.begin(), b->timeline.begin() + (ptrdiff_t)drop);
}

struct rd_batch_timed {                        // RAII around ONE launch on stream s: start event now, end event on scope exit
    rd_batch *b; hipStream_t s; rd_batch::timed_launch tl; bool on = false;
    rd_batch_timed(rd_batch *bb, hipStream_t ss) : b(bb), s(ss)
    {
        if (!b->timing_keep) return;
        tl.start = rd_batch_timing_event(b); tl.end = rd_batch_timing_event(b); tl.call = b->timing_call;
        on = tl.start && tl.end && hipEventRecord(tl.start, s) == hipSuccess;
        if (!on) { if (tl.start) (void)hipEventDestroy(tl.start); if (tl.end) (void)hipEventDestroy(tl.end); }
    }
    ~rd_batch_timed()                          // (a destructor: nothing may leave it)
    {
        if (!on) return;
        (void)hipEventRecord(tl.end, s);
        try {
            b->timeline.push_back(tl);         // capacity reserved by rd_batch_set_launch_timing for 64 launches per call
        } catch (...) {                        // beyond it a failed growth costs this launch its record, nothing else
            (void)hipEventDestroy(tl.start);
            (void)hipEventDestroy(tl.end);
        }
    }
};

extern "C" int rd_batch_create(int device, uint32_t w, uint32_t h, uint32_t fmt, uint32_t with_histogram,
                               rd_batch **out) try
{
    RD_ENTRY(rd_batch_create);
    if (!out) return rd_fail(RD_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!w || !h) return rd_fail(RD_ERR_INVALID_ARG, "empty frame %ux%u", w, h);
    if (!rd_format_bytes_per_pixel(fmt)) return rd_fail(RD_ERR_INVALID_ARG, "unknown format %u", fmt);
    if (fmt == RD_FMT_RGB_U8 && w < 128u) return rd_fail(RD_ERR_UNSUPPORTED, "RGB8 batch export needs a frame at least 128 pixels wide (got %u)", w);
    const uint64_t items = (uint64_t)(h / 2u + 1u) * rd_tiles_per_unit(w, fmt) * 64u;
    if (items >= 0xffffffffull) return rd_fail(RD_ERR_UNSUPPORTED, "frame %ux%u too large", w, h);
    int n_cu = 0;
    int rc = rd_check_device(device, &n_cu);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    rc = rd_q8_lut_ensure(device);
    if (rc == RD_OK && fmt == RD_FMT_RGBA_F16) rc = rd_f16_lut_ensure(device);
    if (rc) return rc;
    rd_batch *b = new (std::nothrow) rd_batch;
    if (!b) return rd_fail(RD_ERR_OOM, "host allocation failed");
    struct undo { void operator()(rd_batch *q) const { rd_batch_destroy(q); } };
    std::unique_ptr<rd_batch, undo> hold(b);                     // whatever fails or throws below: the half-built batch is destroyed
    b->device = device; b->w = w; b->h = h; b->fmt = fmt; b->hist = with_histogram != 0;
    b->cfg.n_cu = n_cu;
    b->cfg.wg_per_cu_plain = rd_env_u32("RD_WG_PER_CU", 2);
    b->cfg.wg_per_cu_hist = rd_env_u32("RD_WG_PER_CU_HIST", 2);      // experiment builds with a smaller RD_BLOCK only
    b->identity_ok = rd_identity_map(w) && rd_identity_map(h);
    if (!b->identity_ok) return rd_fail(RD_ERR_UNSUPPORTED, "export map is not the identity for %ux%u", w, h);
    // RD_BATCH_STREAMS=2: two lanes for one-launch-per-frame / row-band calls (implies RD_BATCH_PERSISTENT=0, as ever).  (Tried in
    // round 6 and dropped, profiles/r06_c5_tiled_ab.txt: four lanes with half-size grids, 317 against 234 us per 100 MP frame; two
    // lanes for the MULTI-FRAME launches of the headline, 78.7-79.2 against 78.5-79.0 us per frame: nothing to gain, its
    // launches follow each other without a gap as it is.)
    b->n_streams = rd_env_u32("RD_BATCH_STREAMS", 1) >= 2 ? 2u : 1u;
    if (b->n_streams > 1 && !getenv("RD_WG_PER_CU") && !getenv("RD_WG_PER_CU_HIST"))
        b->cfg.wg_per_cu_plain = b->cfg.wg_per_cu_hist = 1;       // the two lanes share the CU's two workgroup slots: one each
    b->blocks = rd_blocks_for(b->cfg, items, b->hist);
    {
        const char *pe = getenv("RD_BATCH_PERSISTENT");
        b->persistent = !(pe && *pe == '0') && b->n_streams == 1;
        // f32: 8 is the flat bottom of the curve; the narrow surfaces are arithmetic-bound and only
        // lose launch tails as launches grow (u8 48.7 / 48.6 / 48.2, f16 61.2 / 60.6 / 60.2 us per frame at 8 / 16 / 32)
        b->max_frames = rd_env_u32("RD_BATCH_MAX_FRAMES", fmt == RD_FMT_RGBA_F32 ? 8u : 32u);
    }
    hipError_t e = hipSuccess;
    for (int j = 0; j < 2 && e == hipSuccess && b->persistent; ++j) {
        e = hipEventCreateWithFlags(&b->db[j].done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->db[j].uploaded, hipEventDisableTiming);
    }
    if (e == hipSuccess && b->persistent) e = hipStreamCreateWithFlags(&b->up, hipStreamNonBlocking);
    if (b->n_streams > 1 && e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming);
    for (uint32_t k = 0; k + 1u < b->n_streams && e == hipSuccess; ++k) {
        e = hipStreamCreateWithFlags(&b->aux[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_join[k], hipEventDisableTiming);
    }
    if (e == hipSuccess && b->hist) {
        const size_t bytes = (size_t)b->n_streams * b->blocks * 768 * sizeof(unsigned long long);
        e = hipMalloc((void **)&b->slab64, bytes);
        if (e == hipSuccess) e = hipMemset(b->slab64, 0, bytes);
    }
    if (e != hipSuccess) return rd_fail(e == hipErrorOutOfMemory ? RD_ERR_OOM : RD_ERR_HIP, "batch resources: %s", hipGetErrorString(e));
    *out = hold.release();
    return RD_OK;
}
RD_CATCH_INT(rd_batch_create)

extern "C" void rd_batch_destroy(rd_batch *b) try
{
    if (!b) return;
    {
        rd_devguard g(b->device);
        (void)hipDeviceSynchronize();        // launches on the caller's streams still use the slab and the tickets
        if (b->slab64) (void)hipFree(b->slab64);
        b->scratch.release();
        for (auto &d : b->db) {
            if (d.dev) (void)hipFree(d.dev);
            if (d.host) (void)hipHostFree(d.host);
            if (d.done) (void)hipEventDestroy(d.done);
            if (d.uploaded) (void)hipEventDestroy(d.uploaded);
        }
        for (auto &t : b->timeline) { if (t.start) (void)hipEventDestroy(t.start); if (t.end) (void)hipEventDestroy(t.end); }
        for (hipEvent_t e : b->ev_free) if (e) (void)hipEventDestroy(e);
        if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
        if (b->up) (void)hipStreamDestroy(b->up);
        for (hipEvent_t ev : b->ev_join) if (ev) (void)hipEventDestroy(ev);
        for (hipStream_t a : b->aux) if (a) (void)hipStreamDestroy(a);
    }
    delete b;
}
RD_CATCH_VOID(rd_batch_destroy)

extern "C" int rd_batch_set_math_mode(rd_batch *b, uint32_t mode) try
{
    RD_ENTRY(rd_batch_set_math_mode);
    if (!b) return rd_fail(RD_ERR_INVALID_ARG, "NULL batch");
    if (mode != RD_MATH_STRICT && mode != RD_MATH_CONTRACTED) return rd_fail(RD_ERR_INVALID_ARG, "unknown math mode %u", mode);
    b->math_mode = mode;
    return RD_OK;
}
RD_CATCH_INT(rd_batch_set_math_mode)

// How many frames one multi-frame launch may hold for frames of w x h: the 32-bit tile index, the u32 histogram bins a
// workgroup keeps in LDS for the whole launch (every pixel of the launch could, in principle, land in one bin of one
// workgroup), and the per-format default / RD_BATCH_MAX_FRAMES cap.
static uint64_t rd_frames_per_launch_limit(uint32_t w, uint32_t h, uint32_t fmt, bool hist, uint32_t cap)
{
    const uint32_t tpu = rd_tiles_per_unit(w, fmt);
    const uint64_t tpf = (uint64_t)(h / 2u + 1u) * tpu;
    uint64_t kmax = tpf ? 0xfffffffeull / tpf : 4096u;           // (a frame ONE pixel wide has no quad, so no tile at all)
    if (hist) { const uint64_t k2 = 0xffffffffull / ((uint64_t)w * h); if (k2 < kmax) kmax = k2; }
    if (kmax > 4096) kmax = 4096;
    if (cap && cap < kmax) kmax = cap;
    return kmax < 1 ? 1 : kmax;
}

// Frames of the next launch, starting at frame i0: as many consecutive frames as the limit allows whose surfaces
// ([out, out + surf_bytes)) overlap none of the launch's earlier ones.
static size_t rd_next_launch_size(const rd_frame *frames, size_t n, size_t i0, size_t surf_bytes, uint64_t kmax)
{
    size_t c = 1;
    for (; i0 + c < n && c < kmax; ++c) {
        const uintptr_t o = (uintptr_t)frames[i0 + c].out_dev;
        bool clash = false;
        for (size_t k = 0; k < c && !clash; ++k) {
            const uintptr_t p = (uintptr_t)frames[i0 + k].out_dev;
            clash = o < p + surf_bytes && p < o + surf_bytes;
        }
        if (clash) break;
    }
    return c;
}

// No device needed: the launches rd_batch_develop would cut a call into (frames per launch, in order).  Returns the number
// of launches, or a negative rd_status; at most `counts_cap` entries are written.
extern "C" int rd_batch_plan_launches(uint32_t width, uint32_t height, uint32_t format, uint32_t with_histogram,
                                      const rd_frame *frames, size_t n_frames, uint32_t max_frames, uint32_t *counts,
                                      size_t counts_cap) try
{
    RD_ENTRY(rd_batch_plan_launches);
    const size_t bpp = rd_format_bytes_per_pixel(format);
    if (!width || !height || !bpp || (!frames && n_frames)) return rd_fail(RD_ERR_INVALID_ARG, "rd_batch_plan_launches: bad argument");
    const uint32_t cap = max_frames ? max_frames : (format == RD_FMT_RGBA_F32 ? 8u : 32u);
    const uint64_t kmax = rd_frames_per_launch_limit(width, height, format, with_histogram != 0, cap);
    const size_t surf = (size_t)width * height * bpp;
    int launches = 0;
    for (size_t i0 = 0; i0 < n_frames;) {
        const size_t c = rd_next_launch_size(frames, n_frames, i0, surf, kmax);
        if (counts && (size_t)launches < counts_cap) counts[launches] = (uint32_t)c;
        ++launches;
        i0 += c;
    }
    return launches;
}
RD_CATCH_INT(rd_batch_plan_launches)

// The multi-frame path of rd_batch_develop: descriptors -> HBM (only when they changed), then as few launches as the
// limits allow.  A launch never holds two frames whose surfaces overlap (the order in which the tiles of DIFFERENT frames
// are stored inside one launch is not defined), never more pixels than a u32 histogram bin can count, and never more
// tiles than the 32-bit tile index.  Row bands need no launches of their own here: the ticket front sweeps a frame in
// row order, so a "band" is a range of tickets.
static int rd_batch_develop_multi(rd_batch *b, const rd_frame *frames, size_t n, hipStream_t s, bool probe = false,
                                  uint32_t *stamps = nullptr)
{
    if (!n) return RD_OK;
    const size_t bpp = rd_format_bytes_per_pixel(b->fmt);
    const size_t surf_bytes = (size_t)b->w * b->h * bpp;
    static thread_local std::vector<rd_frame_desc> tmp;
    RD_FAULT_POINT("batch.descs");
    tmp.resize(n);
    memset(tmp.data(), 0, n * sizeof(rd_frame_desc));
    bool aligned16 = true;
    for (size_t f = 0; f < n; ++f) {
        const rd_frame &fr = frames[f];
        if (!fr.cfa_dev || !fr.out_dev) return rd_fail(RD_ERR_INVALID_ARG, "frame %zu: NULL device pointer", f);
        if ((uintptr_t)fr.cfa_dev % 4u) return rd_fail(RD_ERR_INVALID_ARG, "frame %zu: cfa_dev not 4-byte aligned", f);
        if ((uintptr_t)fr.out_dev % rd_align_for(b->fmt)) return rd_fail(RD_ERR_INVALID_ARG, "frame %zu: out_dev misaligned", f);
        if ((uintptr_t)fr.cfa_dev % 16u) aligned16 = false;
        tmp[f].cfa = fr.cfa_dev;
        tmp[f].out = fr.out_dev;
        if (fr.matrix_layout != RD_MATRIX_REFERENCE && fr.matrix_layout != RD_MATRIX_ROW_MAJOR)
            return rd_fail(RD_ERR_INVALID_ARG, "frame %zu: unknown matrix layout %u", f, fr.matrix_layout);
        tmp[f].u = rd_frame_ku(fr.params, fr.wb_multipliers, fr.color_matrix, 1.0f, 0.0f, 0.0f, fr.black_level, b->math_mode, fr.matrix_layout);
        static const bool no_elide = rd_env_u32("RD_NO_ELIDE", 0) != 0;
        if (no_elide) tmp[f].u.elide = 0u;
    }
    // descriptor array in HBM: reuse, or rewrite the one not used by the previous call
    int j = -1;
    for (int k = 0; k < 2; ++k)
        if (b->db[k].valid && b->db[k].n == n && memcmp(b->db[k].host, tmp.data(), n * sizeof(rd_frame_desc)) == 0) j = k;
    if (j < 0) {
        j = b->db_last ^ 1;
        rd_batch::desc_buf &d = b->db[j];
        RD_HIP(hipEventSynchronize(d.done));                 // launches that read this array (two calls ago) have finished
        d.valid = false;
        if (d.cap < n) {
            if (d.dev) { (void)hipFree(d.dev); d.dev = nullptr; }
            if (d.host) { (void)hipHostFree(d.host); d.host = nullptr; }
            d.cap = 0;
            const size_t cap = n < 64 ? 64 : n;
            RD_HIP(hipMalloc((void **)&d.dev, cap * sizeof(rd_frame_desc)));
            RD_HIP(hipHostMalloc((void **)&d.host, cap * sizeof(rd_frame_desc), hipHostMallocDefault));
            d.cap = cap;
        }
        memcpy(d.host, tmp.data(), n * sizeof(rd_frame_desc));
        // on the batch's own upload stream: the array is free (its readers finished: `done` above), so the copy starts at once,
        // under whatever launches of the previous call are still queued on s, instead of sitting between two kernels of s
        // (a copy-engine transfer there is a bubble of ~15 us per call: 0.5 % of a 64-frame narrow batch, HISTORY round 5)
        static const bool inline_upload = rd_env_u32("RD_DESC_UPLOAD_INLINE", 0) != 0;       // A/B: the copy on s itself, as through round 5
        hipStream_t us = (b->up && !inline_upload) ? b->up : s;
        RD_HIP(hipMemcpyAsync(d.dev, d.host, n * sizeof(rd_frame_desc), hipMemcpyHostToDevice, us));
        RD_HIP(hipEventRecord(d.uploaded, us));
        d.n = n;
        d.valid = true;
    }
    RD_HIP(hipStreamWaitEvent(s, b->db[j].uploaded, 0));         // (a reused array's copy may have been enqueued by a call on another stream)
    // `done` is ONE event: re-recorded on this stream it would forget a reader still running on another stream, and the rewrite two
    // calls on would wait for the wrong one.  A call that reuses an array from another stream therefore queues behind that reader.
    if (b->db[j].done_set && b->db[j].done_on != s) RD_HIP(hipStreamWaitEvent(s, b->db[j].done, 0));
    b->db_last = j;
    const rd_frame_desc *descs = b->db[j].dev;

    const uint64_t kmax = rd_frames_per_launch_limit(b->w, b->h, b->fmt, b->hist, b->max_frames);
    if ((probe || stamps) && !rd_probe_launchable(b->w, b->h, aligned16))
        return rd_fail(RD_ERR_UNSUPPORTED, "the diagnostic instances exist for the read-burst kernel only: frames at least 128 pixels wide, "
                                           "16-byte aligned CFA planes, at least 1 MB of CFA rows");
    const rd_scratch::lease l = b->scratch.get(s, false);
    if (l.idx < 0) return rd_fail(RD_ERR_OOM, "scheduler state allocation failed");
    uint32_t *tq = l.tq;
    int rc = RD_OK;
    b->last_launches = 0;
    rd_batch_timing_begin(b);
    (void)hipGetLastError();                     // see rd_enqueue_render
    for (size_t i0 = 0; i0 < n && rc == RD_OK;) {
        const size_t c = rd_next_launch_size(frames, n, i0, surf_bytes, kmax);
        hipError_t e;
        {
            rd_batch_timed timed(b, s);
            if (probe) rd_launch_batch_t<RD_FMT_RGBA_F32, false, RD_MATH_PROBE>(descs + i0, (uint32_t)c, b->w, b->h, b->blocks, aligned16, b->slab64, tq, s);
            else RD_DISPATCH(rd_launch_batch_t, b->fmt, b->hist, b->math_mode, descs + i0, (uint32_t)c, b->w, b->h, b->blocks, aligned16,
                             b->slab64, tq, s, stamps);
            e = hipGetLastError();
        }
        if (e != hipSuccess) rc = rd_fail(RD_ERR_HIP, "multi-frame launch failed: %s", hipGetErrorString(e));
        else b->last_launches += 1;
        i0 += c;
    }
    b->scratch.used(l, s, rc != RD_OK);
    RD_HIP(hipEventRecord(b->db[j].done, s));
    b->db[j].done_on = s;
    b->db[j].done_set = true;
    return rc;
}

extern "C" int rd_batch_develop(rd_batch *b, const rd_frame *frames, size_t n, uint32_t row_bands, void *stream) try
{
    RD_ENTRY(rd_batch_develop);
    if (!b || (!frames && n)) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    rd_devguard g(b->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", b->device);
    if (b->persistent) return rd_batch_develop_multi(b, frames, n, (hipStream_t)stream);
    const uint32_t units = b->h / 2u + 1u;
    uint32_t bands = row_bands ? row_bands : 1u;
    if (bands > units) bands = units;
    hipStream_t lanes[2] = { (hipStream_t)stream, (hipStream_t)stream };
    const bool fork = b->n_streams > 1 && (uint64_t)n * bands > 1u;
    const uint32_t nl = fork ? b->n_streams : 1u;
    if (fork) {
        RD_HIP(hipEventRecord(b->ev_fork, lanes[0]));
        for (uint32_t k = 1; k < nl; ++k) {
            RD_HIP(hipStreamWaitEvent(b->aux[k - 1u], b->ev_fork, 0));
            lanes[k] = b->aux[k - 1u];
        }
    }
    int rc = RD_OK;
    size_t launch = 0;
    b->last_launches = 0;
    rd_batch_timing_begin(b);
    rd_scratch::lease ls[2];
    for (uint32_t k = 0; k < nl; ++k) {
        ls[k] = b->scratch.get(lanes[k], false);
        if (ls[k].idx < 0) return rd_fail(RD_ERR_OOM, "scheduler state allocation failed");
    }
    for (size_t f = 0; f < n && rc == RD_OK; ++f) {
        const rd_frame &fr = frames[f];
        if (!fr.cfa_dev || !fr.out_dev) { rc = rd_fail(RD_ERR_INVALID_ARG, "frame %zu: NULL device pointer", f); break; }
        if ((uintptr_t)fr.cfa_dev % 4u) { rc = rd_fail(RD_ERR_INVALID_ARG, "frame %zu: cfa_dev not 4-byte aligned", f); break; }
        if ((uintptr_t)fr.out_dev % rd_align_for(b->fmt)) { rc = rd_fail(RD_ERR_INVALID_ARG, "frame %zu: out_dev misaligned", f); break; }
        if (fr.matrix_layout != RD_MATRIX_REFERENCE && fr.matrix_layout != RD_MATRIX_ROW_MAJOR) { rc = rd_fail(RD_ERR_INVALID_ARG, "frame %zu: unknown matrix layout %u", f, fr.matrix_layout); break; }
        const rd_ku u = rd_frame_ku(fr.params, fr.wb_multipliers, fr.color_matrix, 1.0f, 0.0f, 0.0f, fr.black_level,
                                    b->math_mode, fr.matrix_layout);
        for (uint32_t k = 0; k < bands && rc == RD_OK; ++k, ++launch) {
            const uint32_t u0 = (uint32_t)(((uint64_t)units * k) / bands);
            const uint32_t u1 = (uint32_t)(((uint64_t)units * (k + 1)) / bands);
            const size_t lane = launch % nl;
            unsigned long long *slab = b->slab64 ? b->slab64 + lane * (size_t)b->blocks * 768u : nullptr;
            rd_batch_timed timed(b, lanes[lane]);
            rc = rd_enqueue_render(b->cfg, fr.cfa_dev, b->w, b->h, b->w, b->h, b->fmt, fr.out_dev, u, true, u0, u1,
                                   b->hist, b->math_mode, nullptr, slab, b->blocks, ls[lane].tq, lanes[lane], nullptr);
            if (rc == RD_OK) b->last_launches += 1;
        }
    }
    for (uint32_t k = 0; k < nl; ++k) b->scratch.used(ls[k], lanes[k], rc != RD_OK);
    for (uint32_t k = 1; k < nl; ++k) {          // join even after an error: what was enqueued stays ordered
        RD_HIP(hipEventRecord(b->ev_join[k - 1u], lanes[k]));
        RD_HIP(hipStreamWaitEvent(lanes[0], b->ev_join[k - 1u], 0));
    }
    return rc;
}
RD_CATCH_INT(rd_batch_develop)

extern "C" uint32_t rd_batch_last_launch_count(const rd_batch *b) try { RD_ENTRY(rd_batch_last_launch_count); return b ? b->last_launches : 0u; } RD_CATCH_VAL(rd_batch_last_launch_count, 0)

// Measurement aid: the launches rd_batch_develop would enqueue for these frames, with the kernel's arithmetic removed
// (rd_kernels.h, RD_MATH_PROBE): every load, sweep, ticket, LDS stage and store of the f32 export kernel on the caller's
// own planes and surfaces, which receive the raw samples as floats -- NOT a develop.  What the memory pattern alone costs
// on this box, in these buffers: the honest ceiling of rd_develop_batch here (bench.py: roofline.box_pattern_GBps).
extern "C" int rd_batch_probe_pattern(rd_batch *b, const rd_frame *frames, size_t n, void *stream) try
{
    RD_ENTRY(rd_batch_probe_pattern);
    if (!b || (!frames && n)) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (b->fmt != RD_FMT_RGBA_F32 || !b->persistent)
        return rd_fail(RD_ERR_UNSUPPORTED, "the pattern probe exists for the RGBA-f32 surface with multi-frame launches only");
    rd_devguard g(b->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", b->device);
    return rd_batch_develop_multi(b, frames, n, (hipStream_t)stream, true);
}
RD_CATCH_INT(rd_batch_probe_pattern)

// Measurement aid: the clock the part holds UNDER the export kernel.  One ordinary rd_batch_develop of these frames (the
// surfaces are developed, the histogram counts them) through the one instance of the kernel that stamps the shader-cycle
// counter and the 100 MHz real-time counter per workgroup, then a synchronise: cycles / ticks x 100 MHz of the LAST launch,
// over its workgroups.  RGBA-f32 + histogram + strict arithmetic + multi-frame launches only (the headline's instance).
extern "C" int rd_batch_measure_clock(rd_batch *b, const rd_frame *frames, size_t n, void *stream, double *ghz_median,
                                      double *ghz_min, double *ghz_max, double *workgroup_us_median) try
{
    RD_ENTRY(rd_batch_measure_clock);
    if (!b || !frames || !n) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (b->fmt != RD_FMT_RGBA_F32 || !b->hist || !b->persistent || b->math_mode != RD_MATH_STRICT)
        return rd_fail(RD_ERR_UNSUPPORTED, "the stamped instance exists for the RGBA-f32 surface with histogram, strict arithmetic and "
                                           "multi-frame launches only");
    rd_devguard g(b->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", b->device);
    // (the stamp buffer is released only after the stream that writes it has been waited for -- also when something below throws)
    struct res {
        uint32_t *dev = nullptr; hipStream_t s = nullptr;
        ~res() { if (dev) { (void)hipStreamSynchronize(s); (void)hipFree(dev); } }
    } r;
    r.s = (hipStream_t)stream;
    const size_t bytes = (size_t)b->blocks * 2u * sizeof(uint32_t);
    std::vector<uint32_t> st((size_t)b->blocks * 2u);            // host side first: nothing is in flight if this fails
    std::vector<double> ghz, us;
    ghz.reserve(b->blocks); us.reserve(b->blocks);
    RD_HIP(hipMalloc((void **)&r.dev, bytes));
    RD_HIP(hipMemsetAsync(r.dev, 0, bytes, (hipStream_t)stream));
    int rc = rd_batch_develop_multi(b, frames, n, (hipStream_t)stream, false, r.dev);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);     // before r.dev goes, whatever rc says
    if (rc) return rc;
    RD_HIP(e);
    RD_HIP(hipMemcpy(st.data(), r.dev, bytes, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < b->blocks; ++k)
        if (st[2u * k] && st[2u * k + 1u]) { ghz.push_back((double)st[2u * k] / (double)st[2u * k + 1u] * 0.1); us.push_back((double)st[2u * k + 1u] * 0.01); }
    if (ghz.empty()) return rd_fail(RD_ERR_HIP, "no workgroup left a clock stamp");
    std::sort(ghz.begin(), ghz.end());
    std::sort(us.begin(), us.end());
    if (ghz_median) *ghz_median = ghz[ghz.size() / 2];
    if (ghz_min) *ghz_min = ghz.front();
    if (ghz_max) *ghz_max = ghz.back();
    if (workgroup_us_median) *workgroup_us_median = us[us.size() / 2];
    return RD_OK;
}
RD_CATCH_INT(rd_batch_measure_clock)

// Measurement aid: keep a HIP event pair around every fused launch of the last `keep_calls` rd_batch_develop /
// rd_batch_probe_pattern calls (0 = off, the default; at most 64 calls).
extern "C" int rd_batch_set_launch_timing(rd_batch *b, uint32_t keep_calls) try
{
    RD_ENTRY(rd_batch_set_launch_timing);
    if (!b) return rd_fail(RD_ERR_INVALID_ARG, "NULL batch");
    if (keep_calls > 64u) return rd_fail(RD_ERR_INVALID_ARG, "at most 64 calls can be kept (got %u)", keep_calls);
    rd_devguard g(b->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", b->device);
    b->timeline.reserve((size_t)keep_calls * 64u + 64u);
    // (room for every event before the loop: no growth, so nothing can throw between an event's two homes)
    b->ev_free.reserve(std::max((size_t)keep_calls * 128u + 128u, b->ev_free.size() + 2u * b->timeline.size()));
    for (auto &t : b->timeline) { b->ev_free.push_back(t.start); b->ev_free.push_back(t.end); }
    b->timeline.clear();
    b->timing_keep = keep_calls;
    b->timing_call = 0;
    return RD_OK;
}
RD_CATCH_INT(rd_batch_set_launch_timing)

// The kept launches, oldest first, after the caller has synchronised the stream(s): start and end of each launch in
// microseconds since the FIRST kept launch's start event, and the call it belongs to (0 = the oldest kept call).  Writes at
// most `cap` entries, *n_out = how many are kept.  A launch's duration is end - start; the idle time before it is its start
// minus its predecessor's end (across a call boundary that gap holds the histogram fold and the descriptor upload).
extern "C" int rd_batch_launch_timeline(rd_batch *b, float *start_us, float *end_us, uint32_t *call_index, size_t cap,
                                        uint32_t *n_out) try
{
    RD_ENTRY(rd_batch_launch_timeline);
    if (!b || !n_out) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    *n_out = (uint32_t)b->timeline.size();
    if (b->timeline.empty()) return RD_OK;
    rd_devguard g(b->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", b->device);
    const hipEvent_t origin = b->timeline.front().start;
    const uint32_t call0 = b->timeline.front().call;
    for (size_t i = 0; i < b->timeline.size() && i < cap; ++i) {
        float t0 = 0.0f, t1 = 0.0f;
        hipError_t e = hipEventElapsedTime(&t0, origin, b->timeline[i].start);
        if (e == hipSuccess) e = hipEventElapsedTime(&t1, origin, b->timeline[i].end);
        if (e != hipSuccess)
            return rd_fail(RD_ERR_HIP, "launch %zu of the timeline: %s (synchronise the stream before reading it)", i, hipGetErrorString(e));
        if (start_us) start_us[i] = t0 * 1e3f;
        if (end_us) end_us[i] = t1 * 1e3f;
        if (call_index) call_index[i] = b->timeline[i].call - call0;
    }
    return RD_OK;
}
RD_CATCH_INT(rd_batch_launch_timeline)

extern "C" int rd_batch_histogram(rd_batch *b, uint64_t *hist_dev, void *stream) try
{
    RD_ENTRY(rd_batch_histogram);
    if (!b || !hist_dev) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (!b->hist) return rd_fail(RD_ERR_INVALID_ARG, "batch was created without a histogram");
    rd_devguard g(b->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", b->device);
    hipLaunchKernelGGL(rd_reduce_slab64, dim3(24), dim3(RD_FOLD_THREADS), 0, (hipStream_t)stream, b->slab64,
                       b->blocks * b->n_streams, (unsigned long long *)hist_dev);
    RD_HIP(hipGetLastError());
    return RD_OK;
}
RD_CATCH_INT(rd_batch_histogram)

// ------------------------------------------------------------------------------------------------
// rd_node_batch: the batch path over the GPUs of one node from ONE process (SURVEY.md section 8b "Batch", 8e)
//
// Frames share nothing (the demosaic clamps at the frame edge, shaders.rs:163-166), so frame i simply belongs to device
// i mod N; its CFA plane and surface live in that device's HBM and no pixel crosses xGMI.  One rd_batch, one stream and
// one 768 x u64 histogram per device; enqueueing is done by one host thread per device.  The only exchange is the global
// histogram: ncclAllReduce(768, ncclUint64, ncclSum) over RCCL (librccl.so is loaded on first use and only when N > 1;
// u64 because 2048 x 24 MP overflows u32).  With N = 1 there is no communicator.
// ------------------------------------------------------------------------------------------------
namespace {
struct rd_rccl_api {
    void *handle = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
    bool ok = false;
    bool standin = false;                        // RD_NODE_REDUCE=standin: the tests' stand-in, not RCCL (ranks may share a device)
};
constexpr int RD_NCCL_UINT64 = 5, RD_NCCL_SUM = 0;           // rccl.h: ncclUint64, ncclSum

rd_rccl_api &rd_rccl()
{
    static rd_rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = getenv("RAWDEV_RCCL_LIB");
        const char *mode = getenv("RD_NODE_REDUCE");
        const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        // One RCCL per process: a copy that is already mapped (a PyTorch process has its own) serves us too and wins over
        // RAWDEV_RCCL_LIB, which only names the file to load when none is.  The exception is explicit:
        // RD_NODE_REDUCE=standin (tests) loads exactly the file RAWDEV_RCCL_LIB names -- the host-memory stand-in of
        // tests/cpp/rccl_standin.cpp -- and only then may ranks share a device.
        if (mode && !strcmp(mode, "standin")) {
            if (!env || !*env) { api.error = "RD_NODE_REDUCE=standin needs RAWDEV_RCCL_LIB=<the stand-in library>"; return; }
            api.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
            if (!api.handle) { api.error = std::string("cannot load RAWDEV_RCCL_LIB=") + env + ": " + (dlerror() ? dlerror() : "?"); return; }
            api.standin = true;
        }
        for (const char *n : { "librccl.so.1", "librccl.so" })
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (!api.handle && env && *env) {
            api.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
            if (!api.handle) { api.error = std::string("cannot load RAWDEV_RCCL_LIB=") + env + ": " + (dlerror() ? dlerror() : "?"); return; }
        }
        for (const char *n : names)
            if (!api.handle && n && *n) api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) { api.error = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "not found"); return; }
        auto sym = [&](const char *n) { void *p = dlsym(api.handle, n); if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + n; return p; };
        api.CommInitAll = (int (*)(void **, int, const int *))sym("ncclCommInitAll");
        api.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
        api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))sym("ncclAllReduce");
        api.GroupStart = (int (*)())sym("ncclGroupStart");
        api.GroupEnd = (int (*)())sym("ncclGroupEnd");
        api.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
        api.ok = api.error.empty();
    });
    return api;
}
}  // namespace

enum { RD_NODE_REDUCE_NONE = 0, RD_NODE_REDUCE_RCCL = 1, RD_NODE_REDUCE_HOST = 2 };

#include "rd_node_worker.h"                   // one host thread per device (plain C++: tests/cpp/test_node_worker.cpp runs it under TSan)

struct rd_node_batch {
    uint32_t n = 0, w = 0, h = 0, fmt = 0;
    bool hist = false;
    int reduce = RD_NODE_REDUCE_NONE;
    std::vector<int> devices;
    std::vector<rd_batch *> batches;
    std::vector<hipStream_t> streams;
    std::vector<uint64_t *> hist_dev;          // 768 x u64 per device: the interval of the last enqueue (after an all-reduce: its global sum)
    std::vector<uint64_t *> hist_acc;          // 768 x u64 per device: the intervals enqueued since the last fetch, summed
    uint64_t *hist_pin = nullptr;              // page-locked, n x 768: where rd_node_batch_histogram_enqueue's read-backs land
    std::vector<hipEvent_t> hist_ready;        // per device: its read-back has landed
    uint32_t hist_pending = 0;                 // devices whose read-back the next fetch waits for (0: nothing enqueued)
    std::vector<void *> comms;                 // ncclComm_t per device (RCCL only)
    std::vector<std::vector<rd_frame>> share;  // the frames of the current call, per device
    std::unique_ptr<rd_node_worker[]> workers; // n of them when n > 1 (a single device is served by the calling thread)
    std::mutex call_mu;                        // one rd_node_batch_develop at a time (the workers hold one job each)
};

extern "C" uint32_t rd_node_batch_device_of(uint32_t n_devices, size_t frame_index) try
{
    RD_ENTRY(rd_node_batch_device_of);
    return n_devices ? (uint32_t)(frame_index % n_devices) : 0u;      // frame i -> GPU i mod N
}
RD_CATCH_VAL(rd_node_batch_device_of, 0)

extern "C" void rd_node_batch_destroy(rd_node_batch *nb) try
{
    if (!nb) return;
    if (nb->workers)
        for (uint32_t d = 0; d < nb->n; ++d) nb->workers[d].stop();      // join what was started (all of them, or the first k of a failed create)
    for (uint32_t d = 0; d < nb->n; ++d) {
        if (d >= nb->streams.size() || !nb->streams[d]) continue;      // nothing was set up on this entry (failed create)
        rd_devguard g(nb->devices[d]);
        (void)hipStreamSynchronize(nb->streams[d]);
    }
    if (nb->reduce == RD_NODE_REDUCE_RCCL && rd_rccl().ok)
        for (void *c : nb->comms) if (c) (void)rd_rccl().CommDestroy(c);
    for (uint32_t d = 0; d < nb->n; ++d) {
        if (d < nb->batches.size()) rd_batch_destroy(nb->batches[d]);
        const bool any = (d < nb->hist_dev.size() && nb->hist_dev[d]) || (d < nb->streams.size() && nb->streams[d]);
        if (!any) continue;
        rd_devguard g(nb->devices[d]);
        if (nb->hist_dev[d]) (void)hipFree(nb->hist_dev[d]);
        if (d < nb->hist_acc.size() && nb->hist_acc[d]) (void)hipFree(nb->hist_acc[d]);
        if (d < nb->hist_ready.size() && nb->hist_ready[d]) (void)hipEventDestroy(nb->hist_ready[d]);
        if (nb->streams[d]) (void)hipStreamDestroy(nb->streams[d]);
    }
    if (nb->hist_pin) { rd_devguard g(nb->devices[0]); (void)hipHostFree(nb->hist_pin); }
    delete nb;
}
RD_CATCH_VOID(rd_node_batch_destroy)

extern "C" int rd_node_batch_create(const int *devices, uint32_t n_devices, uint32_t width, uint32_t height, uint32_t format,
                                    uint32_t with_histogram, rd_node_batch **out) try
{
    RD_ENTRY(rd_node_batch_create);
    if (!out) return rd_fail(RD_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!devices || !n_devices || n_devices > 64) return rd_fail(RD_ERR_INVALID_ARG, "need 1..64 devices");
    bool dup = false;
    for (uint32_t a = 0; a < n_devices; ++a)
        for (uint32_t b = a + 1; b < n_devices; ++b) dup = dup || devices[a] == devices[b];
    const char *env = getenv("RD_NODE_REDUCE");              // "host": fold on the host; "rccl": a communicator even for N = 1;
    const bool want_host = env && !strcmp(env, "host");      // "standin": the RCCL branch over the tests' stand-in library
    const bool want_standin = env && !strcmp(env, "standin");
    const bool want_rccl = want_standin || (env && !strcmp(env, "rccl"));
    // A device listed twice is a rehearsal of N > 1 on a one-GPU box: allowed only on explicit request -- the host fold, or
    // the stand-in for librccl (real RCCL wants one rank per device).
    if (dup && !want_host && !want_standin)
        return rd_fail(RD_ERR_INVALID_ARG, "device list holds a device twice (RCCL wants one rank per device; RD_NODE_REDUCE=host "
                                           "allows it for rehearsals on a one-GPU box)");
    rd_node_batch *nb = new (std::nothrow) rd_node_batch;
    if (!nb) return rd_fail(RD_ERR_OOM, "host allocation failed");
    struct undo { void operator()(rd_node_batch *q) const { rd_node_batch_destroy(q); } };
    std::unique_ptr<rd_node_batch, undo> hold(nb);               // whatever fails or throws below: what exists is torn down, threads joined
    nb->n = n_devices; nb->w = width; nb->h = height; nb->fmt = format; nb->hist = with_histogram != 0;
    nb->devices.assign(devices, devices + n_devices);
    nb->batches.assign(n_devices, nullptr);
    nb->streams.assign(n_devices, nullptr);
    nb->hist_dev.assign(n_devices, nullptr);
    nb->hist_acc.assign(n_devices, nullptr);
    nb->hist_ready.assign(n_devices, nullptr);
    nb->comms.assign(n_devices, nullptr);
    nb->share.resize(n_devices);
    int rc = RD_OK;
    for (uint32_t d = 0; d < n_devices && rc == RD_OK; ++d) {
        rc = rd_batch_create(devices[d], width, height, format, with_histogram, &nb->batches[d]);
        if (rc) break;
        rd_devguard g(devices[d]);
        hipError_t e = hipStreamCreateWithFlags(&nb->streams[d], hipStreamNonBlocking);
        if (e == hipSuccess && nb->hist) e = hipMalloc((void **)&nb->hist_dev[d], 768 * sizeof(uint64_t));
        if (e == hipSuccess && nb->hist) e = hipMalloc((void **)&nb->hist_acc[d], 768 * sizeof(uint64_t));
        if (e == hipSuccess && nb->hist) e = hipMemset(nb->hist_acc[d], 0, 768 * sizeof(uint64_t));
        if (e == hipSuccess && nb->hist) e = hipEventCreateWithFlags(&nb->hist_ready[d], hipEventDisableTiming);
        // page-locked, visible to every device of the node, host-coherent: the read-back kernels store into it and the
        // host reads it after their event -- the three properties are REQUESTED, not left to the HIP_HOST_COHERENT default
        if (e == hipSuccess && nb->hist && d == 0)
            e = hipHostMalloc((void **)&nb->hist_pin, (size_t)n_devices * 768 * sizeof(uint64_t),
                              hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) rc = rd_fail(e == hipErrorOutOfMemory ? RD_ERR_OOM : RD_ERR_HIP, "device %d: %s", devices[d], hipGetErrorString(e));
    }
    if (rc == RD_OK && nb->hist) {
        if (want_host) nb->reduce = n_devices > 1 ? RD_NODE_REDUCE_HOST : RD_NODE_REDUCE_NONE;
        else if (n_devices > 1 || want_rccl) {
            rd_rccl_api &api = rd_rccl();
            if (!api.ok) rc = rd_fail(RD_ERR_UNSUPPORTED, "global histogram over %u devices needs RCCL: %s", n_devices, api.error.c_str());
            else {
                const int r = api.CommInitAll(nb->comms.data(), (int)n_devices, nb->devices.data());
                if (r != 0) rc = rd_fail(RD_ERR_HIP, "ncclCommInitAll: %s", api.GetErrorString(r));
                else nb->reduce = RD_NODE_REDUCE_RCCL;
            }
        }
    }
    if (rc == RD_OK && n_devices > 1) {                          // the workers: a thread that cannot be started is a status, not a terminate
        nb->workers.reset(new rd_node_worker[n_devices]);
        for (uint32_t d = 0; d < n_devices; ++d) {
            rd_node_worker &w = nb->workers[d];
            w.index = d; w.device = devices[d];
            w.on_start = [](int dev) { (void)hipSetDevice(dev); };
            w.caught = rd_caught;
            w.last_error = rd_last_error;
            try {
                RD_FAULT_POINT("node.thread");
                w.th = std::thread([&w] { w.loop(); });
            } catch (const std::exception &e) {
                rc = rd_fail(RD_ERR_INTERNAL, "cannot start the host thread of device %d (%u of %u): %s", devices[d], d + 1u, n_devices, e.what());
                break;
            }
        }
    }
    if (rc) {                                                    // the teardown (hold) may overwrite the message: keep it
        char keep[sizeof g_err];
        snprintf(keep, sizeof keep, "%s", g_err);
        hold.reset();
        snprintf(g_err, sizeof g_err, "%s", keep);
        return rc;
    }
    *out = hold.release();
    return RD_OK;
}
RD_CATCH_INT(rd_node_batch_create)

extern "C" int rd_node_batch_set_math_mode(rd_node_batch *nb, uint32_t mode) try
{
    RD_ENTRY(rd_node_batch_set_math_mode);
    if (!nb) return rd_fail(RD_ERR_INVALID_ARG, "NULL node batch");
    for (rd_batch *b : nb->batches) { int rc = rd_batch_set_math_mode(b, mode); if (rc) return rc; }
    return RD_OK;
}
RD_CATCH_INT(rd_node_batch_set_math_mode)

// run fn(d) for every device -- on the device's worker thread when there is more than one; first error wins.  Nothing here
// allocates or starts a thread: the job is a pointer to the caller's functor, which outlives the wait below.
template <typename F> static int rd_node_for_each(rd_node_batch *nb, F fn)
{
    if (nb->n == 1 || !nb->workers) return fn(0u);
    int (*tramp)(void *, uint32_t) = [](void *c, uint32_t d) -> int { return (*static_cast<F *>(c))(d); };
    for (uint32_t d = 0; d < nb->n; ++d) nb->workers[d].post(tramp, &fn);
    int first = RD_OK;
    uint32_t who = 0;
    for (uint32_t d = 0; d < nb->n; ++d) {                      // wait for ALL of them, whatever the first one says
        const int rc = nb->workers[d].wait();
        if (rc && !first) { first = rc; who = d; }
    }
    if (first) return rd_fail(first, "device %d: %s", nb->devices[who], nb->workers[who].msg);
    return RD_OK;
}

extern "C" int rd_node_batch_develop(rd_node_batch *nb, const rd_frame *frames, size_t n_frames, uint32_t row_bands) try
{
    RD_ENTRY(rd_node_batch_develop);
    if (!nb || (!frames && n_frames)) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> call(nb->call_mu);
    for (auto &v : nb->share) v.clear();
    RD_FAULT_POINT("node.share");
    for (size_t i = 0; i < n_frames; ++i) nb->share[i % nb->n].push_back(frames[i]);      // frame i -> GPU i mod N
    return rd_node_for_each(nb, [&](uint32_t d) -> int {
        const std::vector<rd_frame> &v = nb->share[d];
        return v.empty() ? (int)RD_OK : rd_batch_develop(nb->batches[d], v.data(), v.size(), row_bands, nb->streams[d]);
    });
}
RD_CATCH_INT(rd_node_batch_develop)

extern "C" void *rd_node_batch_stream(rd_node_batch *nb, uint32_t index) try
{
    RD_ENTRY(rd_node_batch_stream);
    return nb && index < nb->n ? (void *)nb->streams[index] : nullptr;
}
RD_CATCH_VAL(rd_node_batch_stream, nullptr)

extern "C" uint32_t rd_node_batch_last_launch_count(const rd_node_batch *nb, uint32_t index) try
{
    RD_ENTRY(rd_node_batch_last_launch_count);
    return nb && index < nb->n ? rd_batch_last_launch_count(nb->batches[index]) : 0u;
}
RD_CATCH_VAL(rd_node_batch_last_launch_count, 0)

extern "C" int rd_node_batch_reduce_kind(const rd_node_batch *nb) try { RD_ENTRY(rd_node_batch_reduce_kind); return nb ? nb->reduce : -1; } RD_CATCH_INT(rd_node_batch_reduce_kind)

// test hook: what devices[index]'s 768 x u64 buffer holds after the last rd_node_batch_histogram (after an all-reduce
// every device must hold the global sum, not only the one the call reads back)
extern "C" int rd_debug_node_histogram_of(rd_node_batch *nb, uint32_t index, uint64_t hist[768]) try
{
    RD_ENTRY(rd_debug_node_histogram_of);
    if (!nb || !hist || index >= nb->n || !nb->hist) return rd_fail(RD_ERR_INVALID_ARG, "rd_debug_node_histogram_of: bad argument");
    rd_devguard g(nb->devices[index]);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", nb->devices[index]);
    RD_HIP(hipMemcpyAsync(hist, nb->hist_dev[index], 768 * sizeof(uint64_t), hipMemcpyDeviceToHost, nb->streams[index]));
    RD_HIP(hipStreamSynchronize(nb->streams[index]));
    return RD_OK;
}
RD_CATCH_INT(rd_debug_node_histogram_of)

extern "C" int rd_node_batch_synchronize(rd_node_batch *nb) try
{
    RD_ENTRY(rd_node_batch_synchronize);
    if (!nb) return rd_fail(RD_ERR_INVALID_ARG, "NULL node batch");
    for (uint32_t d = 0; d < nb->n; ++d) {
        rd_devguard g(nb->devices[d]);
        if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", nb->devices[d]);
        RD_HIP(hipStreamSynchronize(nb->streams[d]));
    }
    return RD_OK;
}
RD_CATCH_INT(rd_node_batch_synchronize)

// The global histogram in two halves, so that a host which runs step after step need not drain its devices for it (the
// one-process-per-GPU host never does: its fold and all-reduce are stream-ordered):
//   enqueue: per-device fold of the slabs into 768 x u64 on each device's stream (behind the launches enqueued there), the
//            RCCL all-reduce in place (one group), then ONE small kernel per read-back device that adds the interval to the
//            device's running sum (hist_acc: everything enqueued since the last fetch) and stores that sum into the handle's
//            page-locked buffer -- device 0 after an all-reduce (every device then holds the interval's global sum), every
//            device for the host fold.  Nothing is waited for.
//   fetch:   waits for those read-backs only (an event per device, not the streams: develop calls already enqueued behind
//            them keep running), hands out the sum of EVERY interval enqueued since the last fetch, and enqueues the reset
//            of the running sums.  So a host may enqueue after every develop call and fetch once (bench.py --host node):
//            no interval is lost, and "everything developed since the last fetch" holds for any mix of the two forms.
static int rd_node_histogram_enqueue(rd_node_batch *nb)
{
    for (uint32_t d = 0; d < nb->n; ++d) {
        int rc = rd_batch_histogram(nb->batches[d], nb->hist_dev[d], nb->streams[d]);
        if (rc) return rc;
    }
    if (nb->reduce == RD_NODE_REDUCE_RCCL) {                  // one in-place all-reduce of 6 KiB per device, grouped
        rd_rccl_api &api = rd_rccl();
        int r = api.GroupStart();
        for (uint32_t d = 0; d < nb->n && r == 0; ++d) {
            rd_devguard g(nb->devices[d]);
            r = api.AllReduce(nb->hist_dev[d], nb->hist_dev[d], 768, RD_NCCL_UINT64, RD_NCCL_SUM, nb->comms[d], nb->streams[d]);
        }
        const int r2 = api.GroupEnd();
        if (r == 0) r = r2;
        if (r != 0) return rd_fail(RD_ERR_HIP, "ncclAllReduce: %s", api.GetErrorString(r));
    }
    const uint32_t take = nb->reduce == RD_NODE_REDUCE_HOST ? nb->n : 1u;      // after an all-reduce every device holds the sum
    // The read-back is a 768-thread KERNEL that stores into the page-locked buffer, not a hipMemcpyAsync: a copy-engine
    // transfer between two kernels of one stream is ordered through signals on both sides and leaves the compute queue idle
    // meanwhile, once per step.  Its visibility to the host rests on the buffer being host-coherent (requested at create)
    // and on the system-scope release of the kernel's end, which the event waits for.
    // hist_pending stays 0 until EVERY read-back and its event are enqueued: a failure half way leaves "nothing enqueued"
    // for fetch to report, never the sum of the first k devices.
    nb->hist_pending = 0;
    for (uint32_t d = 0; d < take; ++d) {
        rd_devguard g(nb->devices[d]);
        if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", nb->devices[d]);
        uint64_t *dst = nb->hist_pin + (size_t)d * 768u;
        hipLaunchKernelGGL(rd_acc_hist64, dim3(1), dim3(768), 0, nb->streams[d], (const unsigned long long *)nb->hist_dev[d],
                           (unsigned long long *)nb->hist_acc[d], (unsigned long long *)dst);
        RD_HIP(hipGetLastError());
        RD_HIP(hipEventRecord(nb->hist_ready[d], nb->streams[d]));
    }
    nb->hist_pending = take;
    return RD_OK;
}

static int rd_node_histogram_fetch(rd_node_batch *nb, uint64_t hist[768])
{
    if (!nb->hist_pending) return rd_fail(RD_ERR_INVALID_ARG, "rd_node_batch_histogram_fetch: nothing was enqueued");
    memset(hist, 0, 768 * sizeof(uint64_t));
    for (uint32_t d = 0; d < nb->hist_pending; ++d) {
        rd_devguard g(nb->devices[d]);
        if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", nb->devices[d]);
        RD_HIP(hipEventSynchronize(nb->hist_ready[d]));
        const uint64_t *part = nb->hist_pin + (size_t)d * 768u;
        for (int k = 0; k < 768; ++k) hist[k] += part[k];
    }
    const uint32_t took = nb->hist_pending;
    nb->hist_pending = 0;
    for (uint32_t d = 0; d < took; ++d) {                        // the running sums start again (stream-ordered behind the read-backs)
        rd_devguard g(nb->devices[d]);
        if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", nb->devices[d]);
        RD_HIP(hipMemsetAsync(nb->hist_acc[d], 0, 768 * sizeof(uint64_t), nb->streams[d]));
    }
    return RD_OK;
}

extern "C" int rd_node_batch_histogram_enqueue(rd_node_batch *nb) try
{
    RD_ENTRY(rd_node_batch_histogram_enqueue);
    if (!nb) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (!nb->hist) return rd_fail(RD_ERR_INVALID_ARG, "node batch was created without a histogram");
    std::lock_guard<std::mutex> call(nb->call_mu);
    return rd_node_histogram_enqueue(nb);
}
RD_CATCH_INT(rd_node_batch_histogram_enqueue)

extern "C" int rd_node_batch_histogram_fetch(rd_node_batch *nb, uint64_t hist[768]) try
{
    RD_ENTRY(rd_node_batch_histogram_fetch);
    if (!nb || !hist) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (!nb->hist) return rd_fail(RD_ERR_INVALID_ARG, "node batch was created without a histogram");
    std::lock_guard<std::mutex> call(nb->call_mu);
    return rd_node_histogram_fetch(nb, hist);
}
RD_CATCH_INT(rd_node_batch_histogram_fetch)

// Both halves and a synchronise: the call returns with every device's work done.
extern "C" int rd_node_batch_histogram(rd_node_batch *nb, uint64_t hist[768]) try
{
    RD_ENTRY(rd_node_batch_histogram);
    if (!nb || !hist) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (!nb->hist) return rd_fail(RD_ERR_INVALID_ARG, "node batch was created without a histogram");
    {
        std::lock_guard<std::mutex> call(nb->call_mu);
        int rc = rd_node_histogram_enqueue(nb);
        if (rc == RD_OK) rc = rd_node_histogram_fetch(nb, hist);
        if (rc) return rc;
    }
    return rd_node_batch_synchronize(nb);
}
RD_CATCH_INT(rd_node_batch_histogram)

// ------------------------------------------------------------------------------------------------
// rd_exporter: develop -> HBM slot -> pinned host slot, copy stream overlapping the compute stream
// ------------------------------------------------------------------------------------------------
struct rd_export_slot {
    void *dev = nullptr;
    void *host = nullptr;
    hipEvent_t kernel_done = nullptr, copy_done = nullptr;
    bool busy = false, used = false;
    // rd_exporter_submit_host only (allocated by the first such call): the slot's own CFA plane in HBM and, for a pageable
    // source, the page-locked staging it is copied through
    void *cfa_dev = nullptr, *cfa_pin = nullptr;
    hipEvent_t upload_done = nullptr;
};

struct rd_exporter {
    int device = 0;
    uint32_t w = 0, h = 0, fmt = 0, math_mode = RD_MATH_STRICT, n_slots = 0, next = 0;
    size_t bytes = 0;
    rd_launch_cfg cfg;
    hipStream_t compute = nullptr, copy = nullptr, upload = nullptr;
    rd_export_slot *slots = nullptr;
    rd_scratch scratch;
    std::mutex mu;
};

extern "C" void rd_exporter_destroy(rd_exporter *e) try
{
    if (!e) return;
    {
        rd_devguard g(e->device);
        if (e->compute) (void)hipStreamSynchronize(e->compute);
        if (e->copy) (void)hipStreamSynchronize(e->copy);
        if (e->upload) (void)hipStreamSynchronize(e->upload);
        for (uint32_t i = 0; e->slots && i < e->n_slots; ++i) {
            if (e->slots[i].cfa_dev) (void)hipFree(e->slots[i].cfa_dev);
            if (e->slots[i].cfa_pin) (void)hipHostFree(e->slots[i].cfa_pin);
            if (e->slots[i].upload_done) (void)hipEventDestroy(e->slots[i].upload_done);
            if (e->slots[i].dev) (void)hipFree(e->slots[i].dev);
            if (e->slots[i].host) (void)hipHostFree(e->slots[i].host);
            if (e->slots[i].kernel_done) (void)hipEventDestroy(e->slots[i].kernel_done);
            if (e->slots[i].copy_done) (void)hipEventDestroy(e->slots[i].copy_done);
        }
        if (e->compute) (void)hipStreamDestroy(e->compute);
        if (e->copy) (void)hipStreamDestroy(e->copy);
        if (e->upload) (void)hipStreamDestroy(e->upload);
        e->scratch.release();
    }
    delete[] e->slots;
    delete e;
}
RD_CATCH_VOID(rd_exporter_destroy)

extern "C" int rd_exporter_create(int device, uint32_t w, uint32_t h, uint32_t fmt, uint32_t math_mode, uint32_t n_slots,
                                  rd_exporter **out) try
{
    RD_ENTRY(rd_exporter_create);
    if (!out) return rd_fail(RD_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!w || !h || !n_slots || n_slots > 64) return rd_fail(RD_ERR_INVALID_ARG, "bad frame size or slot count");
    const size_t bpp = rd_format_bytes_per_pixel(fmt);
    if (!bpp) return rd_fail(RD_ERR_INVALID_ARG, "unknown format %u", fmt);
    if (fmt == RD_FMT_RGB_U8 && w < 128u) return rd_fail(RD_ERR_UNSUPPORTED, "RGB8 export needs a frame at least 128 pixels wide (got %u)", w);
    if (math_mode != RD_MATH_STRICT && math_mode != RD_MATH_CONTRACTED) return rd_fail(RD_ERR_INVALID_ARG, "unknown math mode %u", math_mode);
    if (!(rd_identity_map(w) && rd_identity_map(h))) return rd_fail(RD_ERR_UNSUPPORTED, "export map is not the identity for %ux%u", w, h);
    int n_cu = 0;
    int rc = rd_check_device(device, &n_cu);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    rc = rd_q8_lut_ensure(device);
    if (rc == RD_OK && fmt == RD_FMT_RGBA_F16) rc = rd_f16_lut_ensure(device);
    if (rc) return rc;
    rd_exporter *e = new (std::nothrow) rd_exporter;
    if (!e) return rd_fail(RD_ERR_OOM, "host allocation failed");
    struct undo { void operator()(rd_exporter *q) const { rd_exporter_destroy(q); } };
    std::unique_ptr<rd_exporter, undo> hold(e);                  // whatever fails or throws below: what exists is released
    RD_FAULT_POINT("exporter.slots");
    e->device = device; e->w = w; e->h = h; e->fmt = fmt; e->math_mode = math_mode; e->n_slots = n_slots;
    e->bytes = (size_t)w * h * bpp;
    e->cfg.n_cu = n_cu;
    e->slots = new (std::nothrow) rd_export_slot[n_slots];
    hipError_t err = e->slots ? hipSuccess : hipErrorOutOfMemory;
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&e->compute, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&e->copy, hipStreamNonBlocking);
    for (uint32_t i = 0; err == hipSuccess && i < n_slots; ++i) {
        err = hipMalloc(&e->slots[i].dev, e->bytes);
        if (err == hipSuccess) err = hipHostMalloc(&e->slots[i].host, e->bytes, hipHostMallocDefault);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->slots[i].kernel_done, hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->slots[i].copy_done, hipEventDisableTiming);
    }
    if (err != hipSuccess)
        return rd_fail(err == hipErrorOutOfMemory ? RD_ERR_OOM : RD_ERR_HIP, "exporter setup failed: %s", hipGetErrorString(err));
    *out = hold.release();
    return RD_OK;
}
RD_CATCH_INT(rd_exporter_create)

// cfa_host != nullptr: the frame's CFA plane comes from HOST memory (rd_exporter_submit_host) and travels through the slot's
// own HBM plane on a third stream, so the upload of frame i+1 runs under the kernel and the read-back of frame i (PCIe is
// full duplex).  A page-locked source is read by the DMA engine where it lies; a pageable one (RawDataResult's Vec<u16>) is
// copied through page-locked staging in 8 MiB pieces -- helper threads copy piece k+1 while piece k is on the bus -- and is
// free to be reused when the call returns.
static int rd_exporter_submit_impl(rd_exporter *e, const rd_frame *fr, const uint16_t *cfa_host, uint32_t *slot_out)
{
    rd_devguard g(e->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    const uint32_t si = e->next;
    rd_export_slot &s = e->slots[si];
    if (s.busy) return rd_fail(RD_ERR_INVALID_ARG, "slot %u has not been released (ring of %u full)", si, e->n_slots);
    if (fr->matrix_layout != RD_MATRIX_REFERENCE && fr->matrix_layout != RD_MATRIX_ROW_MAJOR) return rd_fail(RD_ERR_INVALID_ARG, "unknown matrix layout %u", fr->matrix_layout);
    const uint16_t *cfa_dev = fr->cfa_dev;
    if (cfa_host) {
        const size_t cfa_bytes = (size_t)e->w * e->h * sizeof(uint16_t);
        if (!e->upload) RD_HIP(hipStreamCreateWithFlags(&e->upload, hipStreamNonBlocking));
        if (!s.cfa_dev) RD_HIP(hipMalloc(&s.cfa_dev, cfa_bytes));
        if (!s.upload_done) RD_HIP(hipEventCreateWithFlags(&s.upload_done, hipEventDisableTiming));
        // the slot's previous kernel read this plane: it has finished (its frame was waited for and released), but say so
        if (s.used) RD_HIP(hipStreamWaitEvent(e->upload, s.kernel_done, 0));
        const int kind = rd_host_memory_kind(cfa_host, cfa_bytes);
        if (kind == RD_MEM_DEVICE) return rd_fail(RD_ERR_INVALID_ARG, "cfa_host is a device pointer: use rd_exporter_submit");
        if (kind == RD_MEM_PINNED && !getenv("RD_ASSUME_PAGEABLE")) {
            RD_HIP(hipMemcpyAsync(s.cfa_dev, cfa_host, cfa_bytes, hipMemcpyHostToDevice, e->upload));
        } else {
            if (!s.cfa_pin) RD_HIP(hipHostMalloc(&s.cfa_pin, cfa_bytes, hipHostMallocDefault));
            else RD_HIP(hipEventSynchronize(s.upload_done));     // the staging's previous upload (long finished)
            const size_t piece = (size_t)8 << 20;
            for (size_t off = 0; off < cfa_bytes; off += piece) {
                const size_t n = cfa_bytes - off < piece ? cfa_bytes - off : piece;
                rd_copy_pool::get().copy((char *)s.cfa_pin + off, (const char *)cfa_host + off, n);
                RD_HIP(hipMemcpyAsync((char *)s.cfa_dev + off, (const char *)s.cfa_pin + off, n, hipMemcpyHostToDevice, e->upload));
            }
        }
        RD_HIP(hipEventRecord(s.upload_done, e->upload));
        RD_HIP(hipStreamWaitEvent(e->compute, s.upload_done, 0));
        cfa_dev = (const uint16_t *)s.cfa_dev;
    }
    // the previous copy out of this HBM slot must have finished before the kernel overwrites it
    if (s.used) RD_HIP(hipStreamWaitEvent(e->compute, s.copy_done, 0));
    const rd_ku u = rd_frame_ku(fr->params, fr->wb_multipliers, fr->color_matrix, 1.0f, 0.0f, 0.0f, fr->black_level, e->math_mode, fr->matrix_layout);
    const rd_scratch::lease l = e->scratch.get(e->compute, false);
    if (l.idx < 0) return rd_fail(RD_ERR_OOM, "scheduler state allocation failed");
    int rc = rd_enqueue_render(e->cfg, cfa_dev, e->w, e->h, e->w, e->h, e->fmt, s.dev, u, true, 0, e->h / 2u + 1u, false,
                               e->math_mode, nullptr, nullptr, 0, l.tq, e->compute, nullptr);
    e->scratch.used(l, e->compute, rc != RD_OK);
    if (rc) return rc;
    RD_HIP(hipEventRecord(s.kernel_done, e->compute));
    RD_HIP(hipStreamWaitEvent(e->copy, s.kernel_done, 0));
    RD_HIP(hipMemcpyAsync(s.host, s.dev, e->bytes, hipMemcpyDeviceToHost, e->copy));
    RD_HIP(hipEventRecord(s.copy_done, e->copy));
    s.busy = true; s.used = true;
    e->next = (si + 1u) % e->n_slots;
    *slot_out = si;
    return RD_OK;
}

extern "C" int rd_exporter_submit(rd_exporter *e, const rd_frame *fr, uint32_t *slot_out) try
{
    RD_ENTRY(rd_exporter_submit);
    if (!e || !fr || !slot_out) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (!fr->cfa_dev || ((uintptr_t)fr->cfa_dev % 4u)) return rd_fail(RD_ERR_INVALID_ARG, "cfa_dev NULL or not 4-byte aligned");
    return rd_exporter_submit_impl(e, fr, nullptr, slot_out);
}
RD_CATCH_INT(rd_exporter_submit)

extern "C" int rd_exporter_submit_host(rd_exporter *e, const rd_frame *fr, const uint16_t *cfa_host, uint32_t *slot_out) try
{
    RD_ENTRY(rd_exporter_submit_host);
    if (!e || !fr || !cfa_host || !slot_out) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    return rd_exporter_submit_impl(e, fr, cfa_host, slot_out);
}
RD_CATCH_INT(rd_exporter_submit_host)

extern "C" int rd_exporter_wait(rd_exporter *e, uint32_t slot, const void **data, size_t *len) try
{
    RD_ENTRY(rd_exporter_wait);
    if (!e || !data) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (slot >= e->n_slots) return rd_fail(RD_ERR_INVALID_ARG, "slot %u out of range", slot);
    rd_devguard g(e->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", e->device);
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!e->slots[slot].busy) return rd_fail(RD_ERR_INVALID_ARG, "slot %u holds no frame", slot);
        ev = e->slots[slot].copy_done;
    }
    RD_HIP(hipEventSynchronize(ev));
    *data = e->slots[slot].host;
    if (len) *len = e->bytes;
    return RD_OK;
}
RD_CATCH_INT(rd_exporter_wait)

extern "C" int rd_exporter_release(rd_exporter *e, uint32_t slot) try
{
    RD_ENTRY(rd_exporter_release);
    if (!e) return rd_fail(RD_ERR_INVALID_ARG, "NULL exporter");
    if (slot >= e->n_slots) return rd_fail(RD_ERR_INVALID_ARG, "slot %u out of range", slot);
    std::lock_guard<std::mutex> lk(e->mu);
    e->slots[slot].busy = false;
    return RD_OK;
}
RD_CATCH_INT(rd_exporter_release)

