// rd_host_pipeline.inl -- part of librawdev.so's host side: included by rawdev.hip (one translation unit; the kernels are
// templates in rd_kernels.h).  rd_pipeline = gpu::RenderPipeline (reference src/gpu/pipeline.rs:112-737): render lanes, the band-pipelined
// full-resolution read-back, the host-side copy pool.

#include "rd_copy_pool.h"                     // host-side copy pool (plain C++: tests/cpp/test_copy_pool.cpp runs it under TSan)

// ------------------------------------------------------------------------------------------------
// rd_pipeline
// ------------------------------------------------------------------------------------------------
// A render LANE: everything one host-side render call needs besides the CFA plane -- a compute stream, a copy stream,
// a device surface, a 768-bin histogram, pinned staging for pageable destinations, events.  The reference shares
// Arc<RenderPipeline> between the UI thread (render_to_bytes + render_to_histogram_bytes per redraw, main.rs:1515-1531)
// and the export thread (render_full_res_to_bytes, main.rs:1749-1754); each call takes a free lane for its duration, so
// the 96.6 MB read-back of an export does not stand between a slider move and its preview.  The pipeline's mutex only
// guards the uniforms: a render snapshots them (rd_shot) and lets go.
#define RD_LANES_MAX 4
#define RD_BANDS_MAX 8                           // row-band launches of a full-resolution host render
#define RD_STAGE_SLOTS 3                         // pinned staging slots of RD_STAGE_BYTES each (pageable destinations)
#define RD_STAGE_BYTES ((size_t)8 << 20)
#define RD_BAND_MIN_BYTES ((size_t)16 << 20)     // smaller surfaces: one launch, one copy
#define RD_LENT_MAX 4                            // page-locked surfaces out on loan at a time (rd_render_full_res_borrow)

struct rd_lane {
    hipStream_t compute = nullptr, copy = nullptr;
    void *out_buf = nullptr; size_t out_cap = 0;
    uint32_t *hist_dev = nullptr;
    void *stage[RD_STAGE_SLOTS] = {};
    hipEvent_t kev[RD_BANDS_MAX] = {};           // band k's kernel has finished (compute stream)
    hipEvent_t cev[RD_STAGE_SLOTS] = {};         // the copy into staging slot j has finished (copy stream)
    bool busy = false;
};

struct rd_shot {                                 // what a render needs from the pipeline's mutable state
    rd_ku u;
    bool export_view;                            // zoom 1, pan 0: the export map may apply
    uint32_t math_mode;
};

struct rd_pipeline {
    int device = 0;
    rd_info info{};
    rd_launch_cfg cfg;
    const uint16_t *cfa = nullptr;
    bool owns_cfa = false;
    bool identity_ok = false;
    rd_edit_params params{};
    float wb[4]{}, cm[9]{};
    float zoom = 1.0f, pan_x = 0.0f, pan_y = 0.0f;
    uint32_t black_level = 0;
    uint32_t math_mode = RD_MATH_STRICT;
    uint32_t matrix_layout = RD_MATRIX_REFERENCE;
    hipStream_t stream = nullptr;     // lane 0's compute stream ("the pipeline's own stream" of the test hooks)
    rd_scratch scratch;               // per stream: ticket counters + histogram slab (has its own lock)
    std::mutex mu;                    // the uniforms (Send + Sync like Arc<RenderPipeline>)
    std::mutex lane_mu;               // the lane pool
    std::condition_variable lane_cv;
    std::vector<rd_lane *> lanes;
    // RD_GRAPH=1 (experiment, profiles/r04_single_frame_gap.txt): develop + histogram fold of a whole-frame render as ONE
    // two-node graph per stream, re-parameterised (hipGraphExecKernelNodeSetParams) and launched per call
    struct graph_cache {
        hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr; hipGraphNode_t n_dev = nullptr, n_fold = nullptr;
        const void *fn = nullptr; uint32_t blocks = 0;
    };
    std::mutex graph_mu;
    std::map<hipStream_t, graph_cache> graphs;
    // page-locked surfaces lent to the caller (rd_render_full_res_borrow): allocated once, reused, freed with the pipeline
    struct lent { void *ptr = nullptr; size_t cap = 0; bool busy = false; };
    std::mutex lent_mu;
    std::vector<lent> lents;
};

static void rd_lane_free(rd_lane *l)             // device set, nothing of the lane in flight
{
    if (!l) return;
    if (l->compute) { (void)hipStreamSynchronize(l->compute); (void)hipStreamDestroy(l->compute); }
    if (l->copy) { (void)hipStreamSynchronize(l->copy); (void)hipStreamDestroy(l->copy); }
    if (l->out_buf) (void)hipFree(l->out_buf);
    if (l->hist_dev) (void)hipFree(l->hist_dev);
    for (void *s : l->stage) if (s) (void)hipHostFree(s);
    for (hipEvent_t e : l->kev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : l->cev) if (e) (void)hipEventDestroy(e);
    delete l;
}

static int rd_lane_new(rd_lane **out)            // device set
{
    *out = nullptr;
    rd_lane *l = new (std::nothrow) rd_lane;
    if (!l) return rd_fail(RD_ERR_OOM, "host allocation failed");
    hipError_t e = hipStreamCreateWithFlags(&l->compute, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&l->copy, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&l->hist_dev, 768 * sizeof(uint32_t));
    for (int k = 0; k < RD_BANDS_MAX && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&l->kev[k], hipEventDisableTiming);
    for (int k = 0; k < RD_STAGE_SLOTS && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&l->cev[k], hipEventDisableTiming);
    if (e != hipSuccess) {
        rd_lane_free(l);
        return rd_fail(e == hipErrorOutOfMemory ? RD_ERR_OOM : RD_ERR_HIP, "render lane setup failed: %s", hipGetErrorString(e));
    }
    *out = l;
    return RD_OK;
}

// A free lane for a render that needs `need` bytes of device surface: the smallest free one that is large enough, else
// any free one (it grows), else a new one (up to RD_LANES_MAX), else wait for a call to finish.
static int rd_lane_acquire(rd_pipeline *p, size_t need, rd_lane **out)
{
    *out = nullptr;
    std::unique_lock<std::mutex> lk(p->lane_mu);
    for (;;) {
        rd_lane *fit = nullptr, *any = nullptr;
        for (rd_lane *l : p->lanes) {
            if (l->busy) continue;
            if (!any || l->out_cap > any->out_cap) any = l;
            if (l->out_cap >= need && (!fit || l->out_cap < fit->out_cap)) fit = l;
        }
        rd_lane *l = fit;
        if (!l && any && (p->lanes.size() >= RD_LANES_MAX || need <= RD_BAND_MIN_BYTES || any->out_cap == 0)) l = any;
        if (!l && p->lanes.size() < RD_LANES_MAX) {
            RD_FAULT_POINT("pipeline.lanes");
            const int rc = rd_lane_new(&l);
            if (rc) return rc;
            p->lanes.push_back(l);                               // capacity reserved by rd_pipeline_new: cannot throw
        }
        if (!l && any) l = any;
        if (l) { l->busy = true; *out = l; return RD_OK; }
        p->lane_cv.wait(lk);
    }
}

static void rd_lane_release(rd_pipeline *p, rd_lane *l)
{
    { std::lock_guard<std::mutex> lk(p->lane_mu); l->busy = false; }
    p->lane_cv.notify_one();
}

struct rd_lane_hold {                            // RAII: a lane for the duration of one call
    rd_pipeline *p; rd_lane *l = nullptr; int rc;
    rd_lane_hold(rd_pipeline *pp, size_t need) : p(pp) { rc = rd_lane_acquire(pp, need, &l); }
    ~rd_lane_hold() { if (l) rd_lane_release(p, l); }
};

static int rd_lane_reserve(rd_lane *l, size_t need)          // the lane's device surface holds `need` bytes
{
    if (l->out_cap >= need) return RD_OK;
    if (l->out_buf) { (void)hipFree(l->out_buf); l->out_buf = nullptr; l->out_cap = 0; }
    RD_HIP(hipMalloc(&l->out_buf, need));
    l->out_cap = need;
    return RD_OK;
}

static int rd_pipeline_new(int device, int64_t image_id, const uint16_t *cfa, bool cfa_on_device,
                           uint32_t w, uint32_t h, const rd_edit_params *params, const float wb[4],
                           const float cm[9], rd_pipeline **out)
{
    if (!out) return rd_fail(RD_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!cfa || !params || !wb || !cm) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (!w || !h) return rd_fail(RD_ERR_INVALID_ARG, "empty frame %ux%u", w, h);
    if ((uint64_t)w * h >= 0xffffffffull) return rd_fail(RD_ERR_UNSUPPORTED, "frame %ux%u exceeds 2^32 pixels", w, h);
    int n_cu = 0;
    int rc = rd_check_device(device, &n_cu);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);

    rc = rd_q8_lut_ensure(device);
    // (the binary16 tables too, here and not at the first RGBA-f16 render: building them takes ~40 ms of host time once per
    // process and the upload is a synchronous null-stream copy -- neither belongs on rd_render_device, which promises an
    // asynchronous enqueue on the caller's stream and may run under stream capture)
    if (rc == RD_OK) rc = rd_f16_lut_ensure(device);
    if (rc) return rc;
    rd_pipeline *p = new (std::nothrow) rd_pipeline;
    if (!p) return rd_fail(RD_ERR_OOM, "host allocation failed");
    struct undo { void operator()(rd_pipeline *q) const { rd_pipeline_destroy(q); } };
    std::unique_ptr<rd_pipeline, undo> hold(p);                  // whatever throws below: the half-built pipeline is destroyed
    p->device = device;                                          // first: the undo synchronises and frees on THIS device
    p->cfg.n_cu = n_cu;
    RD_FAULT_POINT("pipeline.lanes");
    p->lanes.reserve(RD_LANES_MAX);
    p->lents.reserve(RD_LENT_MAX);
    p->cfg.wg_per_cu_plain = rd_env_u32("RD_WG_PER_CU", 2);
    p->info.width = w; p->info.height = h; p->info.image_id = image_id;
    rd_derived_dims(w, h, &p->info.preview_width, &p->info.preview_height, &p->info.histogram_width,
                    &p->info.histogram_height);
    p->params = *params;
    memcpy(p->wb, wb, sizeof p->wb);
    memcpy(p->cm, cm, sizeof p->cm);
    p->identity_ok = rd_identity_map(w) && rd_identity_map(h);

    rd_lane *l0 = nullptr;
    rc = rd_lane_new(&l0);
    if (rc) return rc;
    p->lanes.push_back(l0);
    p->stream = l0->compute;
    hipError_t e = hipSuccess;
    if (cfa_on_device) {
        p->cfa = cfa;
    } else {
        void *d = nullptr;
        e = hipMalloc(&d, (size_t)w * h * sizeof(uint16_t));
        if (e == hipSuccess) {
            p->cfa = (const uint16_t *)d; p->owns_cfa = true;
            e = hipMemcpy(d, cfa, (size_t)w * h * sizeof(uint16_t), hipMemcpyHostToDevice);
        }
    }
    if (e != hipSuccess)
        return rd_fail(e == hipErrorOutOfMemory ? RD_ERR_OOM : RD_ERR_HIP, "pipeline setup failed: %s", hipGetErrorString(e));
    *out = hold.release();
    return RD_OK;
}

extern "C" int rd_pipeline_create(int device, int64_t image_id, const uint16_t *cfa, uint32_t w, uint32_t h,
                                  const rd_edit_params *params, const float wb[4], const float cm[9],
                                  rd_pipeline **out) try
{
    RD_ENTRY(rd_pipeline_create);
    return rd_pipeline_new(device, image_id, cfa, false, w, h, params, wb, cm, out);
}
RD_CATCH_INT(rd_pipeline_create)

extern "C" int rd_pipeline_create_from_device(int device, int64_t image_id, const uint16_t *cfa_dev, uint32_t w,
                                              uint32_t h, const rd_edit_params *params, const float wb[4],
                                              const float cm[9], rd_pipeline **out) try
{
    RD_ENTRY(rd_pipeline_create_from_device);
    return rd_pipeline_new(device, image_id, cfa_dev, true, w, h, params, wb, cm, out);
}
RD_CATCH_INT(rd_pipeline_create_from_device)

extern "C" void rd_pipeline_destroy(rd_pipeline *p) try
{
    if (!p) return;
    {
        rd_devguard g(p->device);
        for (rd_lane *l : p->lanes) rd_lane_free(l);
        p->lanes.clear();
        for (auto &kv : p->graphs) { if (kv.second.ex) (void)hipGraphExecDestroy(kv.second.ex); if (kv.second.g) (void)hipGraphDestroy(kv.second.g); }
        p->graphs.clear();
        for (auto &b : p->lents) if (b.ptr) (void)hipHostFree(b.ptr);
        p->lents.clear();
        if (p->owns_cfa && p->cfa) (void)hipFree((void *)p->cfa);
        (void)hipDeviceSynchronize();        // renders enqueued on caller streams (rd_render_device) may still draw tickets
        p->scratch.release();
    }
    delete p;
}
RD_CATCH_VOID(rd_pipeline_destroy)

extern "C" int rd_pipeline_info(const rd_pipeline *p, rd_info *out) try
{
    RD_ENTRY(rd_pipeline_info);
    if (!p || !out) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    *out = p->info;
    return RD_OK;
}
RD_CATCH_INT(rd_pipeline_info)

extern "C" int rd_pipeline_set_black_level(rd_pipeline *p, uint32_t bl) try
{
    RD_ENTRY(rd_pipeline_set_black_level);
    if (!p) return rd_fail(RD_ERR_INVALID_ARG, "NULL pipeline");
    std::lock_guard<std::mutex> lk(p->mu);
    p->black_level = bl;
    return RD_OK;
}
RD_CATCH_INT(rd_pipeline_set_black_level)

extern "C" int rd_pipeline_set_matrix_layout(rd_pipeline *p, uint32_t layout) try
{
    RD_ENTRY(rd_pipeline_set_matrix_layout);
    if (!p) return rd_fail(RD_ERR_INVALID_ARG, "NULL pipeline");
    if (layout != RD_MATRIX_REFERENCE && layout != RD_MATRIX_ROW_MAJOR) return rd_fail(RD_ERR_INVALID_ARG, "unknown matrix layout %u", layout);
    std::lock_guard<std::mutex> lk(p->mu);
    p->matrix_layout = layout;
    return RD_OK;
}
RD_CATCH_INT(rd_pipeline_set_matrix_layout)

extern "C" int rd_pipeline_set_math_mode(rd_pipeline *p, uint32_t mode) try
{
    RD_ENTRY(rd_pipeline_set_math_mode);
    if (!p) return rd_fail(RD_ERR_INVALID_ARG, "NULL pipeline");
    if (mode != RD_MATH_STRICT && mode != RD_MATH_CONTRACTED) return rd_fail(RD_ERR_INVALID_ARG, "unknown math mode %u", mode);
    std::lock_guard<std::mutex> lk(p->mu);
    p->math_mode = mode;
    return RD_OK;
}
RD_CATCH_INT(rd_pipeline_set_math_mode)

extern "C" int rd_update_uniforms_with_zoom(rd_pipeline *p, const rd_edit_params *params, float zoom, float pan_x,
                                            float pan_y) try
{
    RD_ENTRY(rd_update_uniforms_with_zoom);
    if (!p || !params) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(p->mu);
    p->params = *params;     // wb / matrix are preserved, as in pipeline.rs:375-381
    p->zoom = zoom; p->pan_x = pan_x; p->pan_y = pan_y;
    return RD_OK;
}
RD_CATCH_INT(rd_update_uniforms_with_zoom)

extern "C" int rd_update_uniforms(rd_pipeline *p, const rd_edit_params *params) try
{
    RD_ENTRY(rd_update_uniforms);
    return rd_update_uniforms_with_zoom(p, params, 1.0f, 0.0f, 0.0f);   // pipeline.rs:367-369
}
RD_CATCH_INT(rd_update_uniforms)

// The uniforms as they stand now: the only thing a render reads under the pipeline's mutex.  (The reference's export
// re-uses whatever view() last wrote, main.rs:1515 vs :1754; a snapshot keeps that and removes the tear a concurrent
// queue.write_buffer can cause there.)
static rd_shot rd_pipeline_snapshot(rd_pipeline *p)
{
    std::lock_guard<std::mutex> lk(p->mu);
    rd_shot s;
    s.u = rd_frame_ku(p->params, p->wb, p->cm, p->zoom, p->pan_x, p->pan_y, p->black_level, p->math_mode, p->matrix_layout);
    s.export_view = p->zoom == 1.0f && p->pan_x == 0.0f && p->pan_y == 0.0f;
    s.math_mode = p->math_mode;
    return s;
}

// Does a tw x th render of this snapshot take the export kernel (one 2x2 block per lane, identity map)?
static bool rd_pipeline_uses_quads(const rd_pipeline *p, const rd_shot &sh, uint32_t tw, uint32_t th, uint32_t fmt)
{
    const uint32_t W = p->info.width, H = p->info.height;
    // (any width, odd ones too since round 6: the export kernel takes the whole quads and rd_develop_lastcol the last column)
    return tw == W && th == H && sh.export_view && p->identity_ok && ((uintptr_t)p->cfa % 4u) == 0 &&
           (fmt != RD_FMT_RGB_U8 || W >= 128u) && !getenv("RD_FORCE_MAP");     // (RGB8 narrower than one tile: the map kernel's byte stores)
}

// Enqueue one render of the snapshot on stream s: units [unit0, unit1) of the export kernel (the whole frame is
// [0, H/2 + 1)), or the map kernel for any other target.  The device is set.  With hist_dev the launch must be the whole frame.
static int rd_pipeline_enqueue(rd_pipeline *p, const rd_shot &sh, uint32_t tw, uint32_t th, uint32_t fmt, void *dst_dev,
                               uint32_t *hist_dev, hipStream_t s, uint32_t unit0 = 0, uint32_t unit1 = 0)
{
    if (!tw || !th) return rd_fail(RD_ERR_INVALID_ARG, "empty target %ux%u", tw, th);
    if (!rd_format_bytes_per_pixel(fmt)) return rd_fail(RD_ERR_INVALID_ARG, "unknown format %u", fmt);
    if ((uintptr_t)dst_dev % rd_align_for(fmt)) return rd_fail(RD_ERR_INVALID_ARG, "dst is not %zu-byte aligned", rd_align_for(fmt));
    const uint32_t W = p->info.width, H = p->info.height;
    const bool quads = rd_pipeline_uses_quads(p, sh, tw, th, fmt);
    if (!unit1) unit1 = H / 2u + 1u;
    uint32_t blocks = 0;
    const rd_scratch::lease l = p->scratch.get(s, hist_dev != nullptr);     // this stream's counters (+ slab)
    if (l.idx < 0) return rd_fail(RD_ERR_OOM, "scheduler state allocation failed");
    static const bool use_graph = rd_env_u32("RD_GRAPH", 0) != 0;
    if (use_graph && quads && hist_dev && unit0 == 0u && unit1 == H / 2u + 1u && (W % 2u) == 0) {
        rd_quads_call call;
        int rc = rd_enqueue_render(p->cfg, p->cfa, W, H, tw, th, fmt, dst_dev, sh.u, true, unit0, unit1, true, sh.math_mode,
                                   l.slab32, nullptr, 0, l.tq, s, &blocks, &call);
        if (rc == RD_OK) {
            uint32_t *slab = l.slab32, nb = blocks, *hd = hist_dev;
            void *fold_args[3] = { &slab, &nb, &hd };
            hipKernelNodeParams kd{}, kf{};
            kd.func = const_cast<void *>(call.fn); kd.gridDim = dim3(call.blocks); kd.blockDim = dim3(RD_BLOCK); kd.kernelParams = call.argv;
            kf.func = (void *)rd_reduce_slab32; kf.gridDim = dim3(24); kf.blockDim = dim3(RD_FOLD_THREADS); kf.kernelParams = fold_args;
            std::lock_guard<std::mutex> gl(p->graph_mu);
            if (p->graphs.size() > 16 && !p->graphs.count(s)) {              // bounded like the scheduler state
                for (auto &kv : p->graphs) { if (kv.second.ex) (void)hipGraphExecDestroy(kv.second.ex); if (kv.second.g) (void)hipGraphDestroy(kv.second.g); }
                p->graphs.clear();
            }
            rd_pipeline::graph_cache &gc = p->graphs[s];
            hipError_t e = hipSuccess;
            if (!gc.ex || gc.fn != call.fn || gc.blocks != call.blocks) {
                if (gc.ex) (void)hipGraphExecDestroy(gc.ex);
                if (gc.g) (void)hipGraphDestroy(gc.g);
                gc = rd_pipeline::graph_cache{};
                e = hipGraphCreate(&gc.g, 0);
                if (e == hipSuccess) e = hipGraphAddKernelNode(&gc.n_dev, gc.g, nullptr, 0, &kd);
                if (e == hipSuccess) e = hipGraphAddKernelNode(&gc.n_fold, gc.g, &gc.n_dev, 1, &kf);
                if (e == hipSuccess) e = hipGraphInstantiate(&gc.ex, gc.g, nullptr, nullptr, 0);
                gc.fn = call.fn; gc.blocks = call.blocks;
            } else {
                e = hipGraphExecKernelNodeSetParams(gc.ex, gc.n_dev, &kd);
                if (e == hipSuccess) e = hipGraphExecKernelNodeSetParams(gc.ex, gc.n_fold, &kf);
            }
            if (e == hipSuccess) e = hipGraphLaunch(gc.ex, s);
            if (e != hipSuccess) rc = rd_fail(RD_ERR_HIP, "graph launch failed: %s", hipGetErrorString(e));
        }
        p->scratch.used(l, s, rc != RD_OK);
        return rc;
    }
    int rc = rd_enqueue_render(p->cfg, p->cfa, W, H, tw, th, fmt, dst_dev, sh.u, quads, unit0, unit1,
                               hist_dev != nullptr, sh.math_mode, l.slab32, nullptr, 0, l.tq, s, &blocks);
    if (rc == RD_OK && hist_dev) {
        hipLaunchKernelGGL(rd_reduce_slab32, dim3(24), dim3(RD_FOLD_THREADS), 0, s, l.slab32, blocks, hist_dev);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = rd_fail(RD_ERR_HIP, "histogram fold launch failed: %s", hipGetErrorString(e));
    }
    p->scratch.used(l, s, rc != RD_OK);
    return rc;
}

extern "C" int rd_render_device(rd_pipeline *p, uint32_t out_w, uint32_t out_h, uint32_t fmt, void *dst_dev,
                                uint32_t *hist_dev, void *stream) try
{
    RD_ENTRY(rd_render_device);
    if (!p || !dst_dev) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    rd_devguard g(p->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", p->device);
    const rd_shot sh = rd_pipeline_snapshot(p);
    return rd_pipeline_enqueue(p, sh, out_w, out_h, fmt, dst_dev, hist_dev, (hipStream_t)stream);
}
RD_CATCH_INT(rd_render_device)

// Both ENDS of a host range being page-locked is not enough to point a DMA engine at it: two registrations with a gap between
// them, or a buffer that outgrew its registered window, have page-locked ends and a pageable middle.  ONE allocation or
// registration must cover [ptr, ptr + n).  hipPointerGetAttribute's RANGE_START_ADDR / RANGE_SIZE give the base and size of
// the runtime's memory object behind a pointer -- a hipHostMalloc / rd_host_alloc block or a hipHostRegister'ed window (what a
// Rust host does to a Vec it wants to keep) alike (tools/probe_registered_host.py; hipMemGetAddressRange reports a NULL base
// for a registered window, and ROCr's hsa_amd_pointer_info does not know it at all).  If the runtime cannot say, the range is
// staged: always correct, one host copy slower.
static bool rd_one_allocation_covers(const void *ptr, size_t n)
{
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)ptr) != hipSuccess ||
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)ptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const uintptr_t lo = (uintptr_t)base, p0 = (uintptr_t)ptr;
    return lo && p0 >= lo && p0 + n <= lo + size;
}

// Is [ptr, ptr + n) page-locked host memory the DMA engines can write (hipHostMalloc / rd_host_alloc / hipHostRegister)?
enum { RD_MEM_PAGEABLE = 0, RD_MEM_PINNED = 1, RD_MEM_DEVICE = 2 };
static int rd_host_memory_kind(const void *ptr, size_t n)
{
    const char *ends[2] = { (const char *)ptr, (const char *)ptr + (n ? n - 1 : 0) };
    int kind = RD_MEM_PINNED;
    for (const char *q : ends) {
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof a);
        const hipError_t e = hipPointerGetAttributes(&a, q);
        if (e != hipSuccess) { (void)hipGetLastError(); kind = RD_MEM_PAGEABLE; continue; }    // plain malloc memory: "invalid value"
        if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeArray) return RD_MEM_DEVICE;
        if (a.type != hipMemoryTypeHost) kind = RD_MEM_PAGEABLE;       // unregistered / managed: staged
    }
    if (kind == RD_MEM_PINNED && n > 1 && !rd_one_allocation_covers(ptr, n)) return RD_MEM_PAGEABLE;
    return kind;
}

static bool rd_is_pinned_host(const void *ptr, size_t n)
{
    if (getenv("RD_ASSUME_PAGEABLE")) return false;          // A/B switch: stage every destination
    return rd_host_memory_kind(ptr, n) == RD_MEM_PINNED;
}

// A pageable destination that has never been touched (the fresh Vec<u8> the reference's signature returns) costs one page
// fault per 4 KiB inside the staging memcpy: 23 600 faults for a 24 MP RGBA8 surface, several times the PCIe transfer.
// Two hints were tried on the GPU box (THP mode "madvise", kernel 6.18; profiles/r04_fullres_ab.txt) and neither pays:
// MADV_HUGEPAGE on the 2 MiB-aligned interior (46 faults instead of 23 600, but each compacts and zeroes 2 MiB: 8.0 ms
// against 6.5 ms without it) and MADV_POPULATE_WRITE (7.9 ms).  So the default is to do nothing -- the faults are the
// caller's, a reused or page-locked destination avoids them -- and RD_DST_ADVISE=huge | populate keeps the experiment.
static void rd_advise_destination(char *dst, size_t n)
{
    static const int mode = [] { const char *e = getenv("RD_DST_ADVISE"); return !e || !*e ? 0 : !strcmp(e, "huge") ? 1 : !strcmp(e, "populate") ? 2 : 0; }();
    if (!mode || n < ((size_t)8 << 20)) return;
    const uintptr_t huge = (uintptr_t)2 << 20;
    const uintptr_t lo = ((uintptr_t)dst + huge - 1) & ~(huge - 1), hi = ((uintptr_t)dst + n) & ~(huge - 1);
    if (hi <= lo) return;
#ifdef MADV_HUGEPAGE
    if (mode == 1) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
#endif
#ifdef MADV_POPULATE_WRITE
    if (mode == 2) (void)madvise((void *)lo, hi - lo, MADV_POPULATE_WRITE);
#endif
}

// Full-resolution host render (render_full_res_to_bytes, pipeline.rs:526-606, and rd_render of the whole frame): the
// reference renders, copies the texture into a MAP_READ buffer, blocks in poll(Wait) and de-pads 96.6 MB row by row
// ("1-2 seconds for 24MP", pipeline.rs:525).  Here, on the lane's two streams:
//   * the frame is launched as up to RD_BANDS_MAX row bands (the export kernel takes a unit range; a band's rows are
//     contiguous bytes of the surface), so the first bytes cross PCIe while the later bands are still being computed;
//   * a page-locked destination (rd_host_alloc / hipHostMalloc / hipHostRegister; detected) is written by the DMA
//     engine directly, in chunks, all enqueued at once: one synchronise, no host copy;
//   * a pageable destination goes through RD_STAGE_SLOTS pinned slots of 8 MiB: the DMA of chunk c+3 runs while the
//     copy pool moves chunk c into the caller's buffer.
// With a fused histogram the slab is written by ONE launch (no bands); the copies are chunked all the same.
static int rd_render_full_host(rd_pipeline *p, rd_lane *l, const rd_shot &sh, uint32_t fmt, char *dst, size_t need, uint32_t *hist)
{
    const uint32_t W = p->info.width, H = p->info.height;
    const size_t row_bytes = (size_t)W * rd_format_bytes_per_pixel(fmt);
    const uint32_t units = H / 2u + 1u;
    static const uint32_t bands_env = rd_env_u32("RD_RENDER_BANDS", RD_BANDS_MAX);
    uint32_t bands = hist ? 1u : (bands_env < RD_BANDS_MAX ? bands_env : RD_BANDS_MAX);
    if (bands > units) bands = units;
    if (!bands) bands = 1u;
    size_t band_end[RD_BANDS_MAX];                           // bytes of the surface complete after band k
    for (uint32_t k = 0; k < bands; ++k) {
        const uint32_t u1 = (uint32_t)(((uint64_t)units * (k + 1)) / bands);
        const uint32_t row_hi = 2u * (u1 - 1u) < H ? 2u * (u1 - 1u) + 1u : H;      // exclusive: the last unit's row b
        band_end[k] = k + 1u == bands ? need : (size_t)row_hi * row_bytes;
    }
    // The chunks of the read-back.  Page-locked destination: band-aligned pieces that grow (bands 0 | 1 | 2-3 | 4-7: the first
    // bytes leave as soon as band 0 is done, the tail is two large transfers; RD_COPY_CHUNK_MB = fixed-size pieces instead).
    // Pageable destination: RD_STAGE_BYTES pieces through the staging slots.
    const bool pinned = rd_is_pinned_host(dst, need);
    static const size_t chunk_fixed = (size_t)rd_env_u32("RD_COPY_CHUNK_MB", 0) << 20;
    struct piece { size_t off, len; };
    piece grown[RD_BANDS_MAX];
    size_t npieces = 0;
    const bool growing = pinned && !chunk_fixed;
    const size_t chunk = pinned ? chunk_fixed : RD_STAGE_BYTES;
    if (growing) {
        size_t off = 0;
        for (uint32_t k = 0, group = 1, used = 0; k < bands; ++k) {
            if (++used == group || k + 1u == bands) {
                grown[npieces++] = piece{ off, band_end[k] - off };
                off = band_end[k];
                used = 0;
                if (npieces >= 2) group *= 2u;
            }
        }
    } else {
        npieces = (need + chunk - 1) / chunk;
    }
    auto piece_of = [&](size_t c) -> piece {
        if (growing) return grown[c];
        return piece{ c * chunk, need - c * chunk < chunk ? need - c * chunk : chunk };
    };
    uint32_t waited = 0;                                     // bands [0, waited) are already ordered before the copy stream's tail
    auto enqueue_piece = [&](size_t c, void *to) -> hipError_t {
        hipError_t e = hipSuccess;
        const piece pc = piece_of(c);
        while (e == hipSuccess && waited < bands && (waited == 0 || band_end[waited - 1u] < pc.off + pc.len))
            e = hipStreamWaitEvent(l->copy, l->kev[waited++], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(to, (const char *)l->out_buf + pc.off, pc.len, hipMemcpyDeviceToHost, l->copy);
        return e;
    };
    const bool staged = !pinned;
    bool stage_ok = true;
    if (staged) {
        for (int j = 0; j < RD_STAGE_SLOTS; ++j)
            if (!l->stage[j] && hipHostMalloc(&l->stage[j], RD_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) stage_ok = false;
        if (!stage_ok) return rd_fail(RD_ERR_OOM, "pinned staging allocation failed");
    }
    // Launch the bands; a piece is enqueued the moment the last band it needs has been (the host spends ~10 us per launch:
    // waiting for all eight before the first copy is even enqueued would cost the transfer that long).
    size_t next = 0;                                         // next piece to enqueue
    hipError_t e = hipSuccess;
    int rc = RD_OK;
    for (uint32_t k = 0; k < bands && rc == RD_OK && e == hipSuccess; ++k) {
        const uint32_t u0 = (uint32_t)(((uint64_t)units * k) / bands), u1 = (uint32_t)(((uint64_t)units * (k + 1)) / bands);
        rc = rd_pipeline_enqueue(p, sh, W, H, fmt, l->out_buf, hist ? l->hist_dev : nullptr, l->compute, u0, u1);
        if (rc == RD_OK) e = hipEventRecord(l->kev[k], l->compute);
        while (rc == RD_OK && e == hipSuccess && next < npieces && (pinned || next < RD_STAGE_SLOTS) &&
               piece_of(next).off + piece_of(next).len <= band_end[k]) {
            e = enqueue_piece(next, pinned ? (void *)(dst + piece_of(next).off) : l->stage[next % RD_STAGE_SLOTS]);
            if (e == hipSuccess && staged) e = hipEventRecord(l->cev[next % RD_STAGE_SLOTS], l->copy);
            ++next;
        }
    }
    if (rc) { (void)hipStreamSynchronize(l->copy); (void)hipStreamSynchronize(l->compute); return rc; }
    if (pinned) {
        if (e == hipSuccess && hist) {
            e = hipStreamWaitEvent(l->copy, l->kev[0], 0);
            if (e == hipSuccess) e = hipMemcpyAsync(hist, l->hist_dev, 768 * sizeof(uint32_t), hipMemcpyDeviceToHost, l->copy);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(l->copy);
    } else {
        rd_copy_pool &pool = rd_copy_pool::get();
        rd_advise_destination(dst, need);
        for (size_t c = 0; c < npieces && e == hipSuccess; ++c) {
            const size_t j = c % RD_STAGE_SLOTS;
            e = hipEventSynchronize(l->cev[j]);
            if (e != hipSuccess) break;
            pool.copy(dst + piece_of(c).off, l->stage[j], piece_of(c).len);
            if (next < npieces) {                            // the slot is free again: the next piece goes into it (next == c + slots)
                e = enqueue_piece(next, l->stage[next % RD_STAGE_SLOTS]);
                if (e == hipSuccess) e = hipEventRecord(l->cev[next % RD_STAGE_SLOTS], l->copy);
                ++next;
            }
        }
        if (e == hipSuccess && hist) {
            e = hipMemcpyAsync(hist, l->hist_dev, 768 * sizeof(uint32_t), hipMemcpyDeviceToHost, l->compute);
            if (e == hipSuccess) e = hipStreamSynchronize(l->compute);
        }
    }
    if (e != hipSuccess) {                                    // whatever ran may have stopped half way: counters are suspect
        (void)hipStreamSynchronize(l->copy);                  // nothing may still be writing `dst` when the caller gets it back
        (void)hipStreamSynchronize(l->compute);
        p->scratch.mark_all_dirty();
        return rd_fail(RD_ERR_HIP, "render readback failed: %s", hipGetErrorString(e));
    }
    return RD_OK;
}

extern "C" int rd_render(rd_pipeline *p, uint32_t out_w, uint32_t out_h, uint32_t fmt, void *dst, size_t dst_len,
                         uint32_t hist[768]) try
{
    RD_ENTRY(rd_render);
    if (!p || !dst) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    const size_t bpp = rd_format_bytes_per_pixel(fmt);
    if (!bpp) return rd_fail(RD_ERR_INVALID_ARG, "unknown format %u", fmt);
    const size_t need = (size_t)out_w * out_h * bpp;
    if (!need) return rd_fail(RD_ERR_INVALID_ARG, "empty target %ux%u", out_w, out_h);
    if (dst_len != need) return rd_fail(RD_ERR_INVALID_ARG, "dst_len %zu != %ux%ux%zu = %zu", dst_len, out_w, out_h, bpp, need);
    rd_devguard g(p->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", p->device);
    if (need >= RD_BAND_MIN_BYTES && rd_host_memory_kind(dst, need) == RD_MEM_DEVICE)      // (the staged path would memcpy into it)
        return rd_fail(RD_ERR_INVALID_ARG, "dst is device memory: rd_render writes host buffers (rd_render_device takes device pointers)");
    const rd_shot sh = rd_pipeline_snapshot(p);               // the pipeline's mutex is held for this line only
    rd_lane_hold hold(p, need);
    rd_lane *l = hold.l;
    if (!l) return hold.rc;
    int rc = rd_lane_reserve(l, need);
    if (rc) return rc;
    if (need >= RD_BAND_MIN_BYTES && rd_pipeline_uses_quads(p, sh, out_w, out_h, fmt))
        return rd_render_full_host(p, l, sh, fmt, (char *)dst, need, hist);
    rc = rd_pipeline_enqueue(p, sh, out_w, out_h, fmt, l->out_buf, hist ? l->hist_dev : nullptr, l->compute);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(dst, l->out_buf, need, hipMemcpyDeviceToHost, l->compute);
    if (e == hipSuccess && hist) e = hipMemcpyAsync(hist, l->hist_dev, 768 * sizeof(uint32_t), hipMemcpyDeviceToHost, l->compute);
    if (e == hipSuccess) e = hipStreamSynchronize(l->compute);
    if (e != hipSuccess) {                                    // whatever ran may have stopped half way: counters are suspect
        p->scratch.mark_all_dirty();
        return rd_fail(RD_ERR_HIP, "render readback failed: %s", hipGetErrorString(e));
    }
    return RD_OK;
}
RD_CATCH_INT(rd_render)

extern "C" int rd_render_to_bytes(rd_pipeline *p, uint8_t *dst, size_t dst_len) try
{
    RD_ENTRY(rd_render_to_bytes);
    if (!p) return rd_fail(RD_ERR_INVALID_ARG, "NULL pipeline");
    return rd_render(p, p->info.preview_width, p->info.preview_height, RD_FMT_RGBA_U8, dst, dst_len, nullptr);
}
RD_CATCH_INT(rd_render_to_bytes)

extern "C" int rd_render_full_res_to_bytes(rd_pipeline *p, uint8_t *dst, size_t dst_len) try
{
    RD_ENTRY(rd_render_full_res_to_bytes);
    if (!p) return rd_fail(RD_ERR_INVALID_ARG, "NULL pipeline");
    return rd_render(p, p->info.width, p->info.height, RD_FMT_RGBA_U8, dst, dst_len, nullptr);
}
RD_CATCH_INT(rd_render_full_res_to_bytes)

// render_full_res_to_bytes without the caller's allocation: the surface is rendered into page-locked memory the PIPELINE owns
// (allocated on first use, reused afterwards: pinning 96.6 MB costs milliseconds, a fresh pageable Vec its page faults) and
// lent to the caller until rd_surface_release.  What export_image_async needs -- a &[u8] for image::save_buffer
// (main.rs:1765-1791) -- at the price of the PCIe transfer.  Up to RD_LENT_MAX surfaces may be out at a time.
extern "C" int rd_render_full_res_borrow(rd_pipeline *p, const uint8_t **data, size_t *len) try
{
    RD_ENTRY(rd_render_full_res_borrow);
    if (!p || !data) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    *data = nullptr;
    const size_t need = (size_t)p->info.width * p->info.height * 4u;
    rd_devguard g(p->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", p->device);
    void *buf = nullptr;
    {
        std::lock_guard<std::mutex> lk(p->lent_mu);
        for (auto &b : p->lents)
            if (!b.busy && b.cap >= need) { b.busy = true; buf = b.ptr; break; }
        if (!buf) {
            if (p->lents.size() >= RD_LENT_MAX) return rd_fail(RD_ERR_INVALID_ARG, "%d borrowed surfaces have not been released", RD_LENT_MAX);
            rd_pipeline::lent b;
            RD_HIP(hipHostMalloc(&b.ptr, need, hipHostMallocDefault));
            b.cap = need; b.busy = true;
            p->lents.push_back(b);
            buf = b.ptr;
        }
    }
    const int rc = rd_render(p, p->info.width, p->info.height, RD_FMT_RGBA_U8, buf, need, nullptr);
    if (rc) {
        std::lock_guard<std::mutex> lk(p->lent_mu);
        for (auto &b : p->lents) if (b.ptr == buf) b.busy = false;
        return rc;
    }
    *data = (const uint8_t *)buf;
    if (len) *len = need;
    return RD_OK;
}
RD_CATCH_INT(rd_render_full_res_borrow)

extern "C" int rd_surface_release(rd_pipeline *p, const uint8_t *data) try
{
    RD_ENTRY(rd_surface_release);
    if (!p) return rd_fail(RD_ERR_INVALID_ARG, "NULL pipeline");
    if (!data) return RD_OK;
    std::lock_guard<std::mutex> lk(p->lent_mu);
    for (auto &b : p->lents)
        if (b.ptr == (const void *)data) {
            if (!b.busy) return rd_fail(RD_ERR_INVALID_ARG, "surface released twice");
            b.busy = false;
            return RD_OK;
        }
    return rd_fail(RD_ERR_INVALID_ARG, "not a surface borrowed from this pipeline");
}
RD_CATCH_INT(rd_surface_release)

extern "C" int rd_render_to_histogram_bytes(rd_pipeline *p, uint8_t *dst, size_t dst_len) try
{
    RD_ENTRY(rd_render_to_histogram_bytes);
    if (!p) return rd_fail(RD_ERR_INVALID_ARG, "NULL pipeline");
    return rd_render(p, p->info.histogram_width, p->info.histogram_height, RD_FMT_RGBA_U8, dst, dst_len, nullptr);
}
RD_CATCH_INT(rd_render_to_histogram_bytes)

extern "C" int rd_calculate_histogram(rd_pipeline *p, const uint8_t *rgba, size_t rgba_len, uint32_t hist[768]) try
{
    RD_ENTRY(rd_calculate_histogram);
    if (!p || !hist || (!rgba && rgba_len)) return rd_fail(RD_ERR_INVALID_ARG, "NULL argument");
    if (rgba_len % 4) return rd_fail(RD_ERR_INVALID_ARG, "rgba_len %zu is not a multiple of 4", rgba_len);
    const size_t npx = rgba_len / 4;
    if (npx >= 0xffffffffull) return rd_fail(RD_ERR_UNSUPPORTED, "too many pixels");
    if (!npx) { memset(hist, 0, 768 * sizeof(uint32_t)); return RD_OK; }
    rd_devguard g(p->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", p->device);
    rd_lane_hold hold(p, rgba_len);
    rd_lane *ln = hold.l;
    if (!ln) return hold.rc;
    int rc = rd_lane_reserve(ln, rgba_len);
    if (rc) return rc;
    RD_HIP(hipMemcpyAsync(ln->out_buf, rgba, rgba_len, hipMemcpyHostToDevice, ln->compute));
    const uint32_t blocks = rd_blocks_for(p->cfg, npx, true);
    const rd_scratch::lease l = p->scratch.get(ln->compute, true);
    if (l.idx < 0) return rd_fail(RD_ERR_OOM, "scheduler state allocation failed");
    hipLaunchKernelGGL(rd_hist_u8, dim3(blocks), dim3(RD_BLOCK), 0, ln->compute, (const uint32_t *)ln->out_buf,
                       (uint32_t)npx, l.slab32);
    hipLaunchKernelGGL(rd_reduce_slab32, dim3(24), dim3(RD_FOLD_THREADS), 0, ln->compute, l.slab32, blocks, ln->hist_dev);
    const hipError_t le = hipGetLastError();
    p->scratch.used(l, ln->compute, le != hipSuccess);
    RD_HIP(le);
    RD_HIP(hipMemcpyAsync(hist, ln->hist_dev, 768 * sizeof(uint32_t), hipMemcpyDeviceToHost, ln->compute));
    RD_HIP(hipStreamSynchronize(ln->compute));
    return RD_OK;
}
RD_CATCH_INT(rd_calculate_histogram)

// Page-locked host memory for render destinations (and sources): what a host that wants the direct-DMA path allocates
// its surface buffer from.  Any thread, any time; rd_host_free(NULL) is a no-op.
extern "C" int rd_host_alloc(int device, size_t bytes, void **out) try
{
    RD_ENTRY(rd_host_alloc);
    if (!out) return rd_fail(RD_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return RD_OK;
}
RD_CATCH_INT(rd_host_alloc)

extern "C" int rd_host_free(int device, void *ptr) try
{
    RD_ENTRY(rd_host_free);
    if (!ptr) return RD_OK;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipHostFree(ptr));
    return RD_OK;
}
RD_CATCH_INT(rd_host_free)

