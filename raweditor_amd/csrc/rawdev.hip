// rawdev.hip -- host side of librawdev.so: the extern "C" entry points of include/rawdev.h.
//
// Stands in for gpu::RenderPipeline (reference src/gpu/pipeline.rs:112-737).  Differences that
// are deliberate and MI355X-first:
//   * the reference creates a wgpu Instance+Device per image (pipeline.rs:144-169) and a target
//     texture + MAP_READ buffer per frame (:444-477); here a pipeline keeps one HBM copy of the
//     CFA plane, one reusable output buffer and one stream;
//   * uniforms are kernel arguments (no uniform buffer, no write_buffer);
//   * linear buffers, not textures: no 8192-px texture limit (pipeline.rs:164), so 100 MP frames
//     need no special casing.
// There is no CPU compute path in this library: every render is a gfx950 kernel launch.
// Layout (one translation unit): this file = errors, device bookkeeping, launch plumbing, ingest helper, plumbing, test hooks;
// rd_host_pipeline.inl = rd_pipeline; rd_host_batch.inl = rd_batch / rd_node_batch / rd_exporter; rd_host_diag.inl = self-tests
// and measurement aids.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/rawdev.h"
#include "rd_kernels.h"
#include "rd_ljpeg.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int rd_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define RD_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return rd_fail(e_ == hipErrorOutOfMemory ? RD_ERR_OOM                                 \
                           : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? RD_ERR_NO_DEVICE \
                                                                                     : RD_ERR_HIP, \
                           "%s failed: %s", #call, hipGetErrorString(e_));                        \
    } while (0)

// ------------------------------------------------------------------------------------------------
// No C++ exception crosses the C ABI ("Errors": never abort, never throw across the ABI; the
// reference's constructor returns Result<Self, String>, pipeline.rs:122, :156, :169).
//
// Every extern "C" definition of this library is a function-try-block that ends in RD_CATCH_INT / _VOID / _VAL: whatever
// the body throws -- std::bad_alloc from a growing vector, std::system_error from a thread or mutex, anything else -- becomes
// a status (RD_ERR_OOM / RD_ERR_INTERNAL) and a message in rd_last_error(); destructors of the body's locals (lock guards,
// device guards, the owning pointers of half-built objects) have run by then.  tests/test_host_cpu.py checks the source: no
// extern "C" without the pattern.
//
// RD_ENTRY(name) is a FAULT POINT: a place where a test can make the library throw, to prove the above (and that the
// object at hand stays usable).  rd_debug_inject_fault(site, kind, after) -- or RD_FAULT_INJECT=site:kind[:after] in the
// environment -- arms one fault: the (after + 1)-th passage through a fault point whose name is `site` ("*" = any) throws
// once and disarms.  Beside the entry points there are fault points at the places that really allocate or start threads
// (RD_FAULT_POINT("node.thread"), "batch.descs", "pipeline.lanes", ...).  Unarmed, a fault point is one relaxed load.
// ------------------------------------------------------------------------------------------------
namespace {
struct rd_fault_state {
    std::atomic<uint32_t> armed{ 0 };
    std::mutex mu;
    char site[64] = "";
    uint32_t kind = 0, after = 0;
};
rd_fault_state &rd_fault()
{
    static rd_fault_state st;
    return st;
}
struct rd_injected_fault {};                                     // RD_FAULT_FOREIGN: not derived from std::exception

void rd_fault_arm(const char *site, uint32_t kind, uint32_t after)
{
    rd_fault_state &st = rd_fault();
    std::lock_guard<std::mutex> lk(st.mu);
    if (!site || !*site || !kind) { st.armed.store(0, std::memory_order_relaxed); st.site[0] = 0; return; }
    snprintf(st.site, sizeof st.site, "%s", site);
    st.kind = kind; st.after = after;
    st.armed.store(1, std::memory_order_release);
}

[[gnu::noinline]] void rd_fault_fire(const char *site)
{
    rd_fault_state &st = rd_fault();
    uint32_t kind = 0;
    {
        std::lock_guard<std::mutex> lk(st.mu);
        if (!st.armed.load(std::memory_order_relaxed)) return;
        if (strcmp(st.site, "*") != 0 && strcmp(st.site, site) != 0) return;
        if (st.after) { st.after -= 1; return; }
        kind = st.kind;
        st.armed.store(0, std::memory_order_relaxed);            // one shot
    }
    switch (kind) {
    case RD_FAULT_BAD_ALLOC: throw std::bad_alloc();
    case RD_FAULT_THREAD_START: throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "injected thread-start failure");
    case RD_FAULT_RUNTIME: throw std::runtime_error(std::string("injected fault at ") + site);
    default: throw rd_injected_fault{};
    }
}

inline void rd_fault_point(const char *site)
{
    if (__builtin_expect(rd_fault().armed.load(std::memory_order_relaxed) != 0, 0)) rd_fault_fire(site);
}

const int rd_fault_env_init = [] {                               // RD_FAULT_INJECT=site:kind[:after]
    const char *e = getenv("RD_FAULT_INJECT");
    if (!e || !*e) return 0;
    char site[64];
    unsigned kind = 0, after = 0;
    const char *c = strchr(e, ':');
    if (!c || (size_t)(c - e) >= sizeof site) return 0;
    memcpy(site, e, (size_t)(c - e)); site[c - e] = 0;
    if (sscanf(c + 1, "%u:%u", &kind, &after) < 1) return 0;
    rd_fault_arm(site, kind, after);
    return 1;
}();

// The handler of every entry point (a Lippincott function: rethrows the exception in flight to sort it).
int rd_caught(const char *name) noexcept
{
    try { throw; }
    catch (const std::bad_alloc &) { return rd_fail(RD_ERR_OOM, "%s: host allocation failed (std::bad_alloc)", name); }
    catch (const std::exception &e) { return rd_fail(RD_ERR_INTERNAL, "%s: C++ exception stopped at the C boundary: %s", name, e.what()); }
    catch (...) { return rd_fail(RD_ERR_INTERNAL, "%s: unknown C++ exception stopped at the C boundary", name); }
}
}  // namespace

#define RD_FAULT_POINT(site) rd_fault_point(site)
#define RD_ENTRY(name) rd_fault_point(#name)
#define RD_CATCH_INT(name) catch (...) { return rd_caught(#name); }
#define RD_CATCH_VOID(name) catch (...) { (void)rd_caught(#name); }
#define RD_CATCH_VAL(name, value) catch (...) { (void)rd_caught(#name); return value; }

extern "C" int rd_abi_version(void) try { return RD_ABI_VERSION; } RD_CATCH_INT(rd_abi_version)
extern "C" const char *rd_last_error(void) try { return g_err; } RD_CATCH_VAL(rd_last_error, g_err)      // (no fault point: it would overwrite what it returns)

extern "C" int rd_device_count(int *count) try
{
    RD_ENTRY(rd_device_count);
    if (!count) return rd_fail(RD_ERR_INVALID_ARG, "rd_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return rd_fail(RD_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return RD_OK;
}
RD_CATCH_INT(rd_device_count)

// PCI bus id ("0000:c1:00.0") and marketing name of a visible device: what tells two ranks of a multi-GPU run apart.
extern "C" int rd_device_identity(int device, char *pci_bus_id, size_t pci_cap, char *name, size_t name_cap) try
{
    RD_ENTRY(rd_device_identity);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return rd_fail(RD_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return rd_fail(RD_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    RD_HIP(hipGetDeviceProperties(&prop, device));
    if (pci_bus_id && pci_cap) snprintf(pci_bus_id, pci_cap, "%04x:%02x:%02x.0", prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    return RD_OK;
}
RD_CATCH_INT(rd_device_identity)

extern "C" void rd_edit_params_default(rd_edit_params *p) try
{
    RD_ENTRY(rd_edit_params_default);
    if (!p) return;
    memset(p, 0, sizeof *p);   // edit.rs:81-95
    p->whites = 1.0f;
}
RD_CATCH_VOID(rd_edit_params_default)

extern "C" int rd_derived_dims(uint32_t w, uint32_t h, uint32_t *pw, uint32_t *ph, uint32_t *hw, uint32_t *hh) try
{
    RD_ENTRY(rd_derived_dims);
    if (!w || !h || !pw || !ph || !hw || !hh) return rd_fail(RD_ERR_INVALID_ARG, "rd_derived_dims: bad argument");
    // pipeline.rs:125-133, same truncating f32 arithmetic
    const float aspect = (float)w / (float)h;
    const uint32_t preview_w = w < 1280u ? w : 1280u;
    *pw = preview_w;
    *ph = (uint32_t)((float)preview_w / aspect);
    *hw = 128u;
    *hh = (uint32_t)((float)128u / aspect);
    return RD_OK;
}
RD_CATCH_INT(rd_derived_dims)

extern "C" size_t rd_format_bytes_per_pixel(uint32_t f) try
{
    RD_ENTRY(rd_format_bytes_per_pixel);
    return f == RD_FMT_RGBA_F32 ? 16 : f == RD_FMT_RGBA_F16 ? 8 : f == RD_FMT_RGBA_U8 ? 4 : f == RD_FMT_RGB_U8 ? 3 : 0;
}
RD_CATCH_VAL(rd_format_bytes_per_pixel, 0)

extern "C" uint32_t rd_elided_steps(const rd_edit_params *p, const float wb[4], const float cm[9], uint32_t math_mode) try
{
    RD_ENTRY(rd_elided_steps);
    if (!p || !wb || !cm) return 0u;
    return rd_make_ku(*p, wb, cm, 1.0f, 0.0f, 0.0f, 0u, math_mode).elide;
}
RD_CATCH_VAL(rd_elided_steps, 0)

// ------------------------------------------------------------------------------------------------
// device bookkeeping
// ------------------------------------------------------------------------------------------------
struct rd_devguard {                             // the calling thread's current device, for the scope of one entry point
    int prev = -1;
    bool ok = false;
    explicit rd_devguard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) { ok = true; prev = -1; }           // already current (the usual case): nothing to set, nothing to restore
        else ok = hipSetDevice(dev) == hipSuccess;
    }
    ~rd_devguard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static int rd_check_device(int device, int *n_cu)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return rd_fail(RD_ERR_NO_DEVICE, "no HIP device visible (librawdev has no CPU fallback)");
    if (device < 0 || device >= n) return rd_fail(RD_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    RD_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rd_fail(RD_ERR_UNSUPPORTED, "device %d is %s; librawdev ships gfx950 code only", device, prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    return RD_OK;
}

// The 8-bit surfaces' threshold table (rd_kernels.h, rd_q8_lut_*): built once per process from the pinned gamma, copied into
// each device's rd_q8_lut_dev the first time that device is used.  The device is current.
static int rd_q8_lut_ensure(int device)
{
    static std::mutex mu;
    static std::vector<uint32_t> table;
    static bool done[64] = {};
    if (device < 0 || device >= 64) return rd_fail(RD_ERR_NO_DEVICE, "device %d out of range", device);
    std::lock_guard<std::mutex> lk(mu);
    if (done[device]) return RD_OK;
    if (table.empty()) { table.assign(RD_Q8_LUT_WORDS + 63u, 0u); rd_q8_lut_build(table.data()); }
    RD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rd_q8_lut_dev), table.data(), table.size() * sizeof(uint32_t)));
    done[device] = true;
    return RD_OK;
}

// The f16 surface's two-level tables (rd_kernels.h, rd_f16_lut_*): the same arrangement, but built (about 40 ms of host time,
// once per process) only when something is about to render an RGBA-f16 surface with the export kernel.  A pinned function
// whose shape the lookup does not assume (rd_f16_lut_build's checks) is refused here, loudly.
static int rd_f16_lut_ensure(int device)
{
    static std::mutex mu;
    static std::vector<uint16_t> fine;
    static std::vector<uint32_t> coarse;
    static int built = 1;                                        // 1: not yet; 0: fine; < 0: refused
    static bool done[64] = {};
    if (device < 0 || device >= 64) return rd_fail(RD_ERR_NO_DEVICE, "device %d out of range", device);
    std::lock_guard<std::mutex> lk(mu);
    if (done[device]) return RD_OK;
    if (built == 1) {
        fine.assign(RD_F16_LUT_NF + 1u + 128u, 0u);
        coarse.assign(RD_F16_LUT_NC * 2u + 64u, 0u);
        built = rd_f16_lut_build(fine.data(), coarse.data());
    }
    if (built != 0) return rd_fail(RD_ERR_UNSUPPORTED, "the binary16 threshold tables cannot be built from this gamma (check %d)", built);
    RD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rd_f16_fine_dev), fine.data(), ((RD_F16_LUT_NF + 1u) / 2u + 64u) * sizeof(uint32_t)));
    RD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rd_f16_coarse_dev), coarse.data(), coarse.size() * sizeof(uint32_t)));
    done[device] = true;
    return RD_OK;
}

static uint32_t rd_env_u32(const char *name, uint32_t dflt)
{
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    long v = strtol(s, nullptr, 10);
    return v > 0 ? (uint32_t)v : dflt;
}

// The uniforms of one frame.  `layout` = RD_MATRIX_REFERENCE hands the host's row-major matrix to the kernel as it is (its
// rows are then consumed as COLUMNS, shaders.rs:209-214: out = M^T c -- the reference's behaviour); RD_MATRIX_ROW_MAJOR is the
// opt-in "intended" form out = M c, obtained by transposing on the host (SURVEY.md D4, section 8b rd_options.matrix_layout).
static rd_ku rd_frame_ku(const rd_edit_params &p, const float wb[4], const float cm[9], float zoom, float pan_x, float pan_y,
                         uint32_t black_level, uint32_t math_mode, uint32_t layout)
{
    if (layout == RD_MATRIX_ROW_MAJOR) {
        const float t[9] = { cm[0], cm[3], cm[6], cm[1], cm[4], cm[7], cm[2], cm[5], cm[8] };
        return rd_make_ku(p, wb, t, zoom, pan_x, pan_y, black_level, math_mode);
    }
    return rd_make_ku(p, wb, cm, zoom, pan_x, pan_y, black_level, math_mode);
}

// ------------------------------------------------------------------------------------------------
// launch plumbing shared by pipelines and batches
// ------------------------------------------------------------------------------------------------
// Does the reference's f32 pixel map (shaders.rs:31-57, :184-187) reduce to px=i, py=j when the
// target is the frame itself at zoom 1 / pan 0?  (It does for every size we have seen; the check
// keeps the quad kernel honest for sizes where f32 rounding could break it.)
static bool rd_identity_map(uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) {
        float s = ((float)i + 0.5f) / (float)n;
        float t = ((s - 0.5f) / 1.0f - 0.0f) + 0.5f;
        if (!(t >= 0.0f && t <= 1.0f)) return false;
        if ((int32_t)(t * (float)n) != (int32_t)i) return false;
    }
    return true;
}

struct rd_launch_cfg {
    int n_cu = 256;
    uint32_t wg_per_cu_hist = 2;     // 24 KiB histogram + 48 KiB store stage per workgroup
    uint32_t wg_per_cu_plain = 2;    // 2 x 1024 threads = the CU's 32 waves
};

// Per-stream scheduler state of the export kernel.
//   * Ticket counters (rd_kernels.h, "Scheduling"): up to RD_MAX_BLOCKS / 4 counters, one per 128-byte line.  They must be
//     zero when a launch starts; the kernel resets what it used, so launches on ONE stream (ordered) can share a set,
//     launches that may overlap cannot.
//   * The u32 histogram slab of single-frame renders (RD_MAX_BLOCKS x 768), allocated on first use.
// A context keeps one entry per stream it has been used with, at most max_entries of them: beyond that the least
// recently used entry whose work has finished (its `done` event, recorded after every use) is handed to the new stream.
// An entry is marked dirty when a launch on it failed or a synchronisation reported an error -- the counters may then be
// anything -- and is re-zeroed on its stream before the next launch.
struct rd_scratch {
    struct entry {
        hipStream_t stream = nullptr;
        uint32_t *tq = nullptr;
        uint32_t *slab32 = nullptr;
        hipEvent_t done = nullptr;
        bool recorded = false;                   // `done` marks the entry's LAST use (see used())
        bool dirty = false;
        uint64_t stamp = 0;
    };
    struct lease { uint32_t *tq = nullptr; uint32_t *slab32 = nullptr; int idx = -1; };
    static constexpr size_t tq_bytes = (size_t)(RD_MAX_BLOCKS / 4) * RD_TQ_STRIDE * sizeof(uint32_t);
    static constexpr size_t slab_bytes = (size_t)RD_MAX_BLOCKS * 768 * sizeof(uint32_t);
    static constexpr size_t max_entries = 16;
    std::mutex mu;
    std::vector<entry> ents;
    uint64_t clock = 0;
    rd_scratch() { ents.reserve(max_entries); }                  // so that no push_back below can throw with device memory in hand

    // The entry of stream s, created / recycled / re-zeroed as needed.  idx < 0: allocation failed.
    lease get(hipStream_t s, bool want_slab)
    {
        std::lock_guard<std::mutex> lk(mu);
        int idx = -1;
        for (size_t i = 0; i < ents.size(); ++i)
            if (ents[i].stream == s) idx = (int)i;
        if (idx < 0 && ents.size() < max_entries) {
            RD_FAULT_POINT("scratch.entry");
            entry e;
            if (hipMalloc((void **)&e.tq, tq_bytes) != hipSuccess) return lease{};
            if (hipMemset(e.tq, 0, tq_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipEventCreateWithFlags(&e.done, hipEventDisableTiming) != hipSuccess) {
                (void)hipFree(e.tq);
                return lease{};
            }
            e.stream = s;
            ents.push_back(e);
            idx = (int)ents.size() - 1;
        }
        if (idx < 0) {                                           // recycle: least recently used among the finished ones
            // An entry whose last use carries no marker -- it was used while the table still had room, see used() -- is asked
            // through ITS OWN stream (hipStreamQuery: finished / not ready / the stream no longer exists, whose work is then
            // over too).  Nothing here synchronises the device: this runs inside rd_render_device, an asynchronous enqueue that
            // other streams of the device must not stall behind and that may sit next to a stream capture.
            auto finished = [](entry &e) -> bool {
                if (e.recorded) return hipEventQuery(e.done) == hipSuccess;
                const hipError_t q = hipStreamQuery(e.stream);
                if (q != hipSuccess) (void)hipGetLastError();
                return q != hipErrorNotReady;                    // success, or a stream that has been destroyed
            };
            int lru_done = -1, lru_any = 0;
            for (size_t i = 0; i < ents.size(); ++i) {
                if (ents[i].stamp < ents[(size_t)lru_any].stamp) lru_any = (int)i;
                if (finished(ents[i]) && (lru_done < 0 || ents[i].stamp < ents[(size_t)lru_done].stamp)) lru_done = (int)i;
            }
            idx = lru_done >= 0 ? lru_done : lru_any;
            if (lru_done < 0) {                                  // every entry busy: wait for the least recently used one alone
                entry &v = ents[(size_t)idx];
                const hipError_t w = v.recorded ? hipEventSynchronize(v.done) : hipStreamSynchronize(v.stream);
                if (w != hipSuccess) { (void)hipGetLastError(); v.dirty = true; }
            }
            ents[(size_t)idx].stream = s;
        }
        entry &e = ents[(size_t)idx];
        if (e.dirty) {
            if (hipMemsetAsync(e.tq, 0, tq_bytes, s) != hipSuccess) return lease{};
            e.dirty = false;
        }
        if (want_slab && !e.slab32 && hipMalloc((void **)&e.slab32, slab_bytes) != hipSuccess) return lease{};
        e.stamp = ++clock;
        return lease{ e.tq, e.slab32, idx };
    }
    void used(const lease &l, hipStream_t s, bool failed)       // after the launches of one call on entry l
    {
        std::lock_guard<std::mutex> lk(mu);
        if (l.idx < 0 || (size_t)l.idx >= ents.size()) return;
        entry &e = ents[(size_t)l.idx];
        if (failed) e.dirty = true;
        // The marker is what recycling asks ("has this entry's last use finished?"), and nothing is recycled while the table
        // has room: until then the call saves itself the event (one barrier packet per render on the launch stream -- it sat
        // between a single-frame render's last kernel and the caller's own event).
        if (ents.size() < max_entries) { e.recorded = false; return; }
        e.recorded = hipEventRecord(e.done, s) == hipSuccess;
        if (!e.recorded) e.dirty = true;
    }
    void mark_all_dirty()
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : ents) e.dirty = true;
    }
    bool poison(hipStream_t s)                                   // test hook: garbage in the counters, as after an aborted launch
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : ents)
            if (e.stream == s) { e.dirty = true; return hipMemsetAsync(e.tq, 0xa5, tq_bytes, s) == hipSuccess; }
        return false;
    }
    size_t size() { std::lock_guard<std::mutex> lk(mu); return ents.size(); }
    void release()                                               // caller has synchronised the device
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : ents) {
            if (e.tq) (void)hipFree(e.tq);
            if (e.slab32) (void)hipFree(e.slab32);
            if (e.done) (void)hipEventDestroy(e.done);
        }
        ents.clear();
    }
};

// One launch of rd_develop_quads as data: the kernel instance, its grid and its 14 arguments.  rd_launch_quads_t fills
// one and either launches it or hands it back (`record`) so that the caller can put it into a graph node.
struct rd_quads_call {
    const void *fn = nullptr;
    uint32_t blocks = 0;
    const uint16_t *cfa; void *out; uint32_t W, H, unit0, unit1, tpu, tpu_magic, tq_k, tq_tmax; uint32_t *tq; rd_ku u;
    uint32_t *slab32; unsigned long long *slab64;
    void *argv[14];
    void bind()
    {
        void *a[14] = { &cfa, &out, &W, &H, &unit0, &unit1, &tpu, &tpu_magic, &tq_k, &tq_tmax, &tq, &u, &slab32, &slab64 };
        memcpy(argv, a, sizeof a);
    }
};

// How the export kernel tiles a row of W pixels (W even; rd_kernels.h, TILES).  RD_TILES=overlap (A/B switch) runs the
// pulled-back-last-tile instance on widths that need none.
static int rd_tiles_mode(uint32_t W, uint32_t fmt)
{
    static const bool force_overlap = getenv("RD_TILES") && !strcmp(getenv("RD_TILES"), "overlap");
    static const bool no_shift = getenv("RD_TILES") && !strcmp(getenv("RD_TILES"), "noshift");   // A/B: such f32 frames through OVERLAP
    if (W < 128u) return RD_TILES_MASKED;
    // f32 rows that do not start on 64-byte boundaries (W % 4 != 0; no camera makes one, a crop can): shifted store windows
    if ((W & 3u) && fmt == RD_FMT_RGBA_F32 && !no_shift) return RD_TILES_SHIFT;
    return (W % 128u == 0 && !force_overlap) ? RD_TILES_WHOLE : RD_TILES_OVERLAP;
}

// Tiles per unit (row pair): 64-quad tiles that abut or whose last one is pulled back; RD_TILES_SHIFT's tiles own 62 quads.
static uint32_t rd_tiles_per_unit(uint32_t W, uint32_t fmt)
{
    const uint32_t qpr = W >> 1;
    return rd_tiles_mode(W, fmt) == RD_TILES_SHIFT ? (qpr + 61u) / 62u : (qpr + 63u) / 64u;
}

template <int FMT, bool HIST, int MATH>
static void rd_launch_quads_t(const uint16_t *cfa, void *out, uint32_t W, uint32_t H, uint32_t unit0,
                              uint32_t unit1, uint32_t blocks, const rd_ku &u_in, uint32_t *slab32,
                              unsigned long long *slab64, uint32_t *tq, hipStream_t s, rd_quads_call *record)
{
    static const bool no_elide = rd_env_u32("RD_NO_ELIDE", 0) != 0;    // A/B switch: evaluate every step (rd_uniforms.h RD_EL_*)
    rd_quads_call local;
    rd_quads_call &c = record ? *record : local;
    c.u = u_in;
    if (no_elide) c.u.elide = 0u;
    const uint32_t tpu = rd_tiles_per_unit(W, FMT);        // 64-quad tiles per unit
    c.tpu = tpu;
    c.tpu_magic = tpu > 1u ? (uint32_t)((1ull << 32) / tpu) : 0xffffffffu;   // rd_kernels.h: split()
    const uint32_t nwaves = blocks * RD_WAVES;
    const uint32_t ntiles = (unit1 - unit0) * tpu;
    // ticket counters: groups of RD_TQ_CLIENTS waves need gridDim % 16 == 0 (rd_blocks_for rounds to that); else one counter
    static const bool static_deal = rd_env_u32("RD_STATIC_DEAL", 0) != 0;                 // A/B switch: no tickets
    c.tq_k = static_deal ? 0u : (blocks >= 16u && blocks % 16u == 0u) ? blocks / 4u : 1u;
    const uint32_t ndyn = ntiles > nwaves ? ntiles - nwaves : 0u;
    c.tq_tmax = c.tq_k ? (ndyn + c.tq_k - 1u) / c.tq_k : 0u;
    static const int burst_env = getenv("RD_BURST") ? atoi(getenv("RD_BURST")) : -1;   // A/B override: 0 / 1
    // read burst (rd_kernels.h): f32 surface by default; needs 16-byte aligned CFA rows and a launch worth it
    const bool burst_ok = ((uintptr_t)cfa % 16u) == 0 && (uint64_t)(unit1 - unit0) * W >= (1u << 19);
    const bool burst = burst_ok && FMT == RD_FMT_RGBA_F32 && (burst_env < 0 || burst_env != 0);
    c.blocks = blocks; c.cfa = cfa; c.out = out; c.W = W; c.H = H; c.unit0 = unit0; c.unit1 = unit1; c.tq = tq;
    c.slab32 = slab32; c.slab64 = slab64;
    // whole tiles for every even width from 128 up: a width that is not a multiple of 128 overlaps its last tile
    // (rd_kernels.h, TILES); the masked instance is left for frames narrower than one tile
    const int tiles = rd_tiles_mode(W, FMT);
    c.fn = tiles == RD_TILES_WHOLE ? (const void *)rd_develop_quads<FMT, HIST, RD_TILES_WHOLE, MATH, false>
         : tiles == RD_TILES_OVERLAP ? (const void *)rd_develop_quads<FMT, HIST, RD_TILES_OVERLAP, MATH, false>
                                     : (const void *)rd_develop_quads<FMT, HIST, RD_TILES_MASKED, MATH, false>;
    if constexpr (FMT == RD_FMT_RGBA_F32) {     // the burst variant and the odd-width tiling exist for the f32 surface only
        if (tiles == RD_TILES_SHIFT) c.fn = burst ? (const void *)rd_develop_quads<FMT, HIST, RD_TILES_SHIFT, MATH, true>
                                                  : (const void *)rd_develop_quads<FMT, HIST, RD_TILES_SHIFT, MATH, false>;
        if (burst && tiles == RD_TILES_WHOLE) c.fn = (const void *)rd_develop_quads<FMT, HIST, RD_TILES_WHOLE, MATH, true>;
        if (burst && tiles == RD_TILES_OVERLAP) c.fn = (const void *)rd_develop_quads<FMT, HIST, RD_TILES_OVERLAP, MATH, true>;
    }
    c.bind();
    if (!record) {
        // (W == 1: no quad, nothing for the export kernel to do -- the column below is the whole frame; a single-frame
        //  histogram's slab rows, which the export kernel would have overwritten, are cleared instead)
        if (W >> 1) (void)hipLaunchKernel(c.fn, dim3(blocks), dim3(RD_BLOCK), c.argv, 0, s);
        else if (HIST && slab32) (void)hipMemsetAsync(slab32, 0, (size_t)blocks * 768u * sizeof(uint32_t), s);
        if (W & 1u) {
            // odd width: the quads above cover columns [0, W - 1); the last column of the band's rows follows on the same
            // stream (rd_develop_lastcol), its histogram counts added to the slab rows just written
            const uint32_t row0 = unit0 ? 2u * unit0 - 1u : 0u;
            const uint32_t row1 = 2u * (unit1 - 1u) + 1u < H ? 2u * (unit1 - 1u) + 1u : H;
            if (row1 > row0) {
                const uint32_t need = (row1 - row0 + 255u) / 256u;
                hipLaunchKernelGGL((rd_develop_lastcol<FMT, HIST, MATH>), dim3(need < blocks ? need : blocks), dim3(256), 0, s, cfa, out,
                                   (const rd_frame_desc *)nullptr, 1u, W, H, row0, row1, c.u, slab32, slab64);
            }
        }
    }
}

// Does a multi-frame launch of W x H frames take the read-burst instance (f32 surface)?  The pattern probe
// (rd_batch_probe_pattern) is built for that instance only.
static bool rd_burst_launchable(uint32_t W, uint32_t H, bool aligned16)
{
    return aligned16 && W >= 128u && (uint64_t)(H / 2u + 1u) * W >= (1u << 19);      // (any parity: the sweep walks bytes)
}
static bool rd_probe_launchable(uint32_t W, uint32_t H, bool aligned16)
{
    static const int burst_env = getenv("RD_BURST") ? atoi(getenv("RD_BURST")) : -1;
    return rd_burst_launchable(W, H, aligned16) && (W % 4u) == 0 && burst_env != 0;      // (the diagnostic instances: no shifted windows)
}

// Multi-frame launch (rd_develop_batch): descs_dev[0 .. nframes-1] are whole frames of W x H.
// stamps != nullptr: the diagnostic instance that stamps its clocks per workgroup (rd_kernels.h, STAMP; f32 + histogram +
// strict arithmetic + read burst only: what rd_batch_measure_clock checks before it asks for it).
template <int FMT, bool HIST, int MATH>
static void rd_launch_batch_t(const rd_frame_desc *descs_dev, uint32_t nframes, uint32_t W, uint32_t H, uint32_t blocks,
                              bool burst_ok, unsigned long long *slab64, uint32_t *tq, hipStream_t s, uint32_t *stamps = nullptr)
{
    const uint32_t tpu = rd_tiles_per_unit(W, FMT);
    const uint32_t tpu_magic = tpu > 1u ? (uint32_t)((1ull << 32) / tpu) : 0xffffffffu;
    const uint32_t tpf = (H / 2u + 1u) * tpu;               // tiles per frame
    const uint32_t tpf_magic = tpf > 1u ? (uint32_t)((1ull << 32) / tpf) : 0xffffffffu;
    const uint32_t nwaves = blocks * RD_WAVES;
    const uint32_t ntiles = nframes * tpf;                  // < 2^32: rd_batch_develop sizes the launches
    static const bool static_deal = rd_env_u32("RD_STATIC_DEAL", 0) != 0;
    const uint32_t tq_k = static_deal ? 0u : (blocks >= 16u && blocks % 16u == 0u) ? blocks / 4u : 1u;
    const uint32_t ndyn = ntiles > nwaves ? ntiles - nwaves : 0u;
    const uint32_t tq_tmax = tq_k ? (ndyn + tq_k - 1u) / tq_k : 0u;
    static const int burst_env = getenv("RD_BURST") ? atoi(getenv("RD_BURST")) : -1;
    const bool burst = FMT == RD_FMT_RGBA_F32 && rd_burst_launchable(W, H, burst_ok) && (burst_env < 0 || burst_env != 0);
    const int tiles = rd_tiles_mode(W, FMT);
#define RD_LAUNCH_BATCH(TILES, BURST)                                                                                        \
    hipLaunchKernelGGL((rd_develop_batch<FMT, HIST, TILES, MATH, BURST>), dim3(blocks), dim3(RD_BLOCK), 0, s, descs_dev, nframes, \
                       W, H, tpu, tpu_magic, tpf, tpf_magic, tq_k, tq_tmax, tq, slab64, (uint32_t *)nullptr)
    // odd width: the frames' last column, all rows, behind whichever instance ran (rd_develop_lastcol)
    auto lastcol = [&]() {
        if constexpr (MATH != RD_MATH_PROBE) {
            if (!(W & 1u)) return;
            const uint64_t need = ((uint64_t)nframes * H + 255u) / 256u;
            hipLaunchKernelGGL((rd_develop_lastcol<FMT, HIST, MATH>), dim3((uint32_t)(need < blocks ? need : blocks)), dim3(256), 0, s,
                               (const uint16_t *)nullptr, (void *)nullptr, descs_dev, nframes, W, H, 0u, H, rd_ku{}, (uint32_t *)nullptr, slab64);
        }
    };
    if constexpr (FMT == RD_FMT_RGBA_F32 && HIST && MATH == RD_MATH_STRICT) {
        if (stamps && burst) {
#define RD_LAUNCH_STAMPED(TILES)                                                                                             \
    hipLaunchKernelGGL((rd_develop_batch<FMT, HIST, TILES, MATH, true, true>), dim3(blocks), dim3(RD_BLOCK), 0, s, descs_dev,    \
                       nframes, W, H, tpu, tpu_magic, tpf, tpf_magic, tq_k, tq_tmax, tq, slab64, stamps)
            if (tiles == RD_TILES_WHOLE) RD_LAUNCH_STAMPED(RD_TILES_WHOLE);
            else RD_LAUNCH_STAMPED(RD_TILES_OVERLAP);
#undef RD_LAUNCH_STAMPED
            lastcol();
            return;
        }
    }
    if constexpr (FMT == RD_FMT_RGBA_F32 && MATH != RD_MATH_PROBE) {
        if (tiles == RD_TILES_SHIFT) {                           // W % 4 != 0: 64-byte-aligned store windows (rd_kernels.h)
            if (burst) RD_LAUNCH_BATCH(RD_TILES_SHIFT, true);
            else RD_LAUNCH_BATCH(RD_TILES_SHIFT, false);
            lastcol();
            return;
        }
    }
    if constexpr (FMT == RD_FMT_RGBA_F32) {
        if (burst) {
            if (tiles == RD_TILES_WHOLE) RD_LAUNCH_BATCH(RD_TILES_WHOLE, true);
            else RD_LAUNCH_BATCH(RD_TILES_OVERLAP, true);
            lastcol();
            return;
        }
    }
    if constexpr (MATH == RD_MATH_PROBE) return;                 // (the probe has burst instances only: rd_probe_launchable)
    else {
        if (!(W >> 1)) { /* one pixel wide: no quad; the column kernel below is the whole frame */ }
        else if (tiles == RD_TILES_WHOLE) RD_LAUNCH_BATCH(RD_TILES_WHOLE, false);
        else if (tiles == RD_TILES_OVERLAP) RD_LAUNCH_BATCH(RD_TILES_OVERLAP, false);
        else RD_LAUNCH_BATCH(RD_TILES_MASKED, false);
        lastcol();
    }
#undef RD_LAUNCH_BATCH
}

template <int FMT, bool HIST, int MATH>
static void rd_launch_map_t(const uint16_t *cfa, void *out, uint32_t W, uint32_t H, uint32_t tw,
                            uint32_t th, uint32_t blocks, const rd_ku &u, uint32_t *slab32,
                            unsigned long long *slab64, hipStream_t s)
{
    hipLaunchKernelGGL((rd_develop_map<FMT, HIST, MATH>), dim3(blocks), dim3(RD_BLOCK), 0, s, cfa, out, W, H, tw,
                       th, u, slab32, slab64);
}

#define RD_DISPATCH3(fn, FMT, hist, math, ...)                                                    \
    do {                                                                                          \
        if (hist) {                                                                               \
            if (math == RD_MATH_CONTRACTED) fn<FMT, true, RD_MATH_CONTRACTED>(__VA_ARGS__);       \
            else fn<FMT, true, RD_MATH_STRICT>(__VA_ARGS__);                                      \
        } else {                                                                                  \
            if (math == RD_MATH_CONTRACTED) fn<FMT, false, RD_MATH_CONTRACTED>(__VA_ARGS__);      \
            else fn<FMT, false, RD_MATH_STRICT>(__VA_ARGS__);                                     \
        }                                                                                         \
    } while (0)
#define RD_DISPATCH(fn, fmt, hist, math, ...)                                                     \
    do {                                                                                          \
        if (fmt == RD_FMT_RGBA_F32) RD_DISPATCH3(fn, RD_FMT_RGBA_F32, hist, math, __VA_ARGS__);   \
        else if (fmt == RD_FMT_RGBA_F16) RD_DISPATCH3(fn, RD_FMT_RGBA_F16, hist, math, __VA_ARGS__); \
        else if (fmt == RD_FMT_RGB_U8) RD_DISPATCH3(fn, RD_FMT_RGB_U8, hist, math, __VA_ARGS__);  \
        else RD_DISPATCH3(fn, RD_FMT_RGBA_U8, hist, math, __VA_ARGS__);                           \
    } while (0)

static uint32_t rd_blocks_for(const rd_launch_cfg &cfg, uint64_t items, bool hist)
{
    uint64_t need = (items + RD_BLOCK - 1) / RD_BLOCK;
    uint64_t cap = (uint64_t)cfg.n_cu * (hist ? cfg.wg_per_cu_hist : cfg.wg_per_cu_plain);
    if (cap > RD_MAX_BLOCKS) cap = RD_MAX_BLOCKS;
    uint64_t b = need < cap ? need : cap;
    if (b >= 16u) b &= ~(uint64_t)15u;           // whole groups of ticket-counter clients (rd_launch_quads_t)
    return b ? (uint32_t)b : 1u;
}

static size_t rd_align_for(uint32_t fmt) { return fmt == RD_FMT_RGB_U8 ? 4 : 16; }   // vector width of the surface stores

// Enqueue one render of (cfa, W, H) to a tw x th target.  Returns the number of workgroups used
// (the slab rows written) through *blocks_out.  use_quads selects the export kernel.
static int rd_enqueue_render(const rd_launch_cfg &cfg, const uint16_t *cfa, uint32_t W, uint32_t H,
                             uint32_t tw, uint32_t th, uint32_t fmt, void *out, const rd_ku &u,
                             bool use_quads, uint32_t unit0, uint32_t unit1, bool hist, uint32_t math,
                             uint32_t *slab32, unsigned long long *slab64, uint32_t fixed_blocks,
                             uint32_t *tq, hipStream_t s, uint32_t *blocks_out, rd_quads_call *record = nullptr)
{
    uint32_t blocks;
    (void)hipGetLastError();                     // HIP's last-error slot is sticky per thread: what is read after the launch
                                                 // below must be THIS launch's, not an earlier failed call's
    if (use_quads) {
        if (!tq) return rd_fail(RD_ERR_OOM, "scheduler state allocation failed");
        const uint64_t items = (uint64_t)(unit1 - unit0) * rd_tiles_per_unit(W, fmt) * 64u;   // lanes
        if (items >= 0xffffffffull) return rd_fail(RD_ERR_UNSUPPORTED, "frame too large for 32-bit item index");
        blocks = fixed_blocks ? fixed_blocks : rd_blocks_for(cfg, items, hist);
        RD_DISPATCH(rd_launch_quads_t, fmt, hist, math, cfa, out, W, H, unit0, unit1, blocks, u, slab32, slab64, tq, s, record);
    } else {
        if (record) return rd_fail(RD_ERR_INVALID_ARG, "only the export kernel can be recorded");
        const uint64_t items = (uint64_t)tw * th;
        if (items >= 0xffffffffull) return rd_fail(RD_ERR_UNSUPPORTED, "target too large for 32-bit pixel index");
        blocks = fixed_blocks ? fixed_blocks : rd_blocks_for(cfg, items, hist);
        RD_DISPATCH(rd_launch_map_t, fmt, hist, math, cfa, out, W, H, tw, th, blocks, u, slab32, slab64, s);
    }
    RD_HIP(hipGetLastError());
    if (blocks_out) *blocks_out = blocks;
    return RD_OK;
}

#include "rd_host_pipeline.inl"
#include "rd_host_batch.inl"
#include "rd_host_diag.inl"

// ------------------------------------------------------------------------------------------------
// ingest helper: lossless-JPEG tiles of compressed DNGs (host code; rd_ljpeg.h)
// ------------------------------------------------------------------------------------------------
extern "C" int rd_ljpeg_decode(const uint8_t *src, size_t len, uint16_t *dst, size_t dst_capacity_samples, uint32_t *width,
                               uint32_t *height, uint32_t *components, uint32_t *precision) try
{
    RD_ENTRY(rd_ljpeg_decode);
    if (!src || (!dst && dst_capacity_samples)) return rd_fail(RD_ERR_INVALID_ARG, "rd_ljpeg_decode: NULL argument");
    if (width) *width = 0;
    if (height) *height = 0;
    if (components) *components = 0;
    if (precision) *precision = 0;
    const int rc = rd_ljpeg::decode(src, len, dst, dst_capacity_samples, width, height, components, precision);
    switch (rc) {
    case rd_ljpeg::OK: return RD_OK;
    case rd_ljpeg::ERR_UNSUPPORTED: return rd_fail(RD_ERR_UNSUPPORTED, "Failed to decode RAW: not a Huffman-coded lossless JPEG this decoder supports");
    case rd_ljpeg::ERR_SIZE: return rd_fail(RD_ERR_INVALID_ARG, "Failed to decode RAW: lossless-JPEG frame is larger than the destination");
    case rd_ljpeg::ERR_TRUNCATED: return rd_fail(RD_ERR_INVALID_ARG, "Failed to decode RAW: lossless-JPEG stream is truncated");
    default: return rd_fail(RD_ERR_INVALID_ARG, "Failed to decode RAW: malformed lossless-JPEG stream");
    }
}
RD_CATCH_INT(rd_ljpeg_decode)

// ------------------------------------------------------------------------------------------------
// plumbing
// ------------------------------------------------------------------------------------------------
extern "C" int rd_device_malloc(int device, size_t bytes, void **out) try
{
    RD_ENTRY(rd_device_malloc);
    if (!out) return rd_fail(RD_ERR_INVALID_ARG, "out is NULL");
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipMalloc(out, bytes ? bytes : 1));
    return RD_OK;
}
RD_CATCH_INT(rd_device_malloc)

extern "C" int rd_device_memory(int device, size_t *free_bytes, size_t *total_bytes) try
{
    RD_ENTRY(rd_device_memory);
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    size_t f = 0, t = 0;
    RD_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return RD_OK;
}
RD_CATCH_INT(rd_device_memory)

extern "C" int rd_device_free(int device, void *ptr) try
{
    RD_ENTRY(rd_device_free);
    if (!ptr) return RD_OK;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipFree(ptr));
    return RD_OK;
}
RD_CATCH_INT(rd_device_free)

extern "C" int rd_memcpy_h2d(int device, void *dst_dev, const void *src, size_t bytes) try
{
    RD_ENTRY(rd_memcpy_h2d);
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipMemcpy(dst_dev, src, bytes, hipMemcpyHostToDevice));
    return RD_OK;
}
RD_CATCH_INT(rd_memcpy_h2d)

extern "C" int rd_memcpy_d2h(int device, void *dst, const void *src_dev, size_t bytes) try
{
    RD_ENTRY(rd_memcpy_d2h);
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost));
    return RD_OK;
}
RD_CATCH_INT(rd_memcpy_d2h)

extern "C" int rd_stream_create(int device, void **out) try
{
    RD_ENTRY(rd_stream_create);
    if (!out) return rd_fail(RD_ERR_INVALID_ARG, "out is NULL");
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    hipStream_t s = nullptr;
    RD_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (void *)s;
    return RD_OK;
}
RD_CATCH_INT(rd_stream_create)

extern "C" int rd_stream_synchronize(int device, void *stream) try
{
    RD_ENTRY(rd_stream_synchronize);
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipStreamSynchronize((hipStream_t)stream));
    return RD_OK;
}
RD_CATCH_INT(rd_stream_synchronize)

extern "C" int rd_stream_destroy(int device, void *stream) try
{
    if (!stream) return RD_OK;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipStreamDestroy((hipStream_t)stream));
    return RD_OK;
}
RD_CATCH_INT(rd_stream_destroy)

// Test hooks (declared in rawdev.h under "test hooks"): never needed by a host.
extern "C" int rd_debug_poison_scheduler(rd_pipeline *p, void *stream) try
{
    RD_ENTRY(rd_debug_poison_scheduler);
    if (!p) return rd_fail(RD_ERR_INVALID_ARG, "NULL pipeline");
    rd_devguard g(p->device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", p->device);
    std::lock_guard<std::mutex> lk(p->mu);
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    if (!p->scratch.poison(s)) return rd_fail(RD_ERR_INVALID_ARG, "this pipeline holds no scheduler state for that stream yet");
    return RD_OK;
}
RD_CATCH_INT(rd_debug_poison_scheduler)

extern "C" uint32_t rd_debug_scheduler_entries(rd_pipeline *p) try { RD_ENTRY(rd_debug_scheduler_entries); return p ? (uint32_t)p->scratch.size() : 0u; } RD_CATCH_VAL(rd_debug_scheduler_entries, 0)

// render lanes this pipeline has created so far (<= RD_LANES_MAX), and whether a host range would take the direct-DMA path
extern "C" uint32_t rd_debug_lane_count(rd_pipeline *p) try
{
    RD_ENTRY(rd_debug_lane_count);
    if (!p) return 0u;
    std::lock_guard<std::mutex> lk(p->lane_mu);
    return (uint32_t)p->lanes.size();
}
RD_CATCH_VAL(rd_debug_lane_count, 0)

extern "C" int rd_debug_is_pinned_host(const void *ptr, size_t len) try { RD_ENTRY(rd_debug_is_pinned_host); return ptr && rd_is_pinned_host(ptr, len) ? 1 : 0; } RD_CATCH_INT(rd_debug_is_pinned_host)

// Arm (kind != 0) or disarm (site NULL / empty, or kind 0) the one-shot injected fault described at the top of this file.
extern "C" int rd_debug_inject_fault(const char *site, uint32_t kind, uint32_t after) try
{
    if (kind > RD_FAULT_FOREIGN) return rd_fail(RD_ERR_INVALID_ARG, "unknown fault kind %u", kind);
    if (site && strlen(site) >= 64) return rd_fail(RD_ERR_INVALID_ARG, "site name too long");
    rd_fault_arm(site, kind, after);
    return RD_OK;
}
RD_CATCH_INT(rd_debug_inject_fault)

extern "C" int rd_device_synchronize(int device) try
{
    RD_ENTRY(rd_device_synchronize);
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    RD_HIP(hipDeviceSynchronize());
    return RD_OK;
}
RD_CATCH_INT(rd_device_synchronize)
