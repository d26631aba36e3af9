/*
 * rawdev.h -- C ABI of librawdev.so: the MI355X (gfx950) drop-in for RawEditor's GPU develop path.
 *
 * The library replaces exactly one thing in the reference: `gpu::RenderPipeline`
 * (/root/reference/src/gpu/pipeline.rs:81-100, :112-737; re-exported gpu/mod.rs:16) together with
 * the WGSL shader it drives (/root/reference/src/gpu/shaders.rs:14-268).  Every entry point below
 * names the reference interface it stands in for.  A Rust host binds these with a plain
 * `extern "C"` block; nothing in the signatures is a torch, HIP or C++ type --
 * device pointers and streams cross the boundary as `void*` / `const uint16_t*`.
 *
 * Conventions
 *   - every function returns an rd_status (0 = ok, negative = error) unless noted; the message of
 *     the last error on the calling thread is rd_last_error().  Nothing aborts or throws: no C++ exception leaves an
 *     entry point (std::bad_alloc -> RD_ERR_OOM, anything else -> RD_ERR_INTERNAL; the handle stays valid).
 *   - host buffers are caller-owned and tightly packed (the reference strips wgpu's 256-byte row
 *     padding, pipeline.rs:511-521, :595-601); `*_len` arguments are byte lengths and are checked.
 *   - a pipeline handle may be used from several threads at once (the reference shares
 *     Arc<RenderPipeline> between the UI thread and a tokio blocking thread, main.rs:1054, :1749).
 *   - there is NO CPU fallback: without a usable gfx950 device every compute call fails with
 *     RD_ERR_NO_DEVICE / RD_ERR_HIP.
 */
#ifndef RAWDEV_H
#define RAWDEV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: round 1.  2: rd_frame gained `matrix_layout` (the struct grew by one u32); rd_batch_develop packs several
 * frames into one launch; new entry points rd_node_batch_*, rd_batch_plan_launches, rd_batch_last_launch_count,
 * rd_pipeline_set_matrix_layout, rd_selftest_q8 / _f16 (+ _codes / _halves), rd_ljpeg_decode, rd_stream_*, rd_debug_*.
 * 3: no signature changed.  Host renders no longer serialise on the pipeline (each call takes a render lane; the mutex only
 * guards the uniforms) and a full-resolution render is band-pipelined (see rd_render_full_res_to_bytes); new entry points
 * rd_host_alloc / rd_host_free (page-locked render destinations), rd_measure_hbm (the box's own streaming ceilings),
 * rd_render_full_res_borrow / rd_surface_release (a lent page-locked surface), rd_measure_valu, rd_device_identity, rd_selftest_q8_lut (+ _codes), rd_q8_lut_table,
 * rd_exporter_submit_host (the export ring fed from host memory), rd_selftest_f16_lut (+ _values), rd_f16_lut_tables.
 * 4: no signature changed.  New status RD_ERR_INTERNAL: a C++ exception inside the library (a failed host allocation is
 * RD_ERR_OOM) stops at the C boundary and comes back as a status -- every entry point is a function-try-block; new test hook
 * rd_debug_inject_fault.  RGB8 export (rd_batch_create, rd_exporter_create, rd_render) takes any even width from 128 up,
 * and every even width >= 128 runs the export kernel's whole-tile instances (it was W % 128 == 0).  rd_node_batch_* keeps
 * one worker thread per device for the life of the handle instead of starting N threads per call; new entry points
 * rd_node_batch_histogram_enqueue / _fetch (the global histogram without draining the devices).
 * 5: no signature changed.  rd_node_batch_histogram_fetch returns the SUM of every interval enqueued since the last fetch
 * (ABI 4 returned the last one and dropped the others); rd_batch_create / rd_exporter_create take odd widths (ABI 4:
 * RD_ERR_UNSUPPORTED) and the pipeline renders them with the export kernel too; new measurement aids
 * rd_batch_set_launch_timing, rd_batch_launch_timeline, rd_batch_probe_pattern, rd_batch_measure_clock. */
#define RD_ABI_VERSION 5

typedef enum rd_status {
    RD_OK = 0,
    RD_ERR_INVALID_ARG = -1,
    RD_ERR_NO_DEVICE = -2,
    RD_ERR_HIP = -3,
    RD_ERR_OOM = -4,         /* device memory, page-locked memory, or a host allocation (std::bad_alloc) */
    RD_ERR_UNSUPPORTED = -5,
    RD_ERR_INTERNAL = -6     /* a C++ exception other than bad_alloc was stopped at the C boundary; rd_last_error() has what() */
} rd_status;

/* Output surface formats.  U8 is what the reference renders (Rgba8Unorm, pipeline.rs:322, :454,
 * :538, :627); F32 is the shader's own return value vec4(color,1.0) (shaders.rs:264-266), the
 * surface BASELINE.json's north_star asks for; F16 is the 100 MP configuration's surface. */
typedef enum rd_format {
    RD_FMT_RGBA_F32 = 0, /* 16 B/px */
    RD_FMT_RGBA_F16 = 1, /*  8 B/px, IEEE binary16, round-to-nearest-even of the f32 value */
    RD_FMT_RGBA_U8 = 2,  /*  4 B/px, trunc(x*255 + 0.5), alpha 255 */
    RD_FMT_RGB_U8 = 3    /*  3 B/px, the RGBA8 surface with alpha dropped: what the reference feeds its JPEG
                             encoder (main.rs:1777-1786 strips alpha on the CPU) */
} rd_format;

/* Arithmetic of the colour stack.  Both are restatements of the same WGSL text
 * within the latitude WGSL leaves (contraction unspecified, division 2.5 ULP) and each is checked bit
 * for bit against the oracle in the same mode.
 *   STRICT     (default) literal operation order, no contraction, IEEE-correct division;
 *   CONTRACTED every a*b+c becomes one fma and x/d (d uniform) becomes x*RN(1/d) -- what an AMD shader
 *              compiler emits for the reference's shader; ~20 % fewer VALU instructions. */
typedef enum rd_math_mode { RD_MATH_STRICT = 0, RD_MATH_CONTRACTED = 1 } rd_math_mode;

/* How `color_matrix` (the host's row-major [9]) is applied (its section 8b's rd_options.matrix_layout).
 *   REFERENCE (default) the reference's behaviour: the shader builds mat3x3(row0, row1, row2), WGSL constructors take
 *             COLUMNS, so the rows are consumed as columns and out = M^T * c (shaders.rs:209-214).  A no-op for the
 *             identity matrix, which is all the reference app ever passes (color.rs:35-47).
 *   ROW_MAJOR the "intended" form out = M * c, for hosts that pass a real camera matrix. */
typedef enum rd_matrix_layout { RD_MATRIX_REFERENCE = 0, RD_MATRIX_ROW_MAJOR = 1 } rd_matrix_layout;

/* state::edit::EditParams (src/state/edit.rs:15-77): ten f32 in this order; #[repr(C)]-compatible. */
typedef struct rd_edit_params {
    float exposure;    /* stops, UI range [-5, 5]          (main.rs:1624-1660 for all ranges) */
    float contrast;    /* UI [-10, 10]; shader divides by 100 (shaders.rs:233) */
    float highlights;  /* UI [-1, 1] */
    float shadows;     /* UI [-1, 1] */
    float whites;      /* UI [0.8, 1.2], default 1.0 */
    float blacks;      /* UI [0, 0.2] */
    float vibrance;    /* UI [-1, 1] */
    float saturation;  /* UI [-100, 100]; shader divides by 100 (shaders.rs:245) */
    float temperature; /* UI [-1, 1] */
    float tint;        /* UI [-1, 1] */
} rd_edit_params;

/* The public fields of RenderPipeline (pipeline.rs:89-96) + dimensions() (:609). */
typedef struct rd_info {
    uint32_t width, height;
    uint32_t preview_width, preview_height;     /* pipeline.rs:125-128 */
    uint32_t histogram_width, histogram_height; /* pipeline.rs:131-133 */
    int64_t image_id;
} rd_info;

typedef struct rd_pipeline rd_pipeline;
typedef struct rd_batch rd_batch;

/* ---- library ------------------------------------------------------------------------------- */
int rd_abi_version(void);              /* returns RD_ABI_VERSION (not a status) */
const char *rd_last_error(void);       /* thread-local, never NULL */
int rd_device_count(int *count);       /* number of visible HIP devices */
/* PCI bus id ("0000:c1:00.0") and name of visible device `device` (either buffer may be NULL): what tells the ranks of a
 * multi-GPU run apart (bench.py prints it per rank and refuses a run in which two ranks share a device). */
int rd_device_identity(int device, char *pci_bus_id, size_t pci_cap, char *name, size_t name_cap);
/* EditParams::default() (edit.rs:81-95): all 0 except whites = 1. */
void rd_edit_params_default(rd_edit_params *p);
/* preview / histogram target sizes with the reference's truncating f32 arithmetic (pipeline.rs:125-133). */
int rd_derived_dims(uint32_t width, uint32_t height, uint32_t *preview_w, uint32_t *preview_h,
                    uint32_t *hist_w, uint32_t *hist_h);
size_t rd_format_bytes_per_pixel(uint32_t format); /* 0 for an unknown format */
/* Diagnostic (no device needed): the steps of the colour stack the export kernel will skip for these uniforms because
 * they are exact identities (bit mask, 1 = temperature/tint, 2 = identity matrix, 4 = exposure, 8 = highlights,
 * 16 = shadows, 32 = saturation, 64 = vibrance, 128 = divide fix-up, 256 = blacks).  Results never
 * depend on it. */
uint32_t rd_elided_steps(const rd_edit_params *params, const float wb_multipliers[4], const float color_matrix[9],
                         uint32_t math_mode);

/* ---- RenderPipeline ------------------------------------------------------------------------ */
/* RenderPipeline::new (pipeline.rs:114-363).  `cfa` is w*h u16, row-major, no padding
 * (raw/loader.rs:11-19); it is borrowed for the call and copied to HBM.  `color_matrix` is the
 * host's row-major [9]; like the reference the rows are consumed as COLUMNS (shaders.rs:209-214).
 * The first pipeline a process creates on a device also builds and uploads the narrow surfaces' code tables
 * (about 40 ms of host time, once): no later render allocates or synchronises on their account. */
int rd_pipeline_create(int device, int64_t image_id, const uint16_t *cfa, uint32_t width,
                       uint32_t height, const rd_edit_params *params, const float wb_multipliers[4],
                       const float color_matrix[9], rd_pipeline **out);
/* Same, but `cfa_dev` already lives in this device's HBM and is borrowed for the pipeline's
 * lifetime (batch export keeps frames resident; no copy). */
int rd_pipeline_create_from_device(int device, int64_t image_id, const uint16_t *cfa_dev,
                                   uint32_t width, uint32_t height, const rd_edit_params *params,
                                   const float wb_multipliers[4], const float color_matrix[9],
                                   rd_pipeline **out);
/* Drop (Arc count -> 0, main.rs:1028).  NULL is a no-op. */
void rd_pipeline_destroy(rd_pipeline *p);
int rd_pipeline_info(const rd_pipeline *p, rd_info *out);
/* Extension: integer black level subtracted (saturating) before normalisation.
 * 0 (default) is the reference, which subtracts nothing (shaders.rs:106-110). */
int rd_pipeline_set_black_level(rd_pipeline *p, uint32_t black_level);
/* Extension: select the arithmetic (rd_math_mode); RD_MATH_STRICT is the default. */
int rd_pipeline_set_math_mode(rd_pipeline *p, uint32_t math_mode);
/* Extension: how color_matrix is applied (rd_matrix_layout); RD_MATRIX_REFERENCE (rows as columns) is the default. */
int rd_pipeline_set_matrix_layout(rd_pipeline *p, uint32_t matrix_layout);

/* update_uniforms (pipeline.rs:367) == update_uniforms_with_zoom(params, 1, 0, 0). */
int rd_update_uniforms(rd_pipeline *p, const rd_edit_params *params);
/* update_uniforms_with_zoom (pipeline.rs:373-398). */
int rd_update_uniforms_with_zoom(rd_pipeline *p, const rd_edit_params *params, float zoom,
                                 float pan_x, float pan_y);

/* render_to_bytes (pipeline.rs:442-522): preview_width x preview_height RGBA8 with the current
 * uniforms (zoom/pan included). dst_len must be preview_w*preview_h*4. */
int rd_render_to_bytes(rd_pipeline *p, uint8_t *dst, size_t dst_len);
/* render_full_res_to_bytes (pipeline.rs:526-606; the export path, main.rs:1749-1754): width x height RGBA8, current
 * uniforms.  The reference renders, copies the texture to a MAP_READ buffer, blocks in poll(Wait) and de-pads 96.6 MB
 * row by row ("SLOW (1-2 seconds for 24MP)", pipeline.rs:525).  Here the frame is launched as row bands whose bytes
 * cross PCIe while the later bands are still computed.  `dst` may be any host memory:
 *   - page-locked (rd_host_alloc, hipHostMalloc, hipHostRegister; detected per call): the DMA engine writes it directly --
 *     the call costs the PCIe transfer (about 1.8 ms for 24 MP) and nothing else;
 *   - pageable (a Vec<u8>, malloc): the bytes pass through a small pinned staging ring and are moved into `dst` by a few
 *     helper threads (RD_COPY_THREADS, default 4) while the next chunk is in flight; first-touch page faults of a fresh
 *     buffer are the caller's and dominate (INTEGRATION.md section 2 has the numbers).
 * The call does NOT hold the pipeline's lock while it runs: a concurrent rd_render_to_bytes / rd_update_uniforms from the
 * UI thread proceeds on another render lane (the uniforms are snapshotted when the call starts -- like the reference, the
 * export uses whatever view() wrote last, main.rs:1515 vs :1754). */
int rd_render_full_res_to_bytes(rd_pipeline *p, uint8_t *dst, size_t dst_len);
/* The same render without an allocation on the caller's side: the width x height RGBA8 surface is rendered into page-locked
 * memory the pipeline owns (allocated on first use, reused afterwards) and LENT to the caller: *data stays valid until
 * rd_surface_release(p, *data).  This is what export_image_async needs -- a byte slice for image::save_buffer
 * (main.rs:1765-1791) -- at the price of the PCIe transfer, with no page faults and no host copy.  Up to four surfaces may be
 * out at a time; they are freed with the pipeline (release them first). */
int rd_render_full_res_borrow(rd_pipeline *p, const uint8_t **data, size_t *len);
int rd_surface_release(rd_pipeline *p, const uint8_t *data);
/* render_to_histogram_bytes (pipeline.rs:615-716): histogram_width x histogram_height RGBA8. */
int rd_render_to_histogram_bytes(rd_pipeline *p, uint8_t *dst, size_t dst_len);
/* calculate_histogram (pipeline.rs:720-736): [R[256], G[256], B[256]] counts of RGBA8 bytes
 * (alpha ignored), computed on the pipeline's device.  rgba_len is a byte length (multiple of 4). */
int rd_calculate_histogram(rd_pipeline *p, const uint8_t *rgba, size_t rgba_len, uint32_t hist[768]);

/* General form of the three renders: any target size and surface format, current uniforms.
 * `hist` (nullable) receives the fused 3x256 histogram of the 8-bit quantised output -- the same
 * counts rd_calculate_histogram would give on the U8 surface of this render.  Up to four host renders of one pipeline
 * run at the same time (one render lane each); a fifth waits for a lane.  Whole-frame targets of 16 MiB and more take the
 * band-pipelined read-back described at rd_render_full_res_to_bytes. */
int rd_render(rd_pipeline *p, uint32_t out_w, uint32_t out_h, uint32_t format, void *dst,
              size_t dst_len, uint32_t hist[768]);
/* Device-resident variant: `dst_dev` (and nullable `hist_dev`, 768 x u32) are device pointers on
 * the pipeline's device; work is enqueued on `stream` (a hipStream_t, NULL = default stream) and
 * NOT synchronised.  Renders on different streams may overlap freely, with or without a histogram: the pipeline keeps
 * its tile-scheduler counters and its histogram slab per stream (at most 16 streams at a time; beyond that the least
 * recently used stream's state is reused once its work has finished).  The first full-resolution render on a stream
 * allocates that state (synchronous; do it before capturing the stream into a graph). */
int rd_render_device(rd_pipeline *p, uint32_t out_w, uint32_t out_h, uint32_t format,
                     void *dst_dev, uint32_t *hist_dev, void *stream);

/* ---- batch export (no reference counterpart; BASELINE.json configs 3-5) -------------------- */
/* One frame of a batch: everything RenderPipeline::new + update_uniforms would be given. */
typedef struct rd_frame {
    const uint16_t *cfa_dev; /* w*h u16 in this device's HBM */
    void *out_dev;           /* w*h*bytes_per_pixel(format) in this device's HBM */
    rd_edit_params params;
    float wb_multipliers[4];
    float color_matrix[9];
    uint32_t black_level;
    uint32_t matrix_layout; /* rd_matrix_layout; 0 = the reference's (rows consumed as columns) */
} rd_frame;

/* A batch context owns the histogram slab for frames of one size/format on one device.  Use one context per
 * stream and per thread (the accumulator is private to the context; calls on one context are not re-entrant).
 * Frame sizes: any height, any width (ABI 5: an odd width is taken too -- the export kernel develops the whole 2 x 2 blocks of
 * a row pair and a small second kernel the last column; through ABI 4 it was RD_ERR_UNSUPPORTED); for RD_FMT_RGB_U8 at least
 * 128 pixels.  Every even width from 128 up runs at the full rate, a multiple of 128 or not; an odd width costs 5-8 %.
 * (RGBA-f32 at a width that is not a multiple of 4 -- rows that do not start on 64-byte blocks -- takes a tiling of its own
 * with shifted store windows: 6-8 % instead of the 20-28 % the plain tiling would cost there.) */
int rd_batch_create(int device, uint32_t width, uint32_t height, uint32_t format,
                    uint32_t with_histogram, rd_batch **out);
void rd_batch_destroy(rd_batch *b);
int rd_batch_set_math_mode(rd_batch *b, uint32_t math_mode); /* rd_math_mode, default RD_MATH_STRICT */
/* Enqueue the fused demosaic+develop(+histogram) of `n_frames` frames on `stream`, full resolution, zoom 1 / pan 0
 * (the export map).  Histogram counts accumulate inside the context in u64.  Not synchronised.
 * Launches: by default ONE launch covers several consecutive frames of the call (up to 8 for the f32 surface, 32 for the
 * narrower ones; RD_BATCH_MAX_FRAMES; the
 * kernel's tile tickets, uniforms and surface pointers change frame inside the launch, so there is no drain / refill
 * between frames).  A launch never holds two frames whose surfaces overlap, so the surfaces of one call are always
 * written in call order where they alias (an output ring).  The frame array is copied before the call returns.
 * `row_bands` is config 5's "tiled multi-launch per frame": with multi-frame launches it needs no launches of its own
 * (the ticket front sweeps a frame in row order; a band is a range of tickets) and is ignored; with
 * RD_BATCH_PERSISTENT=0 in the environment every frame is `row_bands` separate row-band launches (0 or 1 = one launch
 * per frame), as in ABI version 1. */
int rd_batch_develop(rd_batch *b, const rd_frame *frames, size_t n_frames, uint32_t row_bands,
                     void *stream);
/* No device needed: how rd_batch_develop would cut `n_frames` frames of this size / format into multi-frame launches
 * (`max_frames` = 0: the default cap, 8 for the f32 surface and 32 otherwise).  Writes the frames per launch, in order,
 * into counts[0 .. counts_cap) (nullable) and returns the number of launches (or a negative rd_status).  The limits: a
 * launch holds no two frames whose surfaces overlap, at most 2^32 - 1 pixels when a histogram is fused (u32 bins per
 * workgroup), at most 2^32 - 2 tiles. */
int rd_batch_plan_launches(uint32_t width, uint32_t height, uint32_t format, uint32_t with_histogram,
                           const rd_frame *frames, size_t n_frames, uint32_t max_frames, uint32_t *counts,
                           size_t counts_cap);
/* Number of fused kernel launches the last rd_batch_develop call on this context enqueued (measurement aid). */
uint32_t rd_batch_last_launch_count(const rd_batch *b);
/* Measurement aids (bench.py's launch_us_by_position / kernel_ms_per_step / frac_of_box_pattern; a host never needs them).
 * rd_batch_set_launch_timing: keep a HIP event pair around every fused launch of the last `keep_calls` develop / probe
 * calls of this context (0 = off, the default; at most 64).  The pairs put two barrier packets between neighbouring
 * launches: a timed call is for looking at, not for quoting.
 * rd_batch_launch_timeline: after the caller has synchronised the stream -- the kept launches, oldest first: start and end
 * of each in microseconds since the first kept launch's start, and the call each belongs to (0 = the oldest kept call).
 * At most `cap` entries are written; *n_launches = how many are kept.  Any of the three arrays may be NULL.
 * rd_batch_probe_pattern: the launches rd_batch_develop would enqueue for these frames with the kernel's ARITHMETIC
 * REMOVED -- every load, sweep, tile ticket, LDS stage and store of the RGBA-f32 export kernel, on the caller's own planes
 * and surfaces, which receive the raw samples as floats (NOT a develop; the histogram is not touched).  Timed, it is what
 * the kernel's memory pattern alone costs on this box in these buffers.  RGBA-f32 contexts with multi-frame launches only,
 * frames the read-burst instance takes (width >= 128, 16-byte aligned planes); RD_ERR_UNSUPPORTED otherwise. */
int rd_batch_set_launch_timing(rd_batch *b, uint32_t keep_calls);
int rd_batch_launch_timeline(rd_batch *b, float *start_us, float *end_us, uint32_t *call_index, size_t cap,
                             uint32_t *n_launches);
int rd_batch_probe_pattern(rd_batch *b, const rd_frame *frames, size_t n_frames, void *stream);
/* rd_batch_measure_clock: the shader clock the part holds UNDER the export kernel (it is power-managed: a pure-VALU loop,
 * a copy and this kernel each get a different one, and devices differ).  One ordinary develop of these frames -- surfaces
 * written, histogram counted -- through the one kernel instance in which thread 0 of every workgroup stamps the cycle counter
 * and the 100 MHz real-time counter into a buffer nothing else reads; synchronises; reports cycles / ticks x 100 MHz over the
 * workgroups of the LAST launch (median / min / max) and the median workgroup's busy time.  RGBA-f32 contexts with
 * histogram, strict arithmetic and multi-frame launches only (the headline's instance); any output may be NULL. */
int rd_batch_measure_clock(rd_batch *b, const rd_frame *frames, size_t n_frames, void *stream, double *ghz_median,
                           double *ghz_min, double *ghz_max, double *workgroup_us_median);
/* Reduce the accumulated histogram into `hist_dev` (768 x u64 on the device: R[256] G[256] B[256])
 * and reset the accumulator.  Enqueued on `stream`; the multi-GPU sum is the caller's all-reduce. */
int rd_batch_histogram(rd_batch *b, uint64_t *hist_dev, void *stream);

/* ---- node-level batch (SURVEY.md section 8b "Batch", 8e; BASELINE.json configs[3]) ----------------------- */
/* The batch path over the GPUs of one node from ONE process -- what a Rust or C host calls; bench.py and the Python
 * mirror reach the same sharding with one process per GPU and torch.distributed (raweditor_amd/batch.py).
 * Frame i of a call belongs to devices[i mod n_devices]; its cfa_dev / out_dev must live in THAT device's HBM
 * (rd_device_malloc(device, ...)).  No pixel crosses xGMI.  One rd_batch, one stream and one host thread per device.
 * The only exchange is the global histogram: ncclAllReduce(768, ncclUint64, ncclSum) over RCCL (librccl.so is loaded on
 * first use, and only when n_devices > 1); with one device there is no communicator.
 * Environment: RD_NODE_REDUCE=host folds the per-device histograms on the host instead (and then accepts a device
 * listed twice: a rehearsal of N > 1 on a one-GPU box); RD_NODE_REDUCE=rccl builds a communicator even for one device.
 * One RCCL per process: a librccl the process has already mapped (a PyTorch host) is used as it is; otherwise
 * RAWDEV_RCCL_LIB names the file to load, otherwise librccl.so.1 from the loader path.  RD_NODE_REDUCE=standin (tests
 * only) loads exactly the file RAWDEV_RCCL_LIB names -- tests/cpp/rccl_standin.cpp -- and accepts a device listed twice. */
typedef struct rd_node_batch rd_node_batch;
int rd_node_batch_create(const int *devices, uint32_t n_devices, uint32_t width, uint32_t height, uint32_t format,
                         uint32_t with_histogram, rd_node_batch **out);
void rd_node_batch_destroy(rd_node_batch *nb);
int rd_node_batch_set_math_mode(rd_node_batch *nb, uint32_t math_mode);
/* The dealing rule, for hosts that place their buffers: index into devices[] that owns frame `frame_index`. */
uint32_t rd_node_batch_device_of(uint32_t n_devices, size_t frame_index);
/* Enqueue all frames (each device's share through rd_batch_develop on that device's stream).  Not synchronised. */
int rd_node_batch_develop(rd_node_batch *nb, const rd_frame *frames, size_t n_frames, uint32_t row_bands);
/* Global histogram of everything developed since the last call: per-device fold, all-reduce, copy to `hist`
 * (R[256] G[256] B[256], u64).  Returns when every device has finished; resets the accumulators. */
int rd_node_batch_histogram(rd_node_batch *nb, uint64_t hist[768]);
/* The same in two halves, for a host that develops call after call and must not drain its devices in between (round 5):
 * _enqueue puts the per-device fold, the all-reduce and the read-back (into a page-locked buffer the handle owns) on the
 * devices' streams and returns at once; _fetch waits for those read-backs only -- not for develop calls enqueued after them --
 * and hands out the sum.  Intervals add up (ABI 5): _fetch returns everything developed since the previous _fetch (or
 * rd_node_batch_histogram), however many _enqueue calls lie between -- no interval is discarded -- and fails with
 * RD_ERR_INVALID_ARG when nothing has been enqueued since then. */
int rd_node_batch_histogram_enqueue(rd_node_batch *nb);
int rd_node_batch_histogram_fetch(rd_node_batch *nb, uint64_t hist[768]);
int rd_node_batch_synchronize(rd_node_batch *nb);
/* Measurement aids (bench.py --host node): the hipStream_t devices[index]'s share is enqueued on (record events there),
 * the fused launches its share of the last develop call took, and how the histogram is reduced (0 = one device, no
 * exchange; 1 = RCCL all-reduce; 2 = host fold, RD_NODE_REDUCE=host). */
void *rd_node_batch_stream(rd_node_batch *nb, uint32_t index);
uint32_t rd_node_batch_last_launch_count(const rd_node_batch *nb, uint32_t index);
int rd_node_batch_reduce_kind(const rd_node_batch *nb);

/* ---- export feed (the step after the path) ----------------------------------- */
/* The GPU half of export_image_async (main.rs:1744-1799) for a stream of frames: render_full_res_to_bytes'
 * blocking map/copy (pipeline.rs:552-605, "1-2 s for 24 MP") becomes a ring of pinned host buffers filled by
 * asynchronous D2H copies on a second stream, so the copy of frame i overlaps the kernel of frame i+1.
 * `format` is normally RD_FMT_RGBA_U8 (PNG path, main.rs:1767-1775) or RD_FMT_RGB_U8 (JPEG path: the alpha strip
 * of main.rs:1777-1786 is fused into the kernel and a quarter of the PCIe bytes disappears).  Widths: as rd_batch_create. */
typedef struct rd_exporter rd_exporter;
int rd_exporter_create(int device, uint32_t width, uint32_t height, uint32_t format, uint32_t math_mode,
                       uint32_t n_slots, rd_exporter **out);
void rd_exporter_destroy(rd_exporter *e);
/* Enqueue one frame (frame->out_dev is ignored; zoom 1 / pan 0).  *slot receives the ring slot used.  Fails with
 * RD_ERR_INVALID_ARG if that slot still holds an un-released frame. */
int rd_exporter_submit(rd_exporter *e, const rd_frame *frame, uint32_t *slot);
/* The same for a frame whose CFA plane is still in HOST memory -- RawDataResult.data as raw/loader.rs:11-19 hands it over
 * (width*height u16, row-major; frame->cfa_dev is ignored): the plane is uploaded into the slot's own HBM copy on a third
 * stream, so with a ring of two or more the upload of frame i+1, the kernel of frame i+1 and the read-back of frame i
 * overlap (pipeline.rs:190-206's queue.write_texture + :526-606 for a stream of files).  A page-locked plane
 * (rd_host_alloc) is read by the DMA engine where it lies and must stay untouched until rd_exporter_wait(slot) returns;
 * any other memory is copied through page-locked staging before the call returns and may be reused at once. */
int rd_exporter_submit_host(rd_exporter *e, const rd_frame *frame, const uint16_t *cfa_host, uint32_t *slot);
/* Block until the slot's surface is in host memory.  *data (pinned, width*height*bpp bytes, tightly packed) stays
 * valid until rd_exporter_release(slot). */
int rd_exporter_wait(rd_exporter *e, uint32_t slot, const void **data, size_t *len);
int rd_exporter_release(rd_exporter *e, uint32_t slot);

/* ---- self-test ----------------------------------------------------------------------------- */
/* The RGBA8 / RGB8 surfaces (the reference's target format, pipeline.rs:322) compute their 8-bit codes with a shortcut
 * (hardware log2/exp2, the pinned evaluation only near a code boundary).  rd_selftest_q8 runs that
 * shortcut and the pinned trunc(255 * gamma(x) + 0.5) over ALL 2^32 float encodings on `device` (about a second) and
 * reports how many differ (must be 0; *first_bad = the smallest differing encoding or 0xffffffff), how many encodings
 * take the pinned evaluation, and the largest distance in codes between the two evaluations before truncation.
 * Any output pointer may be NULL. */
int rd_selftest_q8(int device, uint64_t *mismatches, uint32_t *first_bad, uint64_t *fallbacks, float *max_dist);
/* The shortcut's codes for the float encodings first_encoding .. first_encoding + n - 1 (n a multiple of 256), so a
 * host-side oracle can be compared with them. */
int rd_selftest_q8_codes(int device, uint32_t first_encoding, uint32_t n, uint8_t *dst);

/* Round 4: the export kernel takes the 8-bit codes of the RGBA8 / RGB8 surfaces from a threshold table in LDS (the code is a
 * monotone step function of x with at most one step per 2^16 float encodings).  rd_selftest_q8_lut
 * runs that table against the pinned trunc(255 * gamma(x) + 0.5) over ALL 2^32 float encodings on `device` (must report 0
 * mismatches); rd_selftest_q8_lut_codes returns its codes for a range of encodings (n a multiple of 256) for a host-side
 * oracle; rd_q8_lut_table (no device needed) returns the table itself: 3969 words, returns the count or a negative status. */
int rd_selftest_q8_lut(int device, uint64_t *mismatches, uint32_t *first_bad);
int rd_selftest_q8_lut_codes(int device, uint32_t first_encoding, uint32_t n, uint8_t *dst);
int rd_q8_lut_table(uint32_t *dst, size_t cap_words);

/* The same for the RGBA-f16 surface's shortcut: the binary16 value (and the 8-bit code for the fused histogram) of every
 * float encoding against binary16(pinned gamma) / the pinned code.  *fallbacks = encodings in binary16's normal range
 * (gamma >= 2^-14) where the pinned evaluation decides the half; below that range it always does. */
int rd_selftest_f16(int device, uint64_t *mismatches, uint32_t *first_bad, uint64_t *fallbacks);
int rd_selftest_f16_halves(int device, uint32_t first_encoding, uint32_t n, uint16_t *dst);
/* Round 4: the export kernel takes the RGBA-f16 surface's halves AND its histogram codes from two-level threshold tables in LDS
 * (binary16(gamma(x)) is a step function with at most one step per 2^13 float encodings for x >= 2^-16 and ONE non-monotone
 * encoding).  rd_selftest_f16_lut runs the lookup -- including the pinned evaluation of the lanes it
 * sends there: 0 < x < 2^-16 and that one encoding, counted in *pinned -- against binary16(pinned gamma) and the pinned code
 * over ALL 2^32 encodings (must report 0 mismatches); rd_selftest_f16_lut_values returns half | code << 16 for a range of
 * encodings; rd_f16_lut_tables (no device needed) returns the tables: 17410 u16 and 2 x 2177 words. */
int rd_selftest_f16_lut(int device, uint64_t *mismatches, uint32_t *first_bad, uint64_t *pinned);
int rd_selftest_f16_lut_values(int device, uint32_t first_encoding, uint32_t n, uint32_t *dst);
int rd_f16_lut_tables(uint16_t *fine, size_t cap_fine, uint32_t *coarse, size_t cap_coarse_words);

/* ---- ingest helper (the step before the path; no device needed) ------------------ */
/* Lossless JPEG (ITU-T T.81 SOF3: Huffman, predictors 1-7, 1-4 interleaved components, precision 2-16, restart
 * intervals) as found in the tiles / strips of compressed DNGs (TIFF Compression = 7).  The reference decodes RAW files
 * with the un-vendored `rawloader` crate (raw/loader.rs:50-54): no parity claim against it; lossless JPEG is exact by
 * construction.  dst receives height * width * components samples, row-major, components interleaved.
 * The frame's dimensions are written as soon as its headers have been walked -- also when the call then fails because
 * dst_capacity_samples is too small, so a first call with dst = NULL and capacity 0 sizes the buffer (a stream that
 * fails earlier leaves them 0).  A stream that ends before its declared frame fails at the row where it ran dry; a
 * restart interval that is not a whole number of lines is refused. */
int rd_ljpeg_decode(const uint8_t *src, size_t len, uint16_t *dst, size_t dst_capacity_samples, uint32_t *width,
                    uint32_t *height, uint32_t *components, uint32_t *precision);

/* ---- plumbing for hosts without a HIP binding (tests, the Python mirror) -------------------- */
/* Page-locked host memory (hipHostMalloc): a render destination the DMA engines write directly (see
 * rd_render_full_res_to_bytes).  rd_host_free(device, NULL) is a no-op. */
int rd_host_alloc(int device, size_t bytes, void **out);
int rd_host_free(int device, void *ptr);
/* Measurement aid (bench.py's roofline.box_*): what THIS device streams right now, with librawdev's own trivial kernels --
 * a float4 copy of `bytes` to another buffer (GB/s counts the bytes read plus the bytes written), a non-temporal float4
 * fill and a float4 read of `bytes` (a wave walks its own contiguous range, eight 1-KiB accesses in flight: the fastest
 * shape tools/hbm_probe.hip finds) and hipMemsetAsync beside them; each launched `reps` times on a private stream, the
 * median launch reported.  Any output may be NULL.  Allocates up to 2 x bytes of device memory for the call: `bytes` is
 * rounded DOWN to a multiple of 512 MiB (one 8-KiB step for each of the copy grid's 65 536 waves; the kernels have no tail
 * handling) and a size below 512 MiB is RD_ERR_INVALID_ARG. */
int rd_measure_hbm(int device, size_t bytes, uint32_t reps, double *copy_GBps, double *fill_GBps, double *read_GBps,
                   double *memset_GBps);
/* Measurement aid (bench.py's valu_issue_frac): nanoseconds one full-rate VALU wave-instruction (v_mul_f32 / v_add_f32, all
 * VGPR: the colour stack's staple) costs a SIMD of this device right now, eight waves per SIMD as the export kernel runs. */
int rd_measure_valu(int device, double *ns_per_full_rate_instruction);
int rd_device_malloc(int device, size_t bytes, void **out);
int rd_device_free(int device, void *ptr);
int rd_device_memory(int device, size_t *free_bytes, size_t *total_bytes);   /* hipMemGetInfo */
int rd_memcpy_h2d(int device, void *dst_dev, const void *src, size_t bytes);
int rd_memcpy_d2h(int device, void *dst, const void *src_dev, size_t bytes);
int rd_device_synchronize(int device);
int rd_stream_create(int device, void **stream_out);   /* a non-blocking hipStream_t on `device` */
int rd_stream_synchronize(int device, void *stream);
int rd_stream_destroy(int device, void *stream);

/* ---- test hooks (the -m gpu suite; a host never needs them) -------------------------------- */
/* Overwrites the tile-scheduler counters this pipeline keeps for `stream` (NULL = its own stream) with garbage and marks
 * them suspect, as the library does itself when a launch or a synchronisation on that stream fails; the next render on
 * the stream must re-zero them and come out right. */
int rd_debug_poison_scheduler(rd_pipeline *p, void *stream);
uint32_t rd_debug_scheduler_entries(rd_pipeline *p);   /* streams this pipeline currently keeps state for (<= 16) */
uint32_t rd_debug_lane_count(rd_pipeline *p);          /* render lanes this pipeline has created so far (<= 4) */
int rd_debug_is_pinned_host(const void *ptr, size_t len);   /* 1: a render into [ptr, ptr + len) takes the direct-DMA path */
/* devices[index]'s own 768 x u64 histogram buffer as the last rd_node_batch_histogram left it (after an all-reduce every
 * device holds the global sum). */
int rd_debug_node_histogram_of(rd_node_batch *nb, uint32_t index, uint64_t hist[768]);
/* Fault injection: proves that nothing thrown inside the library crosses this boundary.  Arms ONE fault: the (after + 1)-th
 * passage through a fault point named `site` throws once, then the fault disarms itself.  Fault points: every entry point
 * under its own name ("rd_batch_develop"), the real allocation / thread-start sites ("pipeline.lanes", "batch.descs",
 * "node.share", "node.thread", "scratch.entry", "exporter.slots"), "*" = whichever comes first.  site NULL or kind 0 disarms.
 * The same through the environment, read when the library is loaded: RD_FAULT_INJECT=site:kind[:after]. */
typedef enum rd_fault_kind {
    RD_FAULT_NONE = 0,
    RD_FAULT_BAD_ALLOC = 1,     /* std::bad_alloc            -> RD_ERR_OOM */
    RD_FAULT_THREAD_START = 2,  /* std::system_error(EAGAIN) -> RD_ERR_INTERNAL */
    RD_FAULT_RUNTIME = 3,       /* std::runtime_error        -> RD_ERR_INTERNAL */
    RD_FAULT_FOREIGN = 4        /* not a std::exception      -> RD_ERR_INTERNAL */
} rd_fault_kind;
int rd_debug_inject_fault(const char *site, uint32_t kind, uint32_t after);

#ifdef __cplusplus
}
#endif
#endif /* RAWDEV_H */
