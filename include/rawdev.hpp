// rawdev.hpp -- C++17 host mirror of the reference interface for the develop path, header-only,
// over the C ABI of rawdev.h.  The reference's host is Rust (no toolchain in this image), so the
// host side above the ABI is written in C++ with the reference's names, argument meaning and error
// behaviour:
//
//   rawdev::EditParams      <- state::edit::EditParams   (reference src/state/edit.rs:15-122)
//   rawdev::RenderPipeline  <- gpu::RenderPipeline       (reference src/gpu/pipeline.rs:81-100, :112-737)
//
// `Result<T, String>` becomes rawdev::Error (an exception carrying the rd_status and the message of
// rd_last_error()); `Vec<u8>` becomes std::vector<uint8_t>; `[[u32; 256]; 3]` becomes
// std::array<std::array<uint32_t, 256>, 3>.  Nothing here computes pixels: every render is a HIP
// kernel launch inside librawdev.so.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rawdev.h"

namespace rawdev {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc)
{
    if (rc != RD_OK) throw Error(rc, rd_last_error());
}

// state::edit::EditParams.  Field order = serde order = uniform-block order.
struct EditParams : rd_edit_params {
    EditParams() { rd_edit_params_default(this); }                       // Default (edit.rs:79-96)
    static EditParams new_() { return EditParams(); }                    // new() (edit.rs:100-102)

    static constexpr const char *kFields[10] = { "exposure", "contrast", "highlights", "shadows", "whites",
                                                 "blacks", "vibrance", "saturation", "temperature", "tint" };
    float *begin() { return &exposure; }
    const float *begin() const { return &exposure; }

    bool operator==(const EditParams &o) const { return std::memcmp(begin(), o.begin(), 10 * sizeof(float)) == 0; }
    bool is_unedited() const { return *this == EditParams(); }           // edit.rs:115-117
    void reset() { *this = EditParams(); }                               // edit.rs:120-122

    // The text serde_json writes for a finite f32: the `ryu` crate's format32.  Shortest decimal digits that round-trip
    // (digits x 10^k, kk = number of digits + k), laid out as
    //   0 <= k && kk <= 13 : digits, k zeros, ".0"      0 < kk <= 13 : point inside the digits
    //   -6 < kk <= 0       : "0.", -kk zeros, digits     otherwise    : d[.ddd]e<exp> (no '+', no padding): 1e-7, 1.5e-7, 1e13
    // with a sign for negatives including -0.0.
    static std::string ryu_f32(float v)
    {
        if (v == 0.0f) return std::signbit(v) ? "-0.0" : "0.0";
        const std::string sign = v < 0.0f ? "-" : "";
        const float a = std::fabs(v);
        std::string digits;
        int e10 = 0;
        for (int n = 1; n <= 9 && digits.empty(); ++n) {                 // the shortest n digits that give `a` back
            char buf[48];
            std::snprintf(buf, sizeof buf, "%.*e", n - 1, (double)a);     // correctly rounded to n digits: d.ddde[+-]xx
            unsigned long long m = 0;
            const char *p = buf;
            for (; *p && *p != 'e'; ++p)
                if (*p >= '0' && *p <= '9') m = m * 10 + (unsigned long long)(*p - '0');
            const int e = std::atoi(p + 1);
            // (the interval of decimals that read back as `a` is lopsided at powers of two: a neighbour of the correctly
            //  rounded n-digit decimal can lie inside it when that one does not)
            const long long tries[3] = { 0, -1, 1 };
            for (long long d : tries) {
                const long long mm = (long long)m + d;
                if (mm <= 0) continue;
                char cand[64];
                std::snprintf(cand, sizeof cand, "%llde%d", mm, e - (n - 1));
                if (std::strtof(cand, nullptr) == a) {
                    digits = std::to_string(mm);
                    e10 = e + ((int)digits.size() - n);                  // m + 1 may have gained a digit (999 -> 1000)
                    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
                    break;
                }
            }
        }
        const int n = (int)digits.size(), k = e10 - (n - 1), kk = e10 + 1;
        std::string body;
        if (0 <= k && kk <= 13) body = digits + std::string((size_t)k, '0') + ".0";
        else if (0 < kk && kk <= 13) body = digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
        else if (-6 < kk && kk <= 0) body = "0." + std::string((size_t)-kk, '0') + digits;
        else if (n == 1) body = digits + "e" + std::to_string(kk - 1);
        else body = digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
        return sign + body;
    }

    // to_json (edit.rs:105-107): {"exposure":0.0,...} in serde's field order, every f32 in ryu's text (byte-identical to
    // serde_json's output, exponent forms included); a non-finite value is `null`, as serde_json writes it.
    std::string to_json() const
    {
        std::string s = "{";
        for (int i = 0; i < 10; ++i) {
            const float v = begin()[i];
            s += std::string(i ? "," : "") + "\"" + kFields[i] + "\":" + (std::isfinite(v) ? ryu_f32(v) : std::string("null"));
        }
        return s + "}";
    }

    // from_json (edit.rs:110-112): every field required (serde derive), unknown fields ignored.
    static EditParams from_json(const std::string &text)
    {
        EditParams p;
        for (int i = 0; i < 10; ++i) {
            const std::string key = std::string("\"") + kFields[i] + "\"";
            size_t k = text.find(key);
            if (k == std::string::npos) throw Error(RD_ERR_INVALID_ARG, std::string("missing field `") + kFields[i] + "`");
            k = text.find(':', k + key.size());
            if (k == std::string::npos) throw Error(RD_ERR_INVALID_ARG, "malformed JSON");
            char *end = nullptr;
            const float v = std::strtof(text.c_str() + k + 1, &end);
            if (end == text.c_str() + k + 1) throw Error(RD_ERR_INVALID_ARG, std::string("invalid value for `") + kFields[i] + "`");
            p.begin()[i] = v;
        }
        return p;
    }
};

// color::calculate_cam_to_srgb_matrix (reference src/color.rs:35-47): the reference returns the identity for ANY
// input (the real maths is commented out); color::is_identity_matrix (:172-178): |m - I| < 0.001 element-wise.
inline std::array<float, 9> calculate_cam_to_srgb_matrix(const std::array<float, 9> & /*xyz_to_cam*/)
{
    return { 1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f };
}
inline bool is_identity_matrix(const std::array<float, 9> &m)
{
    const float id[9] = { 1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f };
    for (int i = 0; i < 9; ++i)
        if (!(std::fabs(m[i] - id[i]) < 0.001f)) return false;
    return true;
}

using Histogram = std::array<std::array<uint32_t, 256>, 3>;

// Page-locked host bytes (rd_host_alloc): a render destination the GPU's DMA engine writes directly -- what
// export_image_async (main.rs:1749-1754) would hold instead of a fresh Vec<u8> to make a 24 MP export cost the PCIe
// transfer and nothing else (include/rawdev.h, rd_render_full_res_to_bytes).  Move-only.
class PinnedBytes {
public:
    explicit PinnedBytes(size_t bytes, int device = 0) : device_(device), bytes_(bytes)
    {
        void *p = nullptr;
        check(rd_host_alloc(device, bytes, &p));
        p_ = static_cast<uint8_t *>(p);
    }
    PinnedBytes(PinnedBytes &&o) noexcept : device_(o.device_), bytes_(o.bytes_), p_(o.p_) { o.p_ = nullptr; }
    PinnedBytes(const PinnedBytes &) = delete;
    PinnedBytes &operator=(const PinnedBytes &) = delete;
    ~PinnedBytes() { if (p_) rd_host_free(device_, p_); }
    uint8_t *data() { return p_; }
    const uint8_t *data() const { return p_; }
    size_t size() const { return bytes_; }

private:
    int device_;
    size_t bytes_;
    uint8_t *p_ = nullptr;
};

// gpu::RenderPipeline.  Move-only owner of an rd_pipeline; share it across threads by reference or
// shared_ptr exactly like the reference shares Arc<RenderPipeline> (main.rs:1054, :1749).
class RenderPipeline {
public:
    // pub fields of the reference (pipeline.rs:89-96)
    uint32_t width = 0, height = 0, preview_width = 0, preview_height = 0;
    int64_t image_id = 0;
    uint32_t histogram_width = 0, histogram_height = 0;

    // RenderPipeline::new (pipeline.rs:114-122).  Throws Error where the reference returns Err(String).
    static RenderPipeline new_(int64_t image_id, const std::vector<uint16_t> &raw_data, uint32_t width,
                               uint32_t height, const EditParams &params, const std::array<float, 4> &wb_multipliers,
                               const std::array<float, 9> &color_matrix, int device = 0)
    {
        if (raw_data.size() != (size_t)width * height)
            throw Error(RD_ERR_INVALID_ARG, "raw_data length does not match width*height");
        RenderPipeline p;
        check(rd_pipeline_create(device, image_id, raw_data.data(), width, height, &params, wb_multipliers.data(),
                                 color_matrix.data(), &p.h_));
        rd_info info;
        check(rd_pipeline_info(p.h_, &info));
        p.width = info.width; p.height = info.height;
        p.preview_width = info.preview_width; p.preview_height = info.preview_height;
        p.histogram_width = info.histogram_width; p.histogram_height = info.histogram_height;
        p.image_id = info.image_id;
        return p;
    }

    RenderPipeline(RenderPipeline &&o) noexcept { *this = std::move(o); }
    RenderPipeline &operator=(RenderPipeline &&o) noexcept
    {
        if (this != &o) {
            rd_pipeline_destroy(h_);
            width = o.width; height = o.height; preview_width = o.preview_width; preview_height = o.preview_height;
            image_id = o.image_id; histogram_width = o.histogram_width; histogram_height = o.histogram_height;
            h_ = o.h_;
            o.h_ = nullptr;
        }
        return *this;
    }
    RenderPipeline(const RenderPipeline &) = delete;
    RenderPipeline &operator=(const RenderPipeline &) = delete;
    ~RenderPipeline() { rd_pipeline_destroy(h_); }

    void update_uniforms(const EditParams &p) const { check(rd_update_uniforms(h_, &p)); }                // :367
    void update_uniforms_with_zoom(const EditParams &p, float zoom, float pan_x, float pan_y) const      // :373
    {
        check(rd_update_uniforms_with_zoom(h_, &p, zoom, pan_x, pan_y));
    }
    std::vector<uint8_t> render_to_bytes() const                                                        // :442
    {
        std::vector<uint8_t> v((size_t)preview_width * preview_height * 4);
        check(rd_render_to_bytes(h_, v.data(), v.size()));
        return v;
    }
    std::vector<uint8_t> render_full_res_to_bytes() const                                               // :526
    {
        std::vector<uint8_t> v((size_t)width * height * 4);
        check(rd_render_full_res_to_bytes(h_, v.data(), v.size()));
        return v;
    }
    // The export without an allocation: a page-locked surface the pipeline owns, lent until the Surface goes out of scope.
    class Surface {
    public:
        Surface(rd_pipeline *p, const uint8_t *d, size_t n) : p_(p), d_(d), n_(n) {}
        Surface(Surface &&o) noexcept : p_(o.p_), d_(o.d_), n_(o.n_) { o.d_ = nullptr; }
        Surface(const Surface &) = delete;
        Surface &operator=(const Surface &) = delete;
        ~Surface() { if (d_) rd_surface_release(p_, d_); }
        const uint8_t *data() const { return d_; }
        size_t size() const { return n_; }

    private:
        rd_pipeline *p_;
        const uint8_t *d_;
        size_t n_;
    };
    Surface render_full_res() const
    {
        const uint8_t *d = nullptr;
        size_t n = 0;
        check(rd_render_full_res_borrow(h_, &d, &n));
        return Surface(h_, d, n);
    }
    // The same render into memory the caller keeps: a reused buffer (no first-touch page faults) or PinnedBytes (direct DMA).
    void render_full_res_into(uint8_t *dst, size_t len) const { check(rd_render_full_res_to_bytes(h_, dst, len)); }
    std::vector<uint8_t> render_to_histogram_bytes() const                                              // :615
    {
        std::vector<uint8_t> v((size_t)histogram_width * histogram_height * 4);
        check(rd_render_to_histogram_bytes(h_, v.data(), v.size()));
        return v;
    }
    Histogram calculate_histogram(const std::vector<uint8_t> &rgba_bytes) const                         // :720
    {
        Histogram h;
        check(rd_calculate_histogram(h_, rgba_bytes.data(), rgba_bytes.size() - rgba_bytes.size() % 4, &h[0][0]));
        return h;
    }
    std::pair<uint32_t, uint32_t> dimensions() const { return { width, height }; }                      // :609

    // extension: the f32 surface the north-star asks for (vec4(color,1.0), shaders.rs:264-266)
    std::vector<float> render_f32(uint32_t out_w, uint32_t out_h, Histogram *hist = nullptr) const
    {
        std::vector<float> v((size_t)out_w * out_h * 4);
        check(rd_render(h_, out_w, out_h, RD_FMT_RGBA_F32, v.data(), v.size() * sizeof(float), hist ? &(*hist)[0][0] : nullptr));
        return v;
    }
    rd_pipeline *handle() const { return h_; }

private:
    RenderPipeline() = default;
    rd_pipeline *h_ = nullptr;
};

// A buffer in one device's HBM (rd_device_malloc / rd_device_free), for hosts without a HIP binding.
class DeviceBuffer {
public:
    DeviceBuffer(int device, size_t bytes) : device_(device), bytes_(bytes) { check(rd_device_malloc(device, bytes, &p_)); }
    DeviceBuffer(int device, const void *src, size_t bytes) : DeviceBuffer(device, bytes) { check(rd_memcpy_h2d(device, p_, src, bytes)); }
    DeviceBuffer(DeviceBuffer &&o) noexcept : device_(o.device_), bytes_(o.bytes_), p_(o.p_) { o.p_ = nullptr; }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { if (p_) rd_device_free(device_, p_); }
    void *get() const { return p_; }
    size_t size() const { return bytes_; }
    void download(void *dst, size_t bytes) const { check(rd_memcpy_d2h(device_, dst, p_, bytes)); }

private:
    int device_;
    size_t bytes_;
    void *p_ = nullptr;
};

// Batch export over the GPUs of one node from one process (rd_node_batch_*; no reference counterpart -- the reference exports
// one frame per click, main.rs:1744-1799).  Frame i of a call belongs to devices[i mod N]; the global histogram is one RCCL
// all-reduce of 768 x u64.
class NodeBatch {
public:
    NodeBatch(const std::vector<int> &devices, uint32_t width, uint32_t height, rd_format format, bool with_histogram = true)
        : devices_(devices)
    {
        check(rd_node_batch_create(devices.data(), (uint32_t)devices.size(), width, height, format, with_histogram ? 1u : 0u, &h_));
    }
    NodeBatch(const NodeBatch &) = delete;
    NodeBatch &operator=(const NodeBatch &) = delete;
    ~NodeBatch() { rd_node_batch_destroy(h_); }
    int device_of(size_t frame_index) const { return devices_[rd_node_batch_device_of((uint32_t)devices_.size(), frame_index)]; }
    void develop(const std::vector<rd_frame> &frames, uint32_t row_bands = 1) { check(rd_node_batch_develop(h_, frames.data(), frames.size(), row_bands)); }
    std::array<std::array<uint64_t, 256>, 3> histogram()
    {
        std::array<std::array<uint64_t, 256>, 3> h;
        check(rd_node_batch_histogram(h_, &h[0][0]));
        return h;
    }
    // the two halves (no drain between develop calls): enqueue now, fetch the last enqueue's result later
    void histogram_enqueue() { check(rd_node_batch_histogram_enqueue(h_)); }
    std::array<std::array<uint64_t, 256>, 3> histogram_fetch()
    {
        std::array<std::array<uint64_t, 256>, 3> h;
        check(rd_node_batch_histogram_fetch(h_, &h[0][0]));
        return h;
    }
    void synchronize() { check(rd_node_batch_synchronize(h_)); }
    // measurement aids: the hipStream_t devices[index]'s share is enqueued on; fused launches of its share of the last call
    void *stream(uint32_t index) const { return rd_node_batch_stream(h_, index); }
    uint32_t last_launch_count(uint32_t index = 0) const { return rd_node_batch_last_launch_count(h_, index); }

private:
    std::vector<int> devices_;
    rd_node_batch *h_ = nullptr;
};

}  // namespace rawdev
