// C++ host-mirror test (include/rawdev.hpp over the C ABI).  The EditParams cases restate the
// reference's own unit tests (src/state/edit.rs:129-163); the RenderPipeline cases drive the
// reference-shaped surface and compare with the CPU oracle (oracle/develop_ref.h -- the checker).
//   g++ -std=c++17 -Iinclude -Ioracle tests/cpp/test_host_mirror.cpp -o /tmp/t \
//       raweditor_amd/librawdev.so oracle/libdevelop_ref.so -Wl,-rpath,... && /tmp/t [--gpu]
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include "develop_ref.h"
#include "rawdev.hpp"

static int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static void test_default_is_unedited()            // edit.rs:129-133
{
    rawdev::EditParams p;
    EXPECT(p.is_unedited());
    EXPECT(p.whites == 1.0f && p.exposure == 0.0f && p.tint == 0.0f);
}

static void test_serialization()                  // edit.rs:135-150
{
    rawdev::EditParams p = rawdev::EditParams::new_();
    p.exposure = 1.5f; p.contrast = 25.0f; p.saturation = -10.0f; p.blacks = 0.005f;
    const std::string json = p.to_json();
    EXPECT(json.find("{\"exposure\":1.5,\"contrast\":25.0,") == 0);
    EXPECT(json.find("\"blacks\":0.005") != std::string::npos);
    EXPECT(rawdev::EditParams::from_json(json) == p);
    bool threw = false;
    try { rawdev::EditParams::from_json("{\"exposure\":1.0}"); } catch (const rawdev::Error &) { threw = true; }
    EXPECT(threw);                                 // serde: missing field
    // serde_json writes an f32 with ryu: exponent form below 1e-6 and from 1e13 on, "-0.0", no '+' and no padding
    const struct { float v; const char *text; } ryu[] = {
        { 1e-7f, "1e-7" }, { 1e20f, "1e20" }, { -0.0f, "-0.0" }, { 0.0f, "0.0" }, { 0.3f, "0.3" }, { 1.0f, "1.0" }, { 1234567.0f, "1234567.0" },
        { 1e13f, "1e13" }, { 1e12f, "1000000000000.0" }, { 1.5e-7f, "1.5e-7" }, { 16777216.0f, "16777216.0" }, { 1e-5f, "0.00001" },
        { 1e-6f, "0.000001" }, { 9.9e-7f, "9.9e-7" }, { 12.34f, "12.34" }, { 3.4028235e38f, "3.4028235e38" }, { -2.5f, "-2.5" },
        { 0.001234f, "0.001234" }, { 1.17549435e-38f, "1.1754944e-38" }, { 1e-45f, "1e-45" }, { 123456.79f, "123456.79" }, { 100.0f, "100.0" } };
    for (const auto &c : ryu) {
        const std::string got = rawdev::EditParams::ryu_f32(c.v);
        if (got != c.text) std::printf("ryu_f32(%a) = %s, expected %s\n", (double)c.v, got.c_str(), c.text);
        EXPECT(got == c.text);
    }
    rawdev::EditParams q;
    q.exposure = 1e-7f; q.contrast = 1e20f; q.tint = -0.0f;
    EXPECT(q.to_json().find("{\"exposure\":1e-7,\"contrast\":1e20,") == 0 && q.to_json().find("\"tint\":-0.0}") != std::string::npos);
    EXPECT(rawdev::EditParams::from_json(q.to_json()) == q);
}

static void test_reset()                          // edit.rs:152-163
{
    rawdev::EditParams p;
    p.exposure = 2.0f; p.contrast = 50.0f;
    EXPECT(!p.is_unedited());
    p.reset();
    EXPECT(p.is_unedited());
}

static void test_color_stub()                     // color.rs:184-206
{
    EXPECT(rawdev::is_identity_matrix({ 1, 0, 0, 0, 1, 0, 0, 0, 1 }));
    EXPECT(rawdev::is_identity_matrix({ 1.0005f, 0, 0, 0, 0.9995f, 0, 0, 0, 1 }));
    EXPECT(!rawdev::is_identity_matrix({ 1.002f, 0, 0, 0, 1, 0, 0, 0, 1 }));
    const auto m = rawdev::calculate_cam_to_srgb_matrix({ 0.8f, 0.1f, 0.1f, 0.2f, 0.7f, 0.1f, 0.0f, 0.3f, 0.9f });
    EXPECT(rawdev::is_identity_matrix(m));          // the reference's stub returns identity unconditionally
    bool any = false;
    for (float v : m) any = any || v != 0.0f;
    EXPECT(any);                                    // test_cam_to_srgb_calculation: "has a non-zero element"
}

static void test_errors_without_compute()
{
    bool threw = false;
    try {
        rawdev::RenderPipeline::new_(1, std::vector<uint16_t>(15), 4, 4, rawdev::EditParams(), { 1, 1, 1, 1 },
                                     { 1, 0, 0, 0, 1, 0, 0, 0, 1 });
    } catch (const rawdev::Error &e) { threw = true; EXPECT(e.code == RD_ERR_INVALID_ARG); }
    EXPECT(threw);
    uint32_t pw, ph, hw, hh;
    EXPECT(rd_derived_dims(6016, 4016, &pw, &ph, &hw, &hh) == RD_OK && pw == 1280 && ph == 854 && hw == 128 && hh == 85);
}

static void test_pipeline_gpu()
{
    const uint32_t w = 96, h = 64;
    std::vector<uint16_t> cfa((size_t)w * h);
    uint64_t s = 0x52415745ull;
    for (auto &v : cfa) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint16_t)((s >> 40) & 4095u); }
    rawdev::EditParams p;
    p.exposure = 0.5f; p.contrast = 4.0f; p.highlights = -0.25f; p.shadows = 0.5f; p.whites = 1.1f; p.blacks = 0.03f;
    p.vibrance = 0.4f; p.saturation = 15.0f; p.temperature = 0.2f; p.tint = -0.1f;
    const std::array<float, 4> wb = { 2.0f, 1.0f, 1.5f, 1.0f };
    const std::array<float, 9> cm = { 1.6f, -0.4f, -0.2f, -0.3f, 1.5f, -0.2f, 0.0f, -0.5f, 1.5f };
    auto pipe = rawdev::RenderPipeline::new_(42, cfa, w, h, p, wb, cm);
    EXPECT(pipe.dimensions() == std::make_pair(w, h) && pipe.image_id == 42);
    EXPECT(pipe.preview_width == 96 && pipe.histogram_width == 128);

    ref_uniforms u;
    std::memcpy(&u.p, &p, sizeof u.p);
    std::memcpy(u.wb, wb.data(), sizeof u.wb);
    std::memcpy(u.cm, cm.data(), sizeof u.cm);
    u.zoom = 1.0f; u.pan_x = u.pan_y = 0.0f; u.black_level = 0;
    std::vector<float> exp((size_t)w * h * 4);
    ref_render_f32(cfa.data(), w, h, &u, w, h, REF_POW_PINNED, exp.data());
    std::vector<uint8_t> exp8(exp.size());
    ref_pack_u8(exp.data(), exp.size(), exp8.data());

    const std::vector<uint8_t> full = pipe.render_full_res_to_bytes();
    EXPECT(full == exp8);
    rawdev::Histogram fused;
    const std::vector<float> f32 = pipe.render_f32(w, h, &fused);
    EXPECT(std::memcmp(f32.data(), exp.data(), exp.size() * sizeof(float)) == 0);
    const rawdev::Histogram hist = pipe.calculate_histogram(full);
    uint32_t ref_hist[768];
    ref_histogram(exp8.data(), (size_t)w * h, ref_hist);
    EXPECT(std::memcmp(&hist[0][0], ref_hist, sizeof ref_hist) == 0);
    EXPECT(std::memcmp(&fused[0][0], ref_hist, sizeof ref_hist) == 0);

    // view_develop(): update_uniforms_with_zoom + preview + histogram-size render (main.rs:1515-1534)
    pipe.update_uniforms_with_zoom(p, 2.0f, 0.1f, -0.05f);
    u.zoom = 2.0f; u.pan_x = 0.1f; u.pan_y = -0.05f;
    std::vector<float> pe((size_t)pipe.preview_width * pipe.preview_height * 4);
    ref_render_f32(cfa.data(), w, h, &u, pipe.preview_width, pipe.preview_height, REF_POW_PINNED, pe.data());
    std::vector<uint8_t> pe8(pe.size());
    ref_pack_u8(pe.data(), pe.size(), pe8.data());
    EXPECT(pipe.render_to_bytes() == pe8);
    std::vector<float> he((size_t)pipe.histogram_width * pipe.histogram_height * 4);
    ref_render_f32(cfa.data(), w, h, &u, pipe.histogram_width, pipe.histogram_height, REF_POW_PINNED, he.data());
    std::vector<uint8_t> he8(he.size());
    ref_pack_u8(he.data(), he.size(), he8.data());
    EXPECT(pipe.render_to_histogram_bytes() == he8);

    // Arc<RenderPipeline> shared with an export thread (main.rs:1749-1754)
    pipe.update_uniforms(p);
    std::vector<uint8_t> from_thread;
    std::thread t([&] { from_thread = pipe.render_full_res_to_bytes(); });
    const std::vector<uint8_t> here = pipe.render_full_res_to_bytes();
    t.join();
    EXPECT(from_thread == exp8 && here == exp8);

    // the export into memory the caller keeps: page-locked (the DMA engine writes it) and an ordinary reused buffer
    rawdev::PinnedBytes pin(exp8.size());
    std::memset(pin.data(), 0x5a, pin.size());
    pipe.render_full_res_into(pin.data(), pin.size());
    EXPECT(std::memcmp(pin.data(), exp8.data(), exp8.size()) == 0);
    std::vector<uint8_t> mine(exp8.size(), 0xa5);
    pipe.render_full_res_into(mine.data(), mine.size());
    EXPECT(mine == exp8);
    bool threw = false;
    try { pipe.render_full_res_into(mine.data(), mine.size() - 4); } catch (const rawdev::Error &) { threw = true; }
    EXPECT(threw);
    {
        const rawdev::RenderPipeline::Surface lent = pipe.render_full_res();       // the pipeline's own page-locked surface
        EXPECT(lent.size() == exp8.size() && std::memcmp(lent.data(), exp8.data(), exp8.size()) == 0);
    }
}

// The node-level batch entry through the C++ mirror: 5 frames on one device, surfaces and u64 histogram against the oracle.
static void test_node_batch_gpu()
{
    const uint32_t w = 256, h = 66, n = 5;
    const std::array<float, 4> wb = { 2.0f, 1.0f, 1.5f, 1.0f };
    const std::array<float, 9> cm = { 1.6f, -0.4f, -0.2f, -0.3f, 1.5f, -0.2f, 0.0f, -0.5f, 1.5f };
    rawdev::NodeBatch nb({ 0 }, w, h, RD_FMT_RGBA_F32, true);
    std::vector<std::vector<uint16_t>> cfas(n, std::vector<uint16_t>((size_t)w * h));
    std::vector<rawdev::DeviceBuffer> in, out;
    std::vector<rd_frame> frames(n);
    std::vector<std::vector<float>> exp(n, std::vector<float>((size_t)w * h * 4));
    uint64_t ref_total[768] = { 0 };
    uint64_t s = 99;
    for (uint32_t i = 0; i < n; ++i) {
        for (auto &v : cfas[i]) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint16_t)((s >> 40) & 4095u); }
        rawdev::EditParams p;
        p.exposure = 0.25f * (float)i - 0.5f; p.contrast = 2.0f * (float)i; p.saturation = 10.0f * (float)i; p.blacks = 0.01f * (float)i;
        EXPECT(nb.device_of(i) == 0);
        in.emplace_back(0, cfas[i].data(), cfas[i].size() * 2);
        out.emplace_back(0, (size_t)w * h * 16);
        std::memset(&frames[i], 0, sizeof(rd_frame));
        frames[i].cfa_dev = (const uint16_t *)in[i].get();
        frames[i].out_dev = out[i].get();
        frames[i].params = p;
        std::memcpy(frames[i].wb_multipliers, wb.data(), sizeof wb);
        std::memcpy(frames[i].color_matrix, cm.data(), sizeof cm);
        ref_uniforms u;
        std::memset(&u, 0, sizeof u);
        std::memcpy(&u.p, &p, sizeof u.p);
        std::memcpy(u.wb, wb.data(), sizeof u.wb);
        std::memcpy(u.cm, cm.data(), sizeof u.cm);
        u.zoom = 1.0f;
        ref_render_f32(cfas[i].data(), w, h, &u, w, h, REF_POW_PINNED, exp[i].data());
        std::vector<uint8_t> e8(exp[i].size());
        ref_pack_u8(exp[i].data(), exp[i].size(), e8.data());
        uint32_t rh[768];
        ref_histogram(e8.data(), (size_t)w * h, rh);
        for (int k = 0; k < 768; ++k) ref_total[k] += rh[k];
    }
    nb.develop(frames);
    const auto hist = nb.histogram();
    EXPECT(std::memcmp(&hist[0][0], ref_total, sizeof ref_total) == 0);
    for (uint32_t i = 0; i < n; ++i) {
        std::vector<float> got((size_t)w * h * 4);
        out[i].download(got.data(), got.size() * sizeof(float));
        EXPECT(std::memcmp(got.data(), exp[i].data(), got.size() * sizeof(float)) == 0);
    }
    bool threw = false;
    try { rawdev::NodeBatch bad({ 0, 0 }, w, h, RD_FMT_RGBA_F32); } catch (const rawdev::Error &e) { threw = true; EXPECT(e.code == RD_ERR_INVALID_ARG); }
    EXPECT(threw);                                         // a device listed twice needs RD_NODE_REDUCE=host
}

int main(int argc, char **argv)
{
    const bool gpu = argc > 1 && std::string(argv[1]) == "--gpu";
    test_default_is_unedited();
    test_serialization();
    test_reset();
    test_errors_without_compute();
    test_color_stub();
    if (gpu) {
        test_pipeline_gpu();
        test_node_batch_gpu();
    } else {
        int n = 0;
        if (rd_device_count(&n) != RD_OK || n == 0) {     // no CPU fallback: construction must fail loudly
            bool threw = false;
            try {
                rawdev::RenderPipeline::new_(1, std::vector<uint16_t>(16), 4, 4, rawdev::EditParams(), { 1, 1, 1, 1 },
                                             { 1, 0, 0, 0, 1, 0, 0, 0, 1 });
            } catch (const rawdev::Error &e) { threw = true; EXPECT(e.code == RD_ERR_NO_DEVICE || e.code == RD_ERR_HIP); }
            EXPECT(threw);
        }
    }
    std::printf("%s (%d failure(s))%s\n", failures ? "FAILED" : "ok", failures, gpu ? " [gpu]" : " [cpu]");
    return failures ? 1 : 0;
}
