// tools/microbench.hip -- A/B harness for the export kernel (not product code).
// Builds variants of the quad kernel from the product's own device functions (rd_kernels.h) and
// times them interleaved in ONE process (cdna_hip_programming.md section 5.4 rule 24) on 24 MP frames,
// cycling through NIN distinct inputs / NOUT outputs so nothing is served from the Infinity Cache.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>
#include "../raweditor_amd/csrc/rd_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { M_COMPUTE = 1, M_STORE = 2, M_HIST = 4, M_PLAIN_ST = 8, M_LDS_T = 16 };

// Variant kernel: same loop as rd_develop_quads, pieces switchable.
template <int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
mb_quads(const uint16_t *__restrict__ cfa, float *__restrict__ out, uint32_t W, uint32_t H,
         uint32_t stride_units, uint32_t stride_rem, rd_ku u, uint32_t *slab32)
{
    constexpr bool COMPUTE = MODE & M_COMPUTE, STORE = MODE & M_STORE, HIST = MODE & M_HIST;
    constexpr bool PLAIN = MODE & M_PLAIN_ST, LDST = MODE & M_LDS_T;
    __shared__ uint32_t lh[HIST ? 768 * RD_HK : 1];
    __shared__ rd_f4 stage[LDST ? BLOCK * 3 : 1];     // per lane: c1, c2, c3
    if (HIST) rd_hist_zero(lh);
    const uint32_t qpr = W >> 1, units = H / 2u + 1u;
    const uint32_t total = units * qpr;
    const uint32_t copy = threadIdx.x & (RD_HK - 1);
    uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t unit = item / qpr, q = item % qpr;
    uint32_t top = 0, bot = 0;
    float sink = 0.0f;
    auto st = [&](size_t px, rd_f4 v) {
        if (PLAIN) reinterpret_cast<rd_f4 *>(out)[px] = v;
        else __builtin_nontemporal_store(v, reinterpret_cast<rd_f4 *>(out) + px);
    };
    if (item < total) {
        uint32_t ra = unit ? 2u * unit - 1u : 0u, rb = 2u * unit < H ? 2u * unit : H - 1u;
        top = *reinterpret_cast<const uint32_t *>(cfa + (size_t)ra * W + 2u * q);
        bot = *reinterpret_cast<const uint32_t *>(cfa + (size_t)rb * W + 2u * q);
    }
    while (item < total) {
        uint32_t nitem = item + stride, nunit = unit + stride_units, nq = q + stride_rem;
        if (nq >= qpr) { nq -= qpr; nunit += 1u; }
        uint32_t ntop = 0, nbot = 0;
        if (nitem < total) {
            uint32_t ra = nunit ? 2u * nunit - 1u : 0u, rb = 2u * nunit < H ? 2u * nunit : H - 1u;
            ntop = *reinterpret_cast<const uint32_t *>(cfa + (size_t)ra * W + 2u * nq);
            nbot = *reinterpret_cast<const uint32_t *>(cfa + (size_t)rb * W + 2u * nq);
        }
        const bool has_a = unit != 0u, has_b = 2u * unit < H;
        const float A = rd_norm(top & 0xffffu, 0), B = rd_norm(top >> 16, 0);
        const float C = rd_norm(bot & 0xffffu, 0), D = rd_norm(bot >> 16, 0);
        rd_rgb c1, c2, c3;
        if (COMPUTE) { c1 = rd_colour(u, C, A, B); c2 = rd_colour(u, C, D, A); c3 = rd_colour(u, C, D, B); }
        else { c1 = { C, A, B }; c2 = { C, D, A }; c3 = { C, D, B }; }
        if (HIST) {
            rd_hist_add(lh, copy, rd_q8(c1.r), rd_q8(c1.g), rd_q8(c1.b), 2u);
            rd_hist_add(lh, copy, rd_q8(c2.r), rd_q8(c2.g), rd_q8(c2.b), 1u);
            rd_hist_add(lh, copy, rd_q8(c3.r), rd_q8(c3.g), rd_q8(c3.b), 1u);
        }
        if (STORE) {
            if (LDST) {
                // wave-private transpose through LDS: lane l of wave w owns stage[(w*64+l)*3 + k].
                // All 64 lanes of a wave always hold 64 consecutive quads of one row? Only when the
                // wave does not straddle a row end; microbench assumes qpr % 64 == 0 (6016/2 = 3008 = 47*64).
                const uint32_t lane = threadIdx.x & 63u, wbase = (threadIdx.x & ~63u) * 3u;
                stage[wbase + lane * 3u + 0] = rd_f4{ c1.r, c1.g, c1.b, 1.0f };
                stage[wbase + lane * 3u + 1] = rd_f4{ c2.r, c2.g, c2.b, 1.0f };
                stage[wbase + lane * 3u + 2] = rd_f4{ c3.r, c3.g, c3.b, 1.0f };
                __builtin_amdgcn_wave_barrier();
                const uint32_t q0 = q - lane;                 // first quad of this wave
                const size_t rowa = (size_t)(2u * unit - 1u) * W + 2u * q0, rowb = (size_t)(2u * unit) * W + 2u * q0;
                // row a: pixel p (0..127) = c1 of quad p/2 ; row b: pixel p = (p&1 ? c3 : c2) of quad p/2
                for (uint32_t half = 0; half < 2; ++half) {
                    const uint32_t p = half * 64u + lane;
                    if (has_a) st(rowa + p, stage[wbase + (p >> 1) * 3u + 0]);
                    if (has_b) st(rowb + p, stage[wbase + (p >> 1) * 3u + 1u + (p & 1u)]);
                }
                __builtin_amdgcn_wave_barrier();
            } else {
                if (has_a) {
                    const size_t px = (size_t)(2u * unit - 1u) * W + 2u * q;
                    st(px, rd_f4{ c1.r, c1.g, c1.b, 1.0f }); st(px + 1, rd_f4{ c1.r, c1.g, c1.b, 1.0f });
                }
                if (has_b) {
                    const size_t px = (size_t)(2u * unit) * W + 2u * q;
                    st(px, rd_f4{ c2.r, c2.g, c2.b, 1.0f }); st(px + 1, rd_f4{ c3.r, c3.g, c3.b, 1.0f });
                }
            }
        } else {
            sink += c1.r + c1.g + c1.b + c2.r + c2.g + c2.b + c3.r + c3.g + c3.b;
        }
        item = nitem; unit = nunit; q = nq; top = ntop; bot = nbot;
    }
    if (!STORE && sink == 1234.5678f) out[0] = sink;     // keep the arithmetic alive
    if (HIST) rd_hist_flush(lh, slab32, nullptr);
}

// pure streaming writes: the surface's bytes with no arithmetic, contiguous 16 B per lane
__global__ void __launch_bounds__(1024) mb_fill(float *__restrict__ out, size_t n4, int nt)
{
    rd_f4 v = { 0.25f, 0.5f, 0.75f, 1.0f };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (nt) __builtin_nontemporal_store(v, reinterpret_cast<rd_f4 *>(out) + i);
        else reinterpret_cast<rd_f4 *>(out)[i] = v;
    }
}

#include <functional>
typedef float rd_f2 __attribute__((ext_vector_type(2)));
// ---- VALU calibration: ITER x 16 independent ops per lane -----------------------------------------
template <int KIND>
__global__ void __launch_bounds__(1024) mb_valu(float *out, float a, float b, int iters)
{
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) acc[i] = __builtin_fmaf(acc[i], a, b);
            else if (KIND == 1) { acc[i] = acc[i] * a; acc[i] = acc[i] + b; }
            else if (KIND == 2) acc[i] = acc[i] / a;
            else if (KIND == 3) acc[i] = rd_gamma_clamp(acc[i]) + b;
        }
        if (KIND == 4) {
            rd_f2 *v = reinterpret_cast<rd_f2 *>(acc);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], rd_f2{ a, a }, rd_f2{ b, b });
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 1234.5678f) out[0] = s;
}

struct Variant { std::string name; std::function<void(int)> run; std::vector<float> ms; };
#include <functional>

int main(int argc, char **argv)
{
    const uint32_t W = 6016, H = 4016;
    const int NIN = 8, NOUT = 4, ROUNDS = argc > 1 ? atoi(argv[1]) : 6, REP = 8;
    const size_t in_bytes = (size_t)W * H * 2, out_bytes = (size_t)W * H * 16;
    std::vector<uint16_t *> din(NIN); std::vector<float *> dout(NOUT);
    std::vector<uint16_t> host((size_t)W * H);
    for (int i = 0; i < NIN; ++i) {
        uint64_t s = 0x52415745ull + i;
        for (auto &v : host) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint16_t)((s >> 40) & 4095u); }
        CK(hipMalloc((void **)&din[i], in_bytes)); CK(hipMemcpy(din[i], host.data(), in_bytes, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < NOUT; ++i) CK(hipMalloc((void **)&dout[i], out_bytes));
    uint32_t *slab; CK(hipMalloc((void **)&slab, (size_t)RD_MAX_BLOCKS * 768 * 4));
    rd_edit_params p{};
    p.exposure = 0.7f; p.contrast = 3.0f; p.highlights = -0.3f; p.shadows = 0.4f; p.whites = 1.1f; p.blacks = 0.02f;
    p.vibrance = 0.3f; p.saturation = 20.0f; p.temperature = 0.2f; p.tint = -0.1f;
    const float wb[4] = { 2.0f, 1.0f, 1.5f, 1.0f };
    const float cm[9] = { 1.6f, -0.4f, -0.2f, -0.3f, 1.5f, -0.2f, 0.0f, -0.5f, 1.5f };
    rd_ku u = rd_make_ku(p, wb, cm, 1.0f, 0.0f, 0.0f, 0);
    const uint32_t qpr = W / 2;
    hipStream_t s; CK(hipStreamCreate(&s));

    std::vector<Variant> vs;
#define ADD(NAME, MODE, BLOCK, BLOCKS)                                                                   \
    vs.push_back({ NAME, [&, qpr](int k) {                                                               \
        const uint32_t stride = (BLOCKS) * (BLOCK);                                                      \
        hipLaunchKernelGGL((mb_quads<MODE, BLOCK>), dim3(BLOCKS), dim3(BLOCK), 0, s, din[k % NIN], dout[k % NOUT], W, H, \
                           stride / qpr, stride % qpr, u, slab); }, {} })
    ADD("full nt hist  1024x256", M_COMPUTE | M_STORE | M_HIST, 1024, 256);
    ADD("full nt nohist 1024x256", M_COMPUTE | M_STORE, 1024, 256);
    ADD("full nt nohist 1024x512", M_COMPUTE | M_STORE, 1024, 512);
    ADD("full nt nohist 256x2048", M_COMPUTE | M_STORE, 256, 2048);
    ADD("full plain nohist 1024x512", M_COMPUTE | M_STORE | M_PLAIN_ST, 1024, 512);
    ADD("full ldsT nt nohist 1024x512", M_COMPUTE | M_STORE | M_LDS_T, 1024, 512);
    ADD("compute only 1024x512", M_COMPUTE, 1024, 512);
    ADD("compute+hist 1024x256", M_COMPUTE | M_HIST, 1024, 256);
    ADD("store only nt 1024x512", M_STORE, 1024, 512);
    ADD("store only plain 1024x512", M_STORE | M_PLAIN_ST, 1024, 512);
    ADD("store only ldsT nt 1024x512", M_STORE | M_LDS_T, 1024, 512);
    ADD("store+hist nt 1024x256", M_STORE | M_HIST, 1024, 256);
    unsigned long long *slab64; CK(hipMalloc((void **)&slab64, (size_t)RD_MAX_BLOCKS * 768 * 8)); CK(hipMemset(slab64, 0, (size_t)RD_MAX_BLOCKS * 768 * 8));
    const uint32_t tpu = (qpr + 63) / 64;
#define PROD(NAME, FMT, HIST, BLOCKS, OUTP, S32, S64)                                                   \
    vs.push_back({ NAME, [&, qpr, tpu](int k) { const uint32_t nw = (BLOCKS) * RD_WAVES;               \
        hipLaunchKernelGGL((rd_develop_quads<FMT, HIST>), dim3(BLOCKS), dim3(1024), 0, s, din[k % NIN], (void *)dout[k % NOUT], \
                           W, H, 0u, H / 2 + 1, tpu, nw / tpu, nw % tpu, u, S32, S64); }, {} })
    PROD("PRODUCT f32 hist slab32 x256", 0, true, 256, 0, slab, (unsigned long long *)nullptr);
    PROD("PRODUCT f32 hist slab64 x256", 0, true, 256, 0, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f32 nohist x256", 0, false, 256, 0, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    PROD("PRODUCT f32 nohist x512", 0, false, 512, 0, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    PROD("PRODUCT f16 hist x256", 1, true, 256, 0, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f16 nohist x512", 1, false, 512, 0, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    PROD("PRODUCT u8 hist x256", 2, true, 256, 0, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT u8 nohist x512", 2, false, 512, 0, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    vs.push_back({ "fill contiguous nt", [&](int k) { hipLaunchKernelGGL(mb_fill, dim3(2048), dim3(1024), 0, s, dout[k % NOUT], out_bytes / 16, 1); }, {} });
    vs.push_back({ "fill contiguous plain", [&](int k) { hipLaunchKernelGGL(mb_fill, dim3(2048), dim3(1024), 0, s, dout[k % NOUT], out_bytes / 16, 0); }, {} });

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // VALU calibration: lane-ops per second
        const char *names[5] = { "v_fma_f32", "v_mul+v_add (2 ops)", "IEEE div", "gamma_clamp+add", "v_pk_fma_f32 (2 lanes-ops)" };
        const double ops_per[5] = { 1, 2, 1, 1, 1 };
        for (int blocks : { 256, 512 }) {
            for (int kind = 0; kind < 5; ++kind) {
                const int iters = (kind == 2 || kind == 3) ? 256 : 2048;
                auto launch = [&]() {
                    switch (kind) {
                    case 0: hipLaunchKernelGGL(mb_valu<0>, dim3(blocks), dim3(1024), 0, s, dout[0], 0.999f, 0.001f, iters); break;
                    case 1: hipLaunchKernelGGL(mb_valu<1>, dim3(blocks), dim3(1024), 0, s, dout[0], 0.999f, 0.001f, iters); break;
                    case 2: hipLaunchKernelGGL(mb_valu<2>, dim3(blocks), dim3(1024), 0, s, dout[0], 0.999f, 0.001f, iters); break;
                    case 3: hipLaunchKernelGGL(mb_valu<3>, dim3(blocks), dim3(1024), 0, s, dout[0], 0.999f, 0.001f, iters); break;
                    default: hipLaunchKernelGGL(mb_valu<4>, dim3(blocks), dim3(1024), 0, s, dout[0], 0.999f, 0.001f, iters); break;
                    }
                };
                launch(); CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s)); for (int k = 0; k < 4; ++k) launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
                double n = (double)blocks * 1024 * 16 * iters * ops_per[kind];
                double wave_instr_per_simd = n / 64 / 1024;            // per SIMD
                printf("calib %-28s blocks %4d: %8.1f us  %7.2f T lane-ops/s  %6.2f ns per wave-instr per SIMD\n",
                       names[kind], blocks, ms * 1e3, n / (ms * 1e-3) / 1e12, ms * 1e6 / wave_instr_per_simd);
            }
        }
    }
    if (argc > 2) {   // sustained run of the full kernel: burst vs throttled clock
        auto &v = vs[13];   // PRODUCT f32 hist slab64
        for (int w = 0; w < atoi(argv[2]); ++w) {
            CK(hipEventRecord(e0, s)); for (int k = 0; k < 256; ++k) v.run(k); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("sustained window %2d: %.1f us per launch\n", w, ms * 1e3 / 256);
        }
    }
    for (int r = -1; r < ROUNDS; ++r) {
        for (auto &v : vs) {
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < REP; ++k) v.run(k);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 0) v.ms.push_back(ms / REP);
        }
    }
    printf("%-34s %10s %10s %10s %8s\n", "variant", "med us", "min us", "GB/s(18B)", "MP/s");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("%-34s %10.1f %10.1f %10.0f %8.0f\n", v.name.c_str(), med * 1e3, mn * 1e3,
               (double)W * H * 18 / (med * 1e-3) / 1e9, (double)W * H / 1e6 / (med * 1e-3));
    }
    return 0;
}
