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
#ifdef MB_LITE
#define RD_COLOUR_HOOK_HEADER "../../tools/mb_lite.h"
#endif
#include "../raweditor_amd/csrc/rd_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { M_COMPUTE = 1, M_STORE = 2, M_HIST = 4, M_PLAIN_ST = 8, M_LDS_T = 16, M_FAKE_CONTIG = 32, M_LDS_ONLY = 64, M_LITE = 128, M_LITE2 = 256 };

#ifndef MB_LITE
#include "mb_lite.h"
#endif
// Variant kernel: same loop as rd_develop_quads, pieces switchable.
template <int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
mb_quads(const uint16_t *__restrict__ cfa, float *__restrict__ out, uint32_t W, uint32_t H,
         uint32_t stride_units, uint32_t stride_rem, rd_ku u, uint32_t *slab32)
{
    constexpr bool COMPUTE = MODE & M_COMPUTE, STORE = MODE & M_STORE, HIST = MODE & M_HIST;
    constexpr bool PLAIN = MODE & M_PLAIN_ST, LDST = MODE & M_LDS_T, FAKE = MODE & M_FAKE_CONTIG, LDSONLY = MODE & M_LDS_ONLY;
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
        if (MODE & M_LITE) { c1 = mb_colour_lite(u, C, A, B, 1); c2 = mb_colour_lite(u, C, D, A, 1); c3 = mb_colour_lite(u, C, D, B, 1); }
        else if (MODE & M_LITE2) { c1 = mb_colour_lite(u, C, A, B, 0); c2 = mb_colour_lite(u, C, D, A, 0); c3 = mb_colour_lite(u, C, D, B, 0); }
        else if (COMPUTE) { c1 = rd_colour_m<0>(u, C, A, B); c2 = rd_colour_m<0>(u, C, D, A); c3 = rd_colour_m<0>(u, C, D, B); }
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
                    if (LDSONLY) { rd_f4 a = stage[wbase + (p >> 1) * 3u + 0], b = stage[wbase + (p >> 1) * 3u + 1u + (p & 1u)]; sink += a.x + b.y; }
                    else {
                    if (has_a) st(rowa + p, stage[wbase + (p >> 1) * 3u + 0]);
                    if (has_b) st(rowb + p, stage[wbase + (p >> 1) * 3u + 1u + (p & 1u)]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else if (FAKE) {
                const uint32_t lane = threadIdx.x & 63u; const uint32_t q0 = q - lane;
                const size_t rowa = (size_t)(2u * unit - 1u) * W + 2u * q0, rowb = (size_t)(2u * unit) * W + 2u * q0;
                if (has_a) { st(rowa + lane, rd_f4{ c1.r, c1.g, c1.b, 1.0f }); st(rowa + 64 + lane, rd_f4{ c1.r, c1.g, c1.b, 1.0f }); }
                if (has_b) { st(rowb + lane, rd_f4{ c2.r, c2.g, c2.b, 1.0f }); st(rowb + 64 + lane, rd_f4{ c3.r, c3.g, c3.b, 1.0f }); }
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
// store-pattern probe: every wave writes `ROWS` rows x (4 KiB / ROWS) contiguous bytes per tile, like the
// export kernel's tile (ROWS = 2) or a one-row tile (ROWS = 1); optional u16 input loads; nt or plain.
template <int ROWS, bool NT, bool LOADS>
__global__ void __launch_bounds__(1024) mb_store_pat(const uint16_t *__restrict__ cfa, float *__restrict__ out, uint32_t W, uint32_t H)
{
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 16u + (threadIdx.x >> 6), nwaves = gridDim.x * 16u;
    const uint32_t px_per_tile = 256u / ROWS;                     // pixels per row per tile
    const uint32_t tpr = W / px_per_tile;                         // tiles per row(-pair)
    const uint32_t ntiles = (H / ROWS) * tpr;
    rd_f4 *o = reinterpret_cast<rd_f4 *>(out);
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        const uint32_t rp = t / tpr, c0 = (t % tpr) * px_per_tile;
        float v = 0.5f;
        if (LOADS) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(cfa + (size_t)(rp * ROWS) * W + c0 * (ROWS == 2 ? 1 : 1));
            v = (float)(src[lane] & 0xfffu) * (1.0f / 4096.0f);
            if (ROWS == 2) v += (float)(reinterpret_cast<const uint32_t *>(cfa + (size_t)(rp * 2 + 1) * W + c0)[lane] >> 16) * (1.0f / 4096.0f);
            else v += (float)(src[lane + 64] >> 16) * (1.0f / 4096.0f);
        }
        const rd_f4 val = { v, v, v, 1.0f };
#pragma unroll
        for (uint32_t r = 0; r < ROWS; ++r)
#pragma unroll
            for (uint32_t k = 0; k < 4u / ROWS; ++k) {
                rd_f4 *dst = o + (size_t)(rp * ROWS + r) * W + c0 + k * 64u + lane;
                if (NT) __builtin_nontemporal_store(val, dst); else *dst = val;
            }
    }
}

// read:write = 1:8 streaming mix in its simplest form: a wave reads LW*64 contiguous bytes and writes 8x as
// many contiguous bytes.  Tells whether ~4.9 TB/s is a property of the traffic mix or of the kernel's pattern.
template <int LW, bool PREFETCH>
__global__ void __launch_bounds__(1024) mb_mix(const uint32_t *__restrict__ in, float *__restrict__ out, size_t in_dwords)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * 16u + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 16u;
    const size_t chunk_dw = (size_t)LW / 4 * 64;                 // input dwords per wave-chunk
    const size_t nchunks = in_dwords / chunk_dw;
    rd_f4 *o = reinterpret_cast<rd_f4 *>(out);
    size_t c = wave;
    if (c >= nchunks) return;
    u4 cur = { 0, 0, 0, 0 };
    auto ld = [&](size_t cc) { u4 v = { 0, 0, 0, 0 };
        if (LW == 4) v.x = in[cc * chunk_dw + lane];
        else {
#pragma unroll
            for (int k = 0; k < LW / 16; ++k) { u4 w = reinterpret_cast<const u4 *>(in + cc * chunk_dw)[k * 64 + lane]; v.x ^= w.x; v.y ^= w.y; v.z ^= w.z; v.w ^= w.w; }
        }
        return v; };
    cur = ld(c);
    if (PREFETCH) asm volatile("" : "+v"(cur));
    for (;;) {
        const size_t n = c + nwaves;
        const bool more = n < nchunks;
        u4 nxt = cur;
        if (PREFETCH) nxt = ld(more ? n : c);
        const float f = (float)(cur.x & 0xfffu) * (1.0f / 4096.0f) + (float)(cur.w & 1u);
        const rd_f4 val = { f, f, f, 1.0f };
        // output: 8x the input bytes, contiguous: LW*8/16 float4 per lane
#pragma unroll
        for (uint32_t k = 0; k < LW * 8 / 16; ++k)
            __builtin_nontemporal_store(val, o + (c * (LW * 8 / 16) + k) * 64u + lane);
        if (!more) break;
        if (!PREFETCH) nxt = ld(n);
        c = n; cur = nxt;
    }
}

// same 1:8 mix, but D loads in flight per wave (register ring, fully unrolled): is the read penalty latency?
template <int D>
__global__ void __launch_bounds__(1024) mb_mix_deep(const uint32_t *__restrict__ in, float *__restrict__ out, size_t in_dwords)
{
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * 16u + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 16u;
    const size_t nchunks = in_dwords / 64;                      // 256 B per wave-chunk, 2 KiB written
    rd_f4 *o = reinterpret_cast<rd_f4 *>(out);
    uint32_t ring[D];
    size_t c = wave;
#pragma unroll
    for (int k = 0; k < D; ++k) { size_t cc = c + k * nwaves; ring[k] = in[(cc < nchunks ? cc : wave) * 64 + lane]; }
    while (c < nchunks) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (c < nchunks) {
                const uint32_t cur = ring[k];
                size_t nn = c + (size_t)D * nwaves;
                ring[k] = in[(nn < nchunks ? nn : wave) * 64 + lane];
                const float f = (float)(cur & 0xfffu) * (1.0f / 4096.0f);
                const rd_f4 val = { f, f, f, 1.0f };
                __builtin_nontemporal_store(val, o + (c * 2) * 64u + lane);
                __builtin_nontemporal_store(val, o + (c * 2 + 1) * 64u + lane);
                c += nwaves;
            }
        }
    }
}

// write-pattern probe: a wave-tile = ROWS rows x RUN_KB KiB contiguous per row (no loads); block-major or
// block-cyclic wave->tile order; ORDER 0 = all of row a then row b, ORDER 1 = 2-KiB pieces interleaved (a0 b0 a1 b1).
template <int ROWS, int RUN_KB, bool CYCLIC, int ORDER = 0>
__global__ void __launch_bounds__(1024) mb_wpat(float *__restrict__ out, uint32_t W, uint32_t H)
{
    const uint32_t lane = threadIdx.x & 63u, w_in_b = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * 16u;
    const uint32_t wave = CYCLIC ? w_in_b * gridDim.x + blockIdx.x : blockIdx.x * 16u + w_in_b;
    const uint32_t px_per_tile = RUN_KB * 64u;                    // 16 B per px
    const uint32_t tpr = W / px_per_tile;
    const uint32_t ntiles = (H / ROWS) * tpr;
    rd_f4 *o = reinterpret_cast<rd_f4 *>(out);
    const rd_f4 val = { 0.25f, 0.5f, 0.75f, 1.0f };
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
        const uint32_t rp = t / tpr, c0 = (t % tpr) * px_per_tile;
        if (ORDER == 0) {
#pragma unroll
            for (uint32_t r = 0; r < ROWS; ++r)
#pragma unroll
                for (uint32_t k = 0; k < RUN_KB; ++k)
                    __builtin_nontemporal_store(val, o + (size_t)(rp * ROWS + r) * W + c0 + k * 64u + lane);
        } else {
#pragma unroll
            for (uint32_t p = 0; p < RUN_KB / 2; ++p)
#pragma unroll
                for (uint32_t r = 0; r < ROWS; ++r)
#pragma unroll
                    for (uint32_t k = 0; k < 2; ++k)
                        __builtin_nontemporal_store(val, o + (size_t)(rp * ROWS + r) * W + c0 + (p * 2 + k) * 64u + lane);
        }
    }
}

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
    const int NIN = getenv("MB_NIN") ? atoi(getenv("MB_NIN")) : 8, NOUT = 4, ROUNDS = argc > 1 ? atoi(argv[1]) : 6, REP = 8;
    const size_t in_bytes = (size_t)W * H * 2, out_bytes = (size_t)W * H * 16;
    std::vector<uint16_t *> din(NIN); std::vector<float *> dout(NOUT);
    std::vector<uint16_t> host((size_t)W * H);
    for (int i = 0; i < NIN; ++i) {
        uint64_t s = 0x52415745ull + i;
        const bool flat = getenv("MB_FLAT") != nullptr;     // constant frame: worst case for histogram contention
        for (auto &v : host) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = flat ? (uint16_t)2048 : (uint16_t)((s >> 40) & 4095u); }
        CK(hipMalloc((void **)&din[i], in_bytes)); CK(hipMemcpy(din[i], host.data(), in_bytes, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < NOUT; ++i) CK(hipMalloc((void **)&dout[i], out_bytes));
    uint32_t *slab; CK(hipMalloc((void **)&slab, (size_t)RD_MAX_BLOCKS * 768 * 4));
    rd_edit_params p{};
    p.exposure = 0.7f; p.contrast = 3.0f; p.highlights = -0.3f; p.shadows = 0.4f; p.whites = 1.1f; p.blacks = 0.02f;
    p.vibrance = 0.3f; p.saturation = 20.0f; p.temperature = 0.2f; p.tint = -0.1f;
    const float wb[4] = { 2.0f, 1.0f, 1.5f, 1.0f };
    const float cm[9] = { 1.6f, -0.4f, -0.2f, -0.3f, 1.5f, -0.2f, 0.0f, -0.5f, 1.5f };
#ifndef MATHMODE
#define MATHMODE 0
#endif
    rd_ku u = rd_make_ku(p, wb, cm, 1.0f, 0.0f, 0.0f, 0, MATHMODE);
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
    ADD("lite1 compute only x512", M_COMPUTE | M_LITE, 1024, 512);
    ADD("lite1 + ldsT st x512", M_COMPUTE | M_LITE | M_STORE | M_LDS_T, 1024, 512);
    ADD("lite2 compute only x512", M_COMPUTE | M_LITE2, 1024, 512);
    ADD("lite2 + ldsT st x512", M_COMPUTE | M_LITE2 | M_STORE | M_LDS_T, 1024, 512);
    ADD("lite2 + ldsT st x256", M_COMPUTE | M_LITE2 | M_STORE | M_LDS_T, 1024, 256);
    ADD("lite2 + ldsT st x1024", M_COMPUTE | M_LITE2 | M_STORE | M_LDS_T, 1024, 1024);
    ADD("compute+fake contig st x512", M_COMPUTE | M_STORE | M_FAKE_CONTIG, 1024, 512);
    ADD("compute+ldsT no global st x512", M_COMPUTE | M_STORE | M_LDS_T | M_LDS_ONLY, 1024, 512);
    ADD("fake contig st only x512", M_STORE | M_FAKE_CONTIG, 1024, 512);
    ADD("compute+hist 1024x256", M_COMPUTE | M_HIST, 1024, 256);
    ADD("store only nt 1024x512", M_STORE, 1024, 512);
    ADD("store only plain 1024x512", M_STORE | M_PLAIN_ST, 1024, 512);
    ADD("store only ldsT nt 1024x512", M_STORE | M_LDS_T, 1024, 512);
    ADD("store+hist nt 1024x256", M_STORE | M_HIST, 1024, 256);
    unsigned long long *slab64; CK(hipMalloc((void **)&slab64, (size_t)RD_MAX_BLOCKS * 768 * 8)); CK(hipMemset(slab64, 0, (size_t)RD_MAX_BLOCKS * 768 * 8));
    const uint32_t tpu = (qpr + 63) / 64;
    uint32_t *tq; CK(hipMalloc((void **)&tq, 256 * 128)); CK(hipMemset(tq, 0, 256 * 128));   // ticket counters
#ifndef MATHMODE
#define MATHMODE 0
#endif
#define PRODS(NAME, FMT, HIST, BLOCKS, BURST, S32, S64)                                                  \
    vs.push_back({ NAME, [&, qpr, tpu](int k) {                                                         \
        hipLaunchKernelGGL((rd_develop_quads<FMT, HIST, true, MATHMODE, BURST>), dim3(BLOCKS), dim3(RD_BLOCK), 0, s, din[k % NIN], (void *)dout[k % NOUT], \
                           W, H, 0u, H / 2 + 1, tpu, (uint32_t)((1ull << 32) / tpu), 0u, 0u, tq, u, S32, S64); }, {} })
#define PROD(NAME, FMT, HIST, BLOCKS, BURST, S32, S64)                                                   \
    vs.push_back({ NAME, [&, qpr, tpu](int k) { const uint32_t nw = (BLOCKS) * RD_WAVES, nt = (H / 2 + 1) * tpu;  \
        const uint32_t tk = ((BLOCKS) % 16 == 0) ? (BLOCKS) / 4 : 1, tmax = (nt - nw + tk - 1) / tk;     \
        hipLaunchKernelGGL((rd_develop_quads<FMT, HIST, true, MATHMODE, BURST>), dim3(BLOCKS), dim3(RD_BLOCK), 0, s, din[k % NIN], (void *)dout[k % NOUT], \
                           W, H, 0u, H / 2 + 1, tpu, (uint32_t)((1ull << 32) / tpu), tk, tmax, tq, u, S32, S64); }, {} })
    PROD("PRODUCT f32 hist slab32 x256", 0, true, 256, false, slab, (unsigned long long *)nullptr);
    PROD("PRODUCT f32 hist slab64 x256", 0, true, 256, false, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f32 hist slab64 x256 BURST", 0, true, 256, true, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f32 hist slab64 x512", 0, true, 512, false, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f32 hist slab64 x512 BURST", 0, true, 512, true, (uint32_t *)nullptr, slab64);
    PRODS("STATIC  f32 hist slab64 x512 BURST", 0, true, 512, true, (uint32_t *)nullptr, slab64);
    PRODS("STATIC  f32 nohist x512 BURST", 0, false, 512, true, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    PRODS("STATIC  u8 hist x256", 2, true, 256, false, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f32 nohist x256", 0, false, 256, false, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    PROD("PRODUCT f32 nohist x512", 0, false, 512, false, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    PROD("PRODUCT f32 nohist x512 BURST", 0, false, 512, true, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    PROD("PRODUCT f16 hist x256", 1, true, 256, false, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f16 hist x256 BURST", 1, true, 256, true, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f16 nohist x512", 1, false, 512, false, (uint32_t *)nullptr, (unsigned long long *)nullptr);
    PROD("PRODUCT u8 hist x256", 2, true, 256, false, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT u8 hist x512", 2, true, 512, false, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT f16 hist x512", 1, true, 512, false, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT rgb8 hist x512", 3, true, 512, false, (uint32_t *)nullptr, slab64);
    PROD("PRODUCT u8 nohist x512", 2, false, 512, false, (uint32_t *)nullptr, (unsigned long long *)nullptr);
#define SPAT(NAME, ROWS, NT, LOADS, BLOCKS) vs.push_back({ NAME, [&](int k) { hipLaunchKernelGGL((mb_store_pat<ROWS, NT, LOADS>), dim3(BLOCKS), dim3(1024), 0, s, din[k % NIN], dout[k % NOUT], W, H); }, {} })
    SPAT("pat 2rows nt loads x512", 2, true, true, 512);
    SPAT("pat 2rows nt noloads x512", 2, true, false, 512);
    SPAT("pat 2rows plain loads x512", 2, false, true, 512);
    SPAT("pat 2rows plain noloads x512", 2, false, false, 512);
    SPAT("pat 1row nt loads x512", 1, true, true, 512);
    SPAT("pat 1row nt noloads x512", 1, true, false, 512);
    SPAT("pat 2rows nt loads x256", 2, true, true, 256);
    SPAT("pat 2rows nt loads x1024", 2, true, true, 1024);
    SPAT("pat 2rows nt loads x2048", 2, true, true, 2048);
#define MIX(NAME, LW, PF, BLOCKS) vs.push_back({ NAME, [&](int k) { hipLaunchKernelGGL((mb_mix<LW, PF>), dim3(BLOCKS), dim3(1024), 0, s, (const uint32_t *)din[k % NIN], dout[k % NOUT], in_bytes / 4); }, {} })
    MIX("mix 1:8 load4B prefetch x512", 4, true, 512);
    MIX("mix 1:8 load4B noprefetch x512", 4, false, 512);
    MIX("mix 1:8 load16B prefetch x512", 16, true, 512);
    MIX("mix 1:8 load16B noprefetch x512", 16, false, 512);
    MIX("mix 1:8 load16B prefetch x2048", 16, true, 2048);
    MIX("mix 1:8 load4B prefetch x2048", 4, true, 2048);
#define MIXD(NAME, D, BLOCKS) vs.push_back({ NAME, [&](int k) { hipLaunchKernelGGL((mb_mix_deep<D>), dim3(BLOCKS), dim3(1024), 0, s, (const uint32_t *)din[k % NIN], dout[k % NOUT], in_bytes / 4); }, {} })
    MIXD("mixdeep D=1 load4B x512", 1, 512);
    MIXD("mixdeep D=2 load4B x512", 2, 512);
    MIXD("mixdeep D=4 load4B x512", 4, 512);
    MIXD("mixdeep D=8 load4B x512", 8, 512);
    MIXD("mixdeep D=4 load4B x256", 4, 256);
    MIXD("mixdeep D=8 load4B x256", 8, 256);
    MIX("mix 1:8 load64B prefetch x512", 64, true, 512);
    MIX("mix 1:8 load64B prefetch x2048", 64, true, 2048);
    MIX("mix 1:8 load256B prefetch x512", 256, true, 512);
    MIX("mix 1:8 load256B prefetch x2048", 256, true, 2048);
#define MIXC(NAME, LW, PF, BLOCKS) vs.push_back({ NAME, [&](int k) { hipLaunchKernelGGL((mb_mix<LW, PF>), dim3(BLOCKS), dim3(1024), 0, s, (const uint32_t *)din[0], dout[k % NOUT], in_bytes / 4); }, {} })
    MIXC("mix 1:8 SAME input load16B x512", 16, true, 512);
    MIXC("mix 1:8 SAME input load4B x512", 4, true, 512);
    MIXC("mix 1:8 SAME input load16B x2048", 16, true, 2048);
#define WPAT(NAME, ROWS, RUN, CYC, ORD, BLOCKS) vs.push_back({ NAME, [&](int k) { hipLaunchKernelGGL((mb_wpat<ROWS, RUN, CYC, ORD>), dim3(BLOCKS), dim3(1024), 0, s, dout[k % NOUT], W, H); }, {} })
    WPAT("wpat 2rows x 2KB blockmajor x512", 2, 2, false, 0, 512);
    WPAT("wpat 2rows x 2KB cyclic x512", 2, 2, true, 0, 512);
    WPAT("wpat 2rows x 2KB cyclic x256", 2, 2, true, 0, 256);
    WPAT("wpat 2rows x 4KB blockmajor x512", 2, 4, false, 0, 512);
    WPAT("wpat 2rows x 4KB cyclic x512", 2, 4, true, 0, 512);
    WPAT("wpat 2rows x 4KB cyclic a0b0a1b1 x512", 2, 4, true, 1, 512);
    WPAT("wpat 2rows x 16KB cyclic x512", 2, 16, true, 0, 512);
    WPAT("wpat 1row x 4KB blockmajor x512", 1, 4, false, 0, 512);
    WPAT("wpat 1row x 1KB blockmajor x512", 1, 1, false, 0, 512);
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
#ifdef RD_PROBE
    {   // per-workgroup timeline of the product kernel: s_memtime (shader clock) against s_memrealtime (100 MHz), HW_ID, tiles
        const uint32_t nw = 512 * RD_WAVES, nt = (H / 2 + 1) * tpu, tk = 128, tmax = (nt - nw + tk - 1) / tk;
        std::vector<uint32_t> hc(512 * 8);
        const int nrep = getenv("MB_CLOCK_REPS") ? atoi(getenv("MB_CLOCK_REPS")) : 3;
        for (int rep = 0; rep < nrep; ++rep) {
            for (int k = 0; k < 256; ++k)
                hipLaunchKernelGGL((rd_develop_quads<0, true, true, MATHMODE, true>), dim3(512), dim3(RD_BLOCK), 0, s, din[k % NIN], (void *)dout[k % NOUT],
                                   W, H, 0u, H / 2 + 1, tpu, (uint32_t)((1ull << 32) / tpu), tk, tmax, tq, u, (uint32_t *)nullptr, slab64);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpyFromSymbol(hc.data(), HIP_SYMBOL(rd_probe_buf), 512 * 8 * 4));
            double st = 0, sr = 0, mx = 0; uint32_t t_min = 0xffffffffu, tl_min = ~0u, tl_max = 0, tl_sum = 0;
            for (int b = 0; b < 512; ++b) {
                st += hc[8 * b]; sr += hc[8 * b + 1]; mx = std::max(mx, (double)hc[8 * b + 1]); t_min = std::min(t_min, hc[8 * b + 2]);
                tl_min = std::min(tl_min, hc[8 * b + 6]); tl_max = std::max(tl_max, hc[8 * b + 6]); tl_sum += hc[8 * b + 6];
            }
            printf("probe: workgroup mean %.1f us, longest %.1f us, shader clock %.0f MHz, tiles per workgroup %u..%u (sum %u of %u)\n",
                   sr / 512 / 100, mx / 100, st / sr * 100, tl_min, tl_max, tl_sum, nt);
            if (rep == nrep - 1 && getenv("MB_TIMELINE"))
                for (int b = 0; b < 512; ++b)
                    printf("wg %3d start %7.2f end %7.2f us  hw_id %08x xcc %x  cu %2u sh %u se %u tiles %u\n", b, (hc[8 * b + 2] - t_min) / 100.0, (hc[8 * b + 3] - t_min) / 100.0,
                           hc[8 * b + 4], hc[8 * b + 5], (hc[8 * b + 4] >> 8) & 15u, (hc[8 * b + 4] >> 12) & 1u, (hc[8 * b + 4] >> 13) & 7u, hc[8 * b + 6]);
        }
    }
#endif
    if (argc > 2) {   // sustained run of the full kernel: burst vs throttled clock
        auto &v = vs[16];   // PRODUCT f32 hist slab64
        for (int w = 0; w < atoi(argv[2]); ++w) {
            CK(hipEventRecord(e0, s)); for (int k = 0; k < 256; ++k) v.run(k); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("sustained window %2d: %.1f us per launch\n", w, ms * 1e3 / 256);
        }
    }
    if (const char *only = getenv("MB_ONLY")) {          // keep variants whose name contains one of the '|'-separated substrings
        std::vector<std::string> keys; std::string cur;
        for (const char *c = only;; ++c) { if (*c == '|' || !*c) { if (!cur.empty()) keys.push_back(cur); cur.clear(); if (!*c) break; } else cur += *c; }
        std::vector<Variant> keep;
        for (auto &v : vs) for (auto &k : keys) if (v.name.find(k) != std::string::npos) { keep.push_back(v); break; }
        vs.swap(keep);
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
