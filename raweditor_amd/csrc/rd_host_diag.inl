// rd_host_diag.inl -- part of librawdev.so's host side: included by rawdev.hip (one translation unit; the kernels are
// templates in rd_kernels.h).  Exhaustive self-tests of the narrow surfaces' gamma shortcuts and the measurement aids behind bench.py's
// box ceilings (rd_measure_hbm, rd_measure_valu).  Nothing here is on a render path.

// ------------------------------------------------------------------------------------------------
// self-test of the 8-bit surfaces' gamma shortcut (rd_kernels.h, rd_q8_gamma) over every float encoding
// ------------------------------------------------------------------------------------------------
struct rd_q8_stats { unsigned long long mismatches, fallbacks; uint32_t first_bad, max_dist_bits; };

__global__ void __launch_bounds__(256) rd_q8_sweep(uint32_t base, rd_q8_stats *st, uint8_t *codes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = rd_u2f(base + i);
    const uint32_t fast = rd_q8_gamma(x);
    const uint32_t exact = rd_q8(rd_gamma_clamp(x));
    if (codes) codes[i] = (uint8_t)fast;
    if (!st) return;
    if (fast != exact || fast > 255u) { atomicAdd(&st->mismatches, 1ull); atomicMin(&st->first_bad, base + i); }
    if (x >= RD_FLT_MIN) {                                       // diagnostics: how often the pinned evaluation decides, and how far
        float z;                                                 // the hardware 255 e is from the pinned 255 g (in codes)
        const float e = rd_hw_gamma01(x, z);
        const float t = __builtin_fmaf(e, 255.0f, RD_MAGIC23);
        const float dn = __builtin_fmaf(e, 255.0f, -(t - RD_MAGIC23));
        if (__builtin_fabsf(dn) > 0.5f - RD_Q8_EPS) atomicAdd(&st->fallbacks, 1ull);
        const float d = __builtin_fabsf(e * 255.0f - rd_gamma_clamp(x) * 255.0f);
        atomicMax(&st->max_dist_bits, rd_f2u(d));               // d >= 0: integer order == float order
    }
}

extern "C" int rd_selftest_q8(int device, uint64_t *mismatches, uint32_t *first_bad, uint64_t *fallbacks, float *max_dist) try
{
    RD_ENTRY(rd_selftest_q8);
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    rd_q8_stats *dst = nullptr, st = { 0, 0, 0xffffffffu, 0 };
    RD_HIP(hipMalloc((void **)&dst, sizeof st));
    hipError_t e = hipMemcpy(dst, &st, sizeof st, hipMemcpyHostToDevice);
    for (uint32_t c = 0; c < 256u && e == hipSuccess; ++c) {    // 256 launches x 2^24 encodings
        hipLaunchKernelGGL(rd_q8_sweep, dim3(1u << 16), dim3(256), 0, 0, c << 24, dst, (uint8_t *)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&st, dst, sizeof st, hipMemcpyDeviceToHost);
    (void)hipFree(dst);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_selftest_q8: %s", hipGetErrorString(e));
    if (mismatches) *mismatches = st.mismatches;
    if (first_bad) *first_bad = st.first_bad;
    if (fallbacks) *fallbacks = st.fallbacks;
    if (max_dist) *max_dist = rd_u2f(st.max_dist_bits);
    return RD_OK;
}
RD_CATCH_INT(rd_selftest_q8)

// The export kernel's threshold table (rd_q8_lut_bits) against the pinned evaluation, same sweep: the table in LDS, as there.
__global__ void __launch_bounds__(256) rd_q8_lut_sweep(uint32_t base, rd_q8_stats *st, uint8_t *codes)
{
    __shared__ __attribute__((aligned(16))) uint32_t lut[RD_Q8_LUT_LDS_WORDS];
    rd_q8_lut_load(lut);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = rd_u2f(base + i);
    const uint32_t sbits = rd_q8_lut_bits(x, lut);
    const uint32_t fast = sbits >> 16;                           // bits 24..31 must be zero: compared as a whole
    const uint32_t exact = rd_q8(rd_gamma_clamp(x));
    if (codes) codes[i] = (uint8_t)fast;
    if (!st) return;
    if (fast != exact) { atomicAdd(&st->mismatches, 1ull); atomicMin(&st->first_bad, base + i); }
}

extern "C" int rd_selftest_q8_lut(int device, uint64_t *mismatches, uint32_t *first_bad) try
{
    RD_ENTRY(rd_selftest_q8_lut);
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    rc = rd_q8_lut_ensure(device);
    if (rc) return rc;
    rd_q8_stats *dst = nullptr, st = { 0, 0, 0xffffffffu, 0 };
    RD_HIP(hipMalloc((void **)&dst, sizeof st));
    hipError_t e = hipMemcpy(dst, &st, sizeof st, hipMemcpyHostToDevice);
    for (uint32_t c = 0; c < 256u && e == hipSuccess; ++c) {    // 256 launches x 2^24 encodings
        hipLaunchKernelGGL(rd_q8_lut_sweep, dim3(1u << 16), dim3(256), 0, 0, c << 24, dst, (uint8_t *)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&st, dst, sizeof st, hipMemcpyDeviceToHost);
    (void)hipFree(dst);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_selftest_q8_lut: %s", hipGetErrorString(e));
    if (mismatches) *mismatches = st.mismatches;
    if (first_bad) *first_bad = st.first_bad;
    return RD_OK;
}
RD_CATCH_INT(rd_selftest_q8_lut)

extern "C" int rd_selftest_q8_lut_codes(int device, uint32_t first_encoding, uint32_t n, uint8_t *dst) try
{
    RD_ENTRY(rd_selftest_q8_lut_codes);
    if (!dst || !n || (n & 255u) || (uint64_t)first_encoding + n > (1ull << 32))
        return rd_fail(RD_ERR_INVALID_ARG, "rd_selftest_q8_lut_codes: n must be a non-zero multiple of 256 inside the 2^32 encodings");
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    rc = rd_q8_lut_ensure(device);
    if (rc) return rc;
    uint8_t *dev = nullptr;
    RD_HIP(hipMalloc((void **)&dev, n));
    hipLaunchKernelGGL(rd_q8_lut_sweep, dim3(n / 256u), dim3(256), 0, 0, first_encoding, (rd_q8_stats *)nullptr, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(dst, dev, n, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_selftest_q8_lut_codes: %s", hipGetErrorString(e));
    return RD_OK;
}
RD_CATCH_INT(rd_selftest_q8_lut_codes)

// No device needed: the table itself (RD_Q8_LUT_WORDS words), for host-side checks of its construction.
extern "C" int rd_q8_lut_table(uint32_t *dst, size_t cap_words) try
{
    RD_ENTRY(rd_q8_lut_table);
    if (!dst || cap_words < RD_Q8_LUT_WORDS) return rd_fail(RD_ERR_INVALID_ARG, "rd_q8_lut_table: need room for %u words", RD_Q8_LUT_WORDS);
    rd_q8_lut_build(dst);
    return (int)RD_Q8_LUT_WORDS;
}
RD_CATCH_INT(rd_q8_lut_table)

// The same for the binary16 surface's shortcut (rd_f16_gamma): halves and, with the histogram, codes.
__global__ void __launch_bounds__(256) rd_f16_sweep(uint32_t base, rd_q8_stats *st, uint16_t *halves)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = rd_u2f(base + i);
    uint32_t h, q, h2, q2;
    rd_f16_gamma<true>(x, h, q);
    rd_f16_gamma<false>(x, h2, q2);                          // the variant without the histogram must give the same half
    const float g = rd_gamma_clamp(x);
    const uint32_t he = __builtin_bit_cast(uint16_t, (_Float16)g), qe = rd_q8(g);
    if (halves) halves[i] = (uint16_t)h;
    if (!st) return;
    if (h != he || q != qe || h2 != he) { atomicAdd(&st->mismatches, 1ull); atomicMin(&st->first_bad, base + i); }
    if (x >= RD_FLT_MIN) {                                    // how often the pinned evaluation decides the half
        const float z = __builtin_amdgcn_logf(x) * RD_INV_GAMMA;
        const float e = __builtin_fminf(__builtin_amdgcn_exp2f(z), 1.0f);
        const float lowbits = rd_u2f((rd_f2u(e) & 0x1fffu) | 0x4b000000u) - 8388608.0f;
        const float k = __builtin_fmaf(__builtin_fabsf(z), RD_F16_KA, RD_F16_KB);
        if (e >= 6.103515625e-05f && __builtin_fabsf(lowbits - 4096.0f) <= k) atomicAdd(&st->fallbacks, 1ull);   // normal halves only
    }
}

extern "C" int rd_selftest_f16(int device, uint64_t *mismatches, uint32_t *first_bad, uint64_t *fallbacks) try
{
    RD_ENTRY(rd_selftest_f16);
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    rd_q8_stats *dst = nullptr, st = { 0, 0, 0xffffffffu, 0 };
    RD_HIP(hipMalloc((void **)&dst, sizeof st));
    hipError_t e = hipMemcpy(dst, &st, sizeof st, hipMemcpyHostToDevice);
    for (uint32_t c = 0; c < 256u && e == hipSuccess; ++c) {
        hipLaunchKernelGGL(rd_f16_sweep, dim3(1u << 16), dim3(256), 0, 0, c << 24, dst, (uint16_t *)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&st, dst, sizeof st, hipMemcpyDeviceToHost);
    (void)hipFree(dst);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_selftest_f16: %s", hipGetErrorString(e));
    if (mismatches) *mismatches = st.mismatches;
    if (first_bad) *first_bad = st.first_bad;
    if (fallbacks) *fallbacks = st.fallbacks;
    return RD_OK;
}
RD_CATCH_INT(rd_selftest_f16)

extern "C" int rd_selftest_f16_halves(int device, uint32_t first_encoding, uint32_t n, uint16_t *dst) try
{
    RD_ENTRY(rd_selftest_f16_halves);
    if (!dst || !n || (n & 255u) || (uint64_t)first_encoding + n > (1ull << 32))
        return rd_fail(RD_ERR_INVALID_ARG, "rd_selftest_f16_halves: n must be a non-zero multiple of 256 inside the 2^32 encodings");
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    uint16_t *dev = nullptr;
    RD_HIP(hipMalloc((void **)&dev, (size_t)n * 2));
    hipLaunchKernelGGL(rd_f16_sweep, dim3(n / 256u), dim3(256), 0, 0, first_encoding, (rd_q8_stats *)nullptr, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(dst, dev, (size_t)n * 2, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_selftest_f16_halves: %s", hipGetErrorString(e));
    return RD_OK;
}
RD_CATCH_INT(rd_selftest_f16_halves)

// The export kernel's two-level tables for the f16 surface (rd_f16_lut_lookup + the pinned evaluation for the lanes it sends
// there) against the pinned function, same sweep: tables in LDS, as there.  dst (optional): (half | code << 16) per encoding.
__global__ void __launch_bounds__(256) rd_f16_lut_sweep(uint32_t base, rd_q8_stats *st, uint32_t *dst)
{
    __shared__ __attribute__((aligned(16))) uint16_t fine[RD_F16_LUT_FINE_LDS];
    __shared__ __attribute__((aligned(16))) uint32_t coarse[RD_F16_LUT_COARSE_LDS];
    rd_f16_lut_load(fine, coarse);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = rd_u2f(base + i);
    uint32_t h, s;
    bool pinned;
    float xc;
    rd_f16_lut_lookup(x, fine, coarse, h, s, pinned, xc);
    if (pinned) {
        const float e = rd_gamma_clamp(xc);
        h = __builtin_bit_cast(uint16_t, (_Float16)e);
        s = rd_q8(e) << 16;
    }
    const float g = rd_gamma_clamp(x);
    const uint32_t he = __builtin_bit_cast(uint16_t, (_Float16)g), qe = rd_q8(g);
    if (dst) dst[i] = (h & 0xffffu) | ((s >> 16) << 16);
    if (!st) return;
    if (h != he || (s >> 16) != qe) { atomicAdd(&st->mismatches, 1ull); atomicMin(&st->first_bad, base + i); }   // h, s >> 16 compared whole: no stray high bits
    if (pinned) atomicAdd(&st->fallbacks, 1ull);
}

extern "C" int rd_selftest_f16_lut(int device, uint64_t *mismatches, uint32_t *first_bad, uint64_t *pinned) try
{
    RD_ENTRY(rd_selftest_f16_lut);
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    rc = rd_f16_lut_ensure(device);
    if (rc) return rc;
    rd_q8_stats *dst = nullptr, st = { 0, 0, 0xffffffffu, 0 };
    RD_HIP(hipMalloc((void **)&dst, sizeof st));
    hipError_t e = hipMemcpy(dst, &st, sizeof st, hipMemcpyHostToDevice);
    for (uint32_t c = 0; c < 256u && e == hipSuccess; ++c) {    // 256 launches x 2^24 encodings
        hipLaunchKernelGGL(rd_f16_lut_sweep, dim3(1u << 16), dim3(256), 0, 0, c << 24, dst, (uint32_t *)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&st, dst, sizeof st, hipMemcpyDeviceToHost);
    (void)hipFree(dst);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_selftest_f16_lut: %s", hipGetErrorString(e));
    if (mismatches) *mismatches = st.mismatches;
    if (first_bad) *first_bad = st.first_bad;
    if (pinned) *pinned = st.fallbacks;
    return RD_OK;
}
RD_CATCH_INT(rd_selftest_f16_lut)

extern "C" int rd_selftest_f16_lut_values(int device, uint32_t first_encoding, uint32_t n, uint32_t *dst) try
{
    RD_ENTRY(rd_selftest_f16_lut_values);
    if (!dst || !n || (n & 255u) || (uint64_t)first_encoding + n > (1ull << 32))
        return rd_fail(RD_ERR_INVALID_ARG, "rd_selftest_f16_lut_values: n must be a non-zero multiple of 256 inside the 2^32 encodings");
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    rc = rd_f16_lut_ensure(device);
    if (rc) return rc;
    uint32_t *dev = nullptr;
    RD_HIP(hipMalloc((void **)&dev, (size_t)n * 4));
    hipLaunchKernelGGL(rd_f16_lut_sweep, dim3(n / 256u), dim3(256), 0, 0, first_encoding, (rd_q8_stats *)nullptr, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(dst, dev, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_selftest_f16_lut_values: %s", hipGetErrorString(e));
    return RD_OK;
}
RD_CATCH_INT(rd_selftest_f16_lut_values)

// No device needed: the two tables themselves (fine: RD_F16_LUT_NF u16; coarse: RD_F16_LUT_NC {E, C} pairs), for host-side checks
// of their construction.  Returns the builder's verdict (0: the shape the lookup assumes) or a negative status.
extern "C" int rd_f16_lut_tables(uint16_t *fine, size_t cap_fine, uint32_t *coarse, size_t cap_coarse_words) try
{
    RD_ENTRY(rd_f16_lut_tables);
    if (!fine || !coarse || cap_fine < RD_F16_LUT_NF + 1u || cap_coarse_words < RD_F16_LUT_NC * 2u)
        return rd_fail(RD_ERR_INVALID_ARG, "rd_f16_lut_tables: need room for %u u16 and %u words", RD_F16_LUT_NF + 1u, RD_F16_LUT_NC * 2u);
    const int rc = rd_f16_lut_build(fine, coarse);
    if (rc) return rd_fail(RD_ERR_UNSUPPORTED, "the binary16 threshold tables cannot be built from this gamma (check %d)", rc);
    return RD_OK;
}
RD_CATCH_INT(rd_f16_lut_tables)

extern "C" int rd_selftest_q8_codes(int device, uint32_t first_encoding, uint32_t n, uint8_t *dst) try
{
    RD_ENTRY(rd_selftest_q8_codes);
    if (!dst || !n || (n & 255u) || (uint64_t)first_encoding + n > (1ull << 32))
        return rd_fail(RD_ERR_INVALID_ARG, "rd_selftest_q8_codes: n must be a non-zero multiple of 256 inside the 2^32 encodings");
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    uint8_t *dev = nullptr;
    RD_HIP(hipMalloc((void **)&dev, n));
    hipLaunchKernelGGL(rd_q8_sweep, dim3(n / 256u), dim3(256), 0, 0, first_encoding, (rd_q8_stats *)nullptr, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(dst, dev, n, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_selftest_q8_codes: %s", hipGetErrorString(e));
    return RD_OK;
}
RD_CATCH_INT(rd_selftest_q8_codes)

// ------------------------------------------------------------------------------------------------
// measurement aid: the streaming ceilings of THIS device, now (bench.py: roofline.box_copy_GBps / box_fill_GBps)
// ------------------------------------------------------------------------------------------------
// Boxes of one pool differ by a few per cent (power-managed clocks), and SURVEY.md section 8d asks for the roofline
// fraction against a ceiling measured on the box, not only against the 8 TB/s of the data sheet.  Three trivial kernels,
// 16 B per lane and access, eight accesses in flight per lane; a WAVE walks its own contiguous range (1 KiB per
// instruction, 8 KiB per step).  That shape is the fastest of the ones tools/hbm_probe.hip tries (profiles/r04_hbm_probe.txt:
// copy 5.72 TB/s, fill 6.11 TB/s against 4.9 / 4.9 TB/s for a grid-stride loop with the same accesses in flight and
// 5.0 / 4.2 TB/s for the one-access grid-stride loop this function started with); hipMemsetAsync is reported beside them.
#define RD_PROBE_U 8
__global__ void __launch_bounds__(1024) rd_probe_copy(const rd_f4 *__restrict__ src, rd_f4 *__restrict__ dst, size_t per_wave)
{
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    for (size_t i = w * per_wave; i < (w + 1) * per_wave; i += 64u * RD_PROBE_U) {
        rd_f4 v[RD_PROBE_U];
#pragma unroll
        for (int k = 0; k < RD_PROBE_U; ++k) v[k] = src[i + (size_t)k * 64u + lane];
#pragma unroll
        for (int k = 0; k < RD_PROBE_U; ++k) __builtin_nontemporal_store(v[k], dst + i + (size_t)k * 64u + lane);
    }
}

__global__ void __launch_bounds__(1024) rd_probe_fill(rd_f4 *__restrict__ dst, size_t per_wave, float x)
{
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const rd_f4 v = { x, x + 1.0f, x + 2.0f, 1.0f };
    for (size_t i = w * per_wave; i < (w + 1) * per_wave; i += 64u * RD_PROBE_U) {
#pragma unroll
        for (int k = 0; k < RD_PROBE_U; ++k) __builtin_nontemporal_store(v, dst + i + (size_t)k * 64u + lane);
    }
}

__global__ void __launch_bounds__(1024) rd_probe_read(const rd_f4 *__restrict__ src, size_t per_wave, float *sink)
{
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    float acc = 0.0f;
    for (size_t i = w * per_wave; i < (w + 1) * per_wave; i += 64u * RD_PROBE_U) {
        rd_f4 v[RD_PROBE_U];
#pragma unroll
        for (int k = 0; k < RD_PROBE_U; ++k) v[k] = src[i + (size_t)k * 64u + lane];
#pragma unroll
        for (int k = 0; k < RD_PROBE_U; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 12345.678f) *sink = acc;                      // never true for the zeroed buffer: keeps the loads alive
}

extern "C" int rd_measure_hbm(int device, size_t bytes, uint32_t reps, double *copy_GBps, double *fill_GBps, double *read_GBps,
                              double *memset_GBps) try
{
    RD_ENTRY(rd_measure_hbm);
    if (!reps || reps > 64) return rd_fail(RD_ERR_INVALID_ARG, "rd_measure_hbm: need 1..64 repetitions");
    // copy: 4096 x 16 waves, fill / read: 2048 x 16 waves.  Every wave of EVERY grid must own a whole number of 8-KiB steps
    // (the kernels have no tail handling: a wave whose share is not a multiple of a step runs into its neighbour's range, the
    // last one past the allocation).  So the size is rounded down to whole steps of the LARGER grid -- 4096 x 16 x 8 KiB =
    // 512 MiB per buffer -- which gives the smaller grid's waves an even number of steps; anything below one such unit is refused.
    const uint32_t blocks[3] = { 4096u, 2048u, 2048u };
    const size_t step = 64u * RD_PROBE_U;                                        // float4 per wave and step
    const size_t unit = (size_t)blocks[0] * 16u * step;                           // float4 per step of the whole copy grid
    const size_t n = bytes / sizeof(rd_f4) / unit * unit;                         // float4 actually moved
    if (!n) return rd_fail(RD_ERR_INVALID_ARG, "rd_measure_hbm: need at least %zu MiB (one 8-KiB step for each of %u waves)",
                           (unit * sizeof(rd_f4)) >> 20, blocks[0] * 16u);
    int rc = rd_check_device(device, nullptr);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    struct res {                                                 // released on every path, an exception included
        void *a = nullptr, *b = nullptr;
        float *sink = nullptr;
        hipStream_t s = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~res()
        {
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            if (sink) (void)hipFree(sink);
        }
    } r_;
    void *&a = r_.a, *&b = r_.b;
    float *&sink = r_.sink;
    hipStream_t &s = r_.s;
    hipEvent_t &e0 = r_.e0, &e1 = r_.e1;
    std::vector<float> ms;
    ms.reserve(reps + 1u);
    hipError_t e = hipMalloc(&a, n * sizeof(rd_f4));
    if (e == hipSuccess) e = hipMalloc(&b, n * sizeof(rd_f4));
    if (e == hipSuccess) e = hipMalloc((void **)&sink, sizeof(float));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0, n * sizeof(rd_f4), s);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, n * sizeof(rd_f4), s);
    double out[4] = { 0.0, 0.0, 0.0, 0.0 };
    for (int which = 0; which < 4 && e == hipSuccess; ++which) {
        ms.clear();
        const size_t per_wave = which < 3 ? n / ((size_t)blocks[which] * 16u) : 0;          // a multiple of `step` for all three grids
        for (uint32_t r = 0; r < reps + 1u && e == hipSuccess; ++r) {             // the first launch warms up
            e = hipEventRecord(e0, s);
            if (which == 0) hipLaunchKernelGGL(rd_probe_copy, dim3(blocks[0]), dim3(1024), 0, s, (const rd_f4 *)a, (rd_f4 *)b, per_wave);
            else if (which == 1) hipLaunchKernelGGL(rd_probe_fill, dim3(blocks[1]), dim3(1024), 0, s, (rd_f4 *)b, per_wave, (float)r);
            else if (which == 2) hipLaunchKernelGGL(rd_probe_read, dim3(blocks[2]), dim3(1024), 0, s, (const rd_f4 *)a, per_wave, sink);
            else if (e == hipSuccess) e = hipMemsetAsync(b, 0, n * sizeof(rd_f4), s);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(e1, s);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float t = 0.0f;
            if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
            if (e == hipSuccess && r) ms.push_back(t);
        }
        if (e == hipSuccess) {
            std::sort(ms.begin(), ms.end());
            const double med = ms[ms.size() / 2];
            out[which] = (which == 0 ? 2.0 : 1.0) * (double)(n * sizeof(rd_f4)) / (med * 1e-3) / 1e9;
        }
    }
    if (e != hipSuccess) return rd_fail(e == hipErrorOutOfMemory ? RD_ERR_OOM : RD_ERR_HIP, "rd_measure_hbm: %s", hipGetErrorString(e));
    if (copy_GBps) *copy_GBps = out[0];
    if (fill_GBps) *fill_GBps = out[1];
    if (read_GBps) *read_GBps = out[2];
    if (memset_GBps) *memset_GBps = out[3];
    return RD_OK;
}
RD_CATCH_INT(rd_measure_hbm)

// What one full-rate VALU wave-instruction costs a SIMD on THIS device right now: 512 x 1024 threads (8 waves per SIMD, as
// the export kernel runs), eight independent chains per lane of alternating v_mul_f32 / v_add_f32 -- the two-operand forms
// the strict colour stack is made of (tools/valu_probe2.hip: 1.05 ns per instruction for this pair, 1.20 ns for the
// three-operand v_fma_f32; the cheaper one is the honest price for a LOWER bound on issue time).  bench.py prices the
// export kernels' static instruction budgets (profiles/isa_budget.json, in units of half such an instruction) with it.
__global__ void __launch_bounds__(1024) rd_probe_valu(float *out, float a, float b, int iters)
{
    float av = a, bv = b, x[8];
    asm volatile("" : "+v"(av), "+v"(bv));
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (r & 1) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(bv));
                else asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(av));
            }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 1234.5678f) out[0] = s;
}

extern "C" int rd_measure_valu(int device, double *ns_per_full_rate_instruction) try
{
    RD_ENTRY(rd_measure_valu);
    if (!ns_per_full_rate_instruction) return rd_fail(RD_ERR_INVALID_ARG, "rd_measure_valu: NULL argument");
    int n_cu = 0;
    int rc = rd_check_device(device, &n_cu);
    if (rc) return rc;
    rd_devguard g(device);
    if (!g.ok) return rd_fail(RD_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    const int iters = 512, blocks = 2 * n_cu;                                   // two 1024-thread workgroups per CU
    float *out = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&out, 64);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    std::vector<float> ms;
    for (int r = 0; r < 6 && e == hipSuccess; ++r) {
        e = hipEventRecord(e0, s);
        hipLaunchKernelGGL(rd_probe_valu, dim3(blocks), dim3(1024), 0, s, out, 0.999f, 0.001f, iters);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float t = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
        if (e == hipSuccess && r) ms.push_back(t);
    }
    if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (out) (void)hipFree(out);
    if (e != hipSuccess) return rd_fail(RD_ERR_HIP, "rd_measure_valu: %s", hipGetErrorString(e));
    std::sort(ms.begin(), ms.end());
    const double per_simd = (double)blocks * 16.0 * 8.0 * 4.0 * iters / ((double)n_cu * 4.0);     // wave-instructions each SIMD issued
    *ns_per_full_rate_instruction = ms.front() * 1e6 / per_simd;      // the fastest of five: the clock the part reaches under pure VALU load
    return RD_OK;
}
RD_CATCH_INT(rd_measure_valu)

