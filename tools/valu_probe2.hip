// tools/valu_probe2.hip -- VALU issue rate of scalar f32 ops by operand kind (VGPR / SGPR / literal / inline constant).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -o tools/valu_probe2 tools/valu_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// KIND: 0 fma v,v,v   1 fma s,v,v (one SGPR)   2 fmaak (literal addend)   3 fma v,v,1.0 (inline const)
//       4 mul v,v     5 mul s,v                6 mul literal,v            7 add v,v   8 add s,v   9 fmac v,v (VOP2, 4 bytes)
//      10 fma v,s,s (same SGPR twice)         11 max v,v                12 cvt_f32_u32        13 mul_f32 then add (2 ops all vgpr)
template <int KIND, int ILP>
__global__ void __launch_bounds__(1024) k(float *out, float a, float b, int iters)
{
    float av = a, bv = b;
    asm volatile("" : "+v"(av), "+v"(bv));
    unsigned long long mask64 = 0x5555555555555555ull; asm volatile("" : "+s"(mask64));
    float x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                float &v = x[i];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(v) : "s"(a), "v"(bv));
                if (KIND == 2) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(v) : "v"(av));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(v) : "v"(av));
                if (KIND == 4) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(v) : "v"(av));
                if (KIND == 5) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(v) : "s"(a));
                if (KIND == 6) asm volatile("v_mul_f32_e32 %0, 0x3f7fbe77, %0" : "+v"(v));
                if (KIND == 7) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(v) : "v"(bv));
                if (KIND == 8) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(v) : "s"(b));
                if (KIND == 9) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "s"(a));
                if (KIND == 11) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(v) : "v"(bv));
                if (KIND == 12) asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(v));
                if (KIND == 13) asm volatile("v_mul_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %2" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 14) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(v) : "v"(bv));
                if (KIND == 15) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v) : "v"(bv));
                if (KIND == 16) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v) : "v"(bv));
                if (KIND == 17) asm volatile("v_rndne_f32_e32 %0, %0" : "+v"(v));
                if (KIND == 18) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v) : "v"(bv), "s"(mask64));
                if (KIND == 19) asm volatile("v_cmp_le_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, 0, %0, vcc" : "+v"(v) : "v"(bv) : "vcc");
                if (KIND == 20) asm volatile("v_cmp_le_f32_e32 vcc, %1, %0" : : "v"(v), "v"(bv) : "vcc");
                if (KIND == 21) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(v) : "v"(bv));
                if (KIND == 22) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 23) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 24) asm volatile("v_and_b32_e32 %0, 0x7fffff, %0" : "+v"(v));
                if (KIND == 25) asm volatile("v_add_u32_e32 %0, 0x3f3504f3, %0" : "+v"(v));
                if (KIND == 26) asm volatile("v_ashrrev_i32_e32 %0, 23, %0" : "+v"(v));
                // round 3: the forms the narrow-surface kernels now use (tools/isa_budget.py's price list)
                if (KIND == 27) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v));
                if (KIND == 28) asm volatile("v_log_f32_e32 %0, %0" : "+v"(v));
                if (KIND == 29) asm volatile("v_exp_f32_e64 %0, %0 clamp" : "+v"(v));
                if (KIND == 30) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v) : "v"(bv), "s"(a));
                if (KIND == 31) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(v) : "v"(bv));
                if (KIND == 32) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(v) : "s"(a));
                if (KIND == 33) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v) : "v"(bv));
                if (KIND == 34) asm volatile("v_cvt_u32_f32_e32 %0, %0" : "+v"(v));
                if (KIND == 35) asm volatile("v_fract_f32_e32 %0, %0" : "+v"(v));
                if (KIND == 36) asm volatile("v_cmp_gt_f32_e64 %1, |%0|, %2" : "+v"(v), "=s"(mask64) : "v"(bv));
                if (KIND == 37) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 38) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 39) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(v) : "v"(bv));
                if (KIND == 40) asm volatile("v_lshlrev_b32_e32 %0, 5, %0" : "+v"(v));
                if (KIND == 41) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 42) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(v));
                if (KIND == 43) asm volatile("v_exp_f32_e32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(v), "+v"(x[(i + 1) % ILP]) : "v"(av), "v"(bv));
                if (KIND == 44) asm volatile("v_fma_f32 %0, |%0|, %1, %2 clamp" : "+v"(v) : "v"(av), "v"(bv));
                if (KIND == 45) asm volatile("v_sub_u32_sdwa %0, %0, %1 clamp dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(v) : "v"(bv));
                if (KIND == 46) asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0" : : "v"(v), "v"(bv) : "vcc");
                if (KIND == 47) asm volatile("v_cvt_f16_f32_e32 %0, %0" : "+v"(v));
                // round 4: the shifts / field extracts around the LDS threshold table (rd_q8_lut_bits, rd_hist_add_b2)
                if (KIND == 55) asm volatile("v_lshrrev_b32_e32 %0, 14, %0" : "+v"(v));
                if (KIND == 56) asm volatile("v_bfe_u32 %0, %0, 16, 8" : "+v"(v));
                if (KIND == 57) asm volatile("v_lshl_or_b32 %0, %0, 5, %1" : "+v"(v) : "v"(bv));
                if (KIND == 58) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(v) : "v"(bv));
                if (KIND == 59) asm volatile("v_ashrrev_i32_e32 %0, 16, %0\n\ts_nop 0" : "+v"(v));
                if (KIND == 60) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(v));
                if (KIND == 61) asm volatile("v_lshrrev_b32_e32 %0, 16, %0\n\tv_lshl_add_u32 %0, %0, 5, %1" : "+v"(v) : "v"(bv));
                if (KIND == 62) asm volatile("v_mul_f32_e32 %0, 0x0f800000, %0" : "+v"(v));
                // packed f32 (two lane-ops per instruction): plain, with op_sel broadcasts, with an inline constant
                if (KIND >= 48 && KIND <= 54) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 &p = *reinterpret_cast<f2 *>(&x[i & ~1]);
                    f2 ua = { av, bv }, ub = { bv, av };
                    if (KIND == 48) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(ua));
                    if (KIND == 49) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p) : "v"(ua));
                    if (KIND == 50) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(p) : "v"(ua));
                    if (KIND == 51) asm volatile("v_pk_add_f32 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(p));
                    if (KIND == 52) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(ua), "v"(ub));
                    if (KIND == 53) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(p) : "v"(ua), "v"(ub));
                    if (KIND == 54) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(p) : "v"(ua), "v"(ub));
                }
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    if (s == 1234.5678f) out[0] = s;
}

template <int KIND, int ILP>
static void run(const char *name, float *out)
{
    const int iters = 512, blocks = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(1024), 0, 0, out, 0.999f, 0.001f, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(1024), 0, 0, out, 0.999f, 0.001f, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
    const double instr = (double)blocks * 16 * ILP * 4 * iters * (KIND == 13 || KIND == 61 ? 2 : KIND == 43 ? 4 : 1);     // wave-instructions (s_nop not counted)
    printf("%-40s ILP %2d: %8.1f us  %6.2f ns per wave-instr per SIMD  (%5.2f T lane-op/s)\n", name, ILP, ms * 1e3,
           ms * 1e6 / (instr / 1024), instr * 64 / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    float *out; CK(hipMalloc((void **)&out, 64));
#define R(K, N) run<K, 8>(N, out); run<K, 2>(N, out)
    R(26, "v_ashrrev_i32"); R(55, "v_lshrrev_b32 14"); R(60, "v_lshlrev_b32 1"); R(56, "v_bfe_u32"); R(57, "v_lshl_or_b32"); R(58, "v_add_u32 v,v");
    R(59, "v_ashrrev_i32 + s_nop 0 (per v instr)"); R(61, "v_lshrrev_b32 + v_lshl_add_u32 (per instr, 2)"); R(62, "v_mul_f32 literal 2^-96 (denormal results)");
    if (argc > 1 && !strcmp(argv[1], "new")) return 0;
    R(0, "v_fma_f32 v,v,v"); R(1, "v_fma_f32 s,v,v"); R(2, "v_fmaak_f32 v,v,literal"); R(14, "v_fmamk_f32 v,literal,v");
    R(3, "v_fma_f32 v,v,1.0 (inline)"); R(10, "v_fma_f32 v,s,s");
    R(9, "v_fmac_f32_e32 v,v"); R(4, "v_mul_f32_e32 v,v"); R(5, "v_mul_f32_e32 s,v"); R(6, "v_mul_f32_e32 literal,v");
    R(7, "v_add_f32_e32 v,v"); R(8, "v_add_f32_e32 s,v"); R(13, "v_mul v,v + v_add v,v");
    R(11, "v_max_f32_e32 v,v"); R(12, "v_cvt_f32_u32_e32"); R(15, "v_cndmask_b32_e32 (vcc never written)"); R(16, "v_lshl_add_u32 (VOP3)"); R(17, "v_rndne_f32_e32");
    R(18, "v_cndmask_b32_e64 v,v,s[pair]"); R(19, "v_cmp_le + v_cndmask (2 instr, per pair)"); R(20, "v_cmp_le_f32 vcc"); R(21, "v_min_u32_e32");
    R(22, "v_min3_f32"); R(23, "v_div_fixup_f32"); R(24, "v_and_b32 literal"); R(25, "v_add_u32 literal"); R(26, "v_ashrrev_i32");
    R(27, "v_exp_f32"); R(28, "v_log_f32"); R(29, "v_exp_f32_e64 clamp"); R(43, "v_exp_f32 + 3 v_fma_f32 (per instr, 4)");
    R(30, "v_perm_b32 v,v,s"); R(31, "v_or_b32_sdwa v,v (WORD_1)"); R(32, "v_or_b32_sdwa v,s (WORD_1)"); R(42, "v_cvt_f32_u32_sdwa (WORD_1)");
    R(45, "v_sub_u32_sdwa v,v clamp"); R(33, "v_cvt_pk_f16_f32"); R(47, "v_cvt_f16_f32"); R(34, "v_cvt_u32_f32"); R(35, "v_fract_f32");
    R(36, "v_cmp_gt_f32_e64 s[pair],|v|,v"); R(46, "v_cmp_gt_u32_e32 vcc"); R(37, "v_max3_f32"); R(38, "v_med3_f32"); R(39, "v_mov_b32");
    R(40, "v_lshlrev_b32"); R(41, "v_and_or_b32"); R(44, "v_fma_f32 |v|,v,v clamp");
    R(48, "v_pk_mul_f32 (per instr = 2 lane-ops)"); R(49, "v_pk_mul_f32 op_sel_hi:[1,0]"); R(50, "v_pk_mul_f32 op_sel:[0,1]");
    R(51, "v_pk_add_f32 v, 1.0 op_sel_hi:[1,0]"); R(52, "v_pk_fma_f32"); R(53, "v_pk_fma_f32 op_sel:[0,1,0]"); R(54, "v_pk_fma_f32 op_sel_hi:[0,1,1] neg");
    return 0;
}
