// valu_bench.hip -- how many cycles does one wave64 VALU instruction of each kind occupy a SIMD of MI355X?  (tuning tool)
//
// The traversal kernel executes ~6000 vector instructions per ray and no arithmetic the matrix cores could take, so its
// ceiling is the issue rate of the vector ALU for ITS instruction mix: compares, selects, min / max, byte-to-float
// conversions, bit-field ops, fma.  The data sheet's 157 TFLOP/s is the rate of packed fp32 fma (v_pk_fma_f32, two floats
// per lane); this tool measures what the other instructions cost.  Every kernel runs `iters` x 64 independent-enough
// instructions (8 accumulators, so a wave never waits for its own result) on W waves per SIMD of every CU and reports
// cycles per wave-instruction per SIMD (s_memtime ticks, i.e. independent of the clock the chip happens to run at) and
// the chip-wide rate from the wall clock.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o /tmp/valu_bench && /tmp/valu_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                                        \
    do {                                                                                                                \
        hipError_t e = (x);                                                                                             \
        if (e != hipSuccess) {                                                                                          \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                                                        \
            return 1;                                                                                                   \
        }                                                                                                               \
    } while (0)

#define BODY8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BODY64(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I)
#define ACC32 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_MAX(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_CMP_S(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CNDMASK_S(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_CVT_UB(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define I_CVT_U32(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_LSHR(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define I_AND_OR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_LSHL_OR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %9\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_ADD_U32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_MUL_LO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_MAD_U24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n"
#define I_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_PK_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PK_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_PK_ADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_MIN(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define I_SUB(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define I_MAX_I32(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_MIN_U32(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_LSHL_ADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %9\n"
#define I_CMP_U32(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I_ADD_CO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define I_ADDC_CO(n) "v_addc_co_u32 %" #n ", vcc, %" #n ", %8, vcc\n"
#define I_CNDMASK_E64_VCC(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
// a select as compiled code has it: the compare that writes vcc, then the select that reads it
#define I_CMP_CNDMASK(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define I_CMP_CNDMASK_S(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, s[20:21]\n"
// one compare feeding four selects (a compare-and-swap of the sorting network)
#define I_CMP_4CND(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
// compare-and-swap of (key, payload) pairs, three ways: as compiled (one vcc, four selects), the mask in an SGPR pair, min / max for the keys
#define I_CSWAP_VCC(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CSWAP_SGPR(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8\n s_nop 1\n v_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n v_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_CSWAP_MINMAX(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n v_min_f32 %" #n ", %" #n ", %9\n v_max_f32 %" #n ", %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_MAX_MIN(n) "v_max_f32 %" #n ", %" #n ", %8\n v_min_f32 %" #n ", %" #n ", %9\n"
#define I_LSHL_ADD_U64(n) "v_lshl_add_u64 %" #n ", %" #n ", 4, %8\n"
// the traversal's plane evaluation as the compiler emits it: byte -> float, then fma
#define I_CVT_FMA(n) "v_cvt_f32_ubyte2 %" #n ", %8\n v_fma_f32 %" #n ", %" #n ", %8, %9\n"

enum Op {
    FMA, FMAC, MUL, ADD, MAX, MAX3, MED3, CMP, CMP_S, CNDMASK, CNDMASK_S, CVT_UB, CVT_U32, AND, LSHR, BFE, AND_OR, LSHL_OR, PERM, MOV, ADD_U32, MUL_LO, MAD_U24,
    RCP, SQRT, BCNT, DPP, PK_FMA, PK_MUL, PK_ADD, CVT_FMA, MIN, SUB, XOR, OR, MAX_I32, MIN_U32, ADD3, LSHL_ADD, CMP_U32, ADD_CO, ADDC_CO,
    CNDMASK_E64_VCC, CMP_CNDMASK, CMP_CNDMASK_S, CMP_4CND, MAX_MIN, LSHL_ADD_U64, CSWAP_VCC, CSWAP_SGPR, CSWAP_MINMAX, N_OPS
};
static const char *OP_NAME[N_OPS] = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_max3_f32", "v_med3_f32", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 sgpr",
                                     "v_cndmask_b32 vcc", "v_cndmask_b32 sgpr", "v_cvt_f32_ubyte1", "v_cvt_f32_u32", "v_and_b32", "v_lshrrev_b32", "v_bfe_u32",
                                     "v_and_or_b32", "v_lshl_or_b32", "v_perm_b32", "v_mov_b32", "v_add_u32", "v_mul_lo_u32", "v_mad_u32_u24", "v_rcp_f32",
                                     "v_sqrt_f32", "v_bcnt_u32_b32", "v_mov_b32 dpp", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "cvt_ubyte + fma (pair)", "v_min_f32", "v_sub_f32", "v_xor_b32", "v_or_b32", "v_max_i32", "v_min_u32",
                                     "v_add3_u32", "v_lshl_add_u32", "v_cmp_lt_u32 vcc", "v_add_co_u32", "v_addc_co_u32", "v_cndmask_b32_e64 vcc", "cmp vcc + cndmask (pair)",
                                     "cmp sgpr + cndmask (pair)", "cmp vcc + 4 cndmask (5)", "max + min (pair)", "v_lshl_add_u64", "cswap: vcc + nop + 4 cnd (5)", "cswap: sgpr + nop + 4 cnd(5)", "cswap: cmp min max 2 cnd (5)"};
static const int OP_INSTRS[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 5, 2, 1, 5, 5, 5};

template <int OP> __global__ __launch_bounds__(64) void k_valu(float *out, const int iters, long long *cycles, const float fb, const float fc) {
    float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    float b = fb + threadIdx.x * 1.0e-9f, c = fc;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = b, dc = c;
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#define RUN(I) asm volatile(BODY64(I) : ACC32 : "v"(b), "v"(c) : "vcc", "s20", "s21");
#define RUN64(I) asm volatile(BODY64(I) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc));
        if constexpr (OP == FMA) { RUN(I_FMA) }
        if constexpr (OP == FMAC) { RUN(I_FMAC) }
        if constexpr (OP == MUL) { RUN(I_MUL) }
        if constexpr (OP == ADD) { RUN(I_ADD) }
        if constexpr (OP == MAX) { RUN(I_MAX) }
        if constexpr (OP == MAX3) { RUN(I_MAX3) }
        if constexpr (OP == MED3) { RUN(I_MED3) }
        if constexpr (OP == CMP) { RUN(I_CMP) }
        if constexpr (OP == CMP_S) { RUN(I_CMP_S) }
        if constexpr (OP == CNDMASK) { RUN(I_CNDMASK) }
        if constexpr (OP == CNDMASK_S) { RUN(I_CNDMASK_S) }
        if constexpr (OP == CVT_UB) { RUN(I_CVT_UB) }
        if constexpr (OP == CVT_U32) { RUN(I_CVT_U32) }
        if constexpr (OP == AND) { RUN(I_AND) }
        if constexpr (OP == LSHR) { RUN(I_LSHR) }
        if constexpr (OP == BFE) { RUN(I_BFE) }
        if constexpr (OP == AND_OR) { RUN(I_AND_OR) }
        if constexpr (OP == LSHL_OR) { RUN(I_LSHL_OR) }
        if constexpr (OP == PERM) { RUN(I_PERM) }
        if constexpr (OP == MOV) { RUN(I_MOV) }
        if constexpr (OP == ADD_U32) { RUN(I_ADD_U32) }
        if constexpr (OP == MUL_LO) { RUN(I_MUL_LO) }
        if constexpr (OP == MAD_U24) { RUN(I_MAD_U24) }
        if constexpr (OP == RCP) { RUN(I_RCP) }
        if constexpr (OP == SQRT) { RUN(I_SQRT) }
        if constexpr (OP == BCNT) { RUN(I_BCNT) }
        if constexpr (OP == DPP) { RUN(I_DPP) }
        if constexpr (OP == PK_FMA) { RUN64(I_PK_FMA) }
        if constexpr (OP == PK_MUL) { RUN64(I_PK_MUL) }
        if constexpr (OP == PK_ADD) { RUN64(I_PK_ADD) }
        if constexpr (OP == CVT_FMA) { RUN(I_CVT_FMA) }
        if constexpr (OP == MIN) { RUN(I_MIN) }
        if constexpr (OP == SUB) { RUN(I_SUB) }
        if constexpr (OP == XOR) { RUN(I_XOR) }
        if constexpr (OP == OR) { RUN(I_OR) }
        if constexpr (OP == MAX_I32) { RUN(I_MAX_I32) }
        if constexpr (OP == MIN_U32) { RUN(I_MIN_U32) }
        if constexpr (OP == ADD3) { RUN(I_ADD3) }
        if constexpr (OP == LSHL_ADD) { RUN(I_LSHL_ADD) }
        if constexpr (OP == CMP_U32) { RUN(I_CMP_U32) }
        if constexpr (OP == ADD_CO) { RUN(I_ADD_CO) }
        if constexpr (OP == ADDC_CO) { RUN(I_ADDC_CO) }
        if constexpr (OP == CNDMASK_E64_VCC) { RUN(I_CNDMASK_E64_VCC) }
        if constexpr (OP == CMP_CNDMASK) { RUN(I_CMP_CNDMASK) }
        if constexpr (OP == CMP_CNDMASK_S) { RUN(I_CMP_CNDMASK_S) }
        if constexpr (OP == CMP_4CND) { RUN(I_CMP_4CND) }
        if constexpr (OP == MAX_MIN) { RUN(I_MAX_MIN) }
        if constexpr (OP == LSHL_ADD_U64) { RUN64(I_LSHL_ADD_U64) }
        if constexpr (OP == CSWAP_VCC) { RUN(I_CSWAP_VCC) }
        if constexpr (OP == CSWAP_SGPR) { RUN(I_CSWAP_SGPR) }
        if constexpr (OP == CSWAP_MINMAX) { RUN(I_CSWAP_MINMAX) }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + float(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (threadIdx.x == 0) {
        cycles[blockIdx.x] = t1 - t0;
    }
}

template <int OP> static int run_op(float *out, long long *cycles, const int waves_per_simd, const int iters, double &cyc_per_instr, double &ginstr_s) {
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k_valu<OP><<<blocks, 64>>>(out, iters, cycles, 1.0000001f, 1.0e-30f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        k_valu<OP><<<blocks, 64>>>(out, iters, cycles, 1.0000001f, 1.0e-30f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cycles, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double sum = 0.0;
    for (long long v : h) {
        sum += double(v);
    }
    const double instrs_per_wave = double(iters) * 64.0 * OP_INSTRS[OP];
    // a SIMD runs `waves_per_simd` waves concurrently: SIMD cycles per wave-instruction = wave's elapsed cycles / (its instructions x waves sharing the SIMD)
    cyc_per_instr = (sum / blocks) / (instrs_per_wave * waves_per_simd);
    ginstr_s = instrs_per_wave * blocks / (best * 1e-3) * 1e-9;
    return 0;
}

template <int OP> static int run_all(float *out, long long *cycles) {
    if constexpr (OP < N_OPS) {
        printf("%-24s", OP_NAME[OP]);
        for (int w : {1, 2, 4, 8}) {
            double c, g;
            if (run_op<OP>(out, cycles, w, 2048, c, g)) {
                return 1;
            }
            printf(" | %dw %6.0f G/s %4.2f cyc", w, g, 1024.0 * 2.4 / g);
            (void)c;
        }
        printf("\n");
        return run_all<OP + 1>(out, cycles);
    }
    return 0;
}

int main() {
    float *out;
    long long *cycles;
    CHECK(hipMalloc(&out, size_t(256) * 4 * 8 * 64 * 4));
    CHECK(hipMalloc(&cycles, size_t(256) * 4 * 8 * 8));
    printf("# wave64 instructions per second over the whole chip (G = 1e9) with 1 / 2 / 4 / 8 waves per SIMD, and the SIMD cycles per instruction that\n");
    printf("# rate means at 2.4 GHz x 1024 SIMDs (2 cycles = 32 lanes per clock, 4 cycles = 16 lanes per clock)\n");
    return run_all<0>(out, cycles);
}
