// vgpr_bank_bench.hip -- does the register bank of a VALU instruction's operands matter on gfx950?
//
// The vector register file is banked by register number (number mod 4).  This program issues one
// stream of independent v_fma_f32 (16 accumulators, four waves per SIMD, every CU busy) three
// times, with physical registers chosen by hand so that the three source operands of every
// instruction come from (A) one bank, (B) three different banks, (C) two banks -- and prints the
// shader cycles per wave instruction per SIMD for each.
// Build: hipcc --offload-arch=gfx950 -O2 tools/vgpr_bank_bench.hip -o tools/vgpr_bank_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 16384;

// accumulators v40, v44, ..., v100: all in bank 0.  S0 / S1: the two other source registers.
#define FMA16(S0, S1)                                                                                       \
    "v_fma_f32 v40, " S0 ", " S1 ", v40\n v_fma_f32 v44, " S0 ", " S1 ", v44\n v_fma_f32 v48, " S0 ", " S1 ", v48\n"   \
    "v_fma_f32 v52, " S0 ", " S1 ", v52\n v_fma_f32 v56, " S0 ", " S1 ", v56\n v_fma_f32 v60, " S0 ", " S1 ", v60\n"   \
    "v_fma_f32 v64, " S0 ", " S1 ", v64\n v_fma_f32 v68, " S0 ", " S1 ", v68\n v_fma_f32 v72, " S0 ", " S1 ", v72\n"   \
    "v_fma_f32 v76, " S0 ", " S1 ", v76\n v_fma_f32 v80, " S0 ", " S1 ", v80\n v_fma_f32 v84, " S0 ", " S1 ", v84\n"   \
    "v_fma_f32 v88, " S0 ", " S1 ", v88\n v_fma_f32 v92, " S0 ", " S1 ", v92\n v_fma_f32 v96, " S0 ", " S1 ", v96\n"   \
    "v_fma_f32 v100, " S0 ", " S1 ", v100\n"
// v_fmac_f32 (the two-operand encoding: the destination is the third source), same accumulators
#define FMAC16(S0, S1)                                                                                      \
    "v_fmac_f32 v40, " S0 ", " S1 "\n v_fmac_f32 v44, " S0 ", " S1 "\n v_fmac_f32 v48, " S0 ", " S1 "\n v_fmac_f32 v52, " S0 ", " S1 "\n" \
    "v_fmac_f32 v56, " S0 ", " S1 "\n v_fmac_f32 v60, " S0 ", " S1 "\n v_fmac_f32 v64, " S0 ", " S1 "\n v_fmac_f32 v68, " S0 ", " S1 "\n" \
    "v_fmac_f32 v72, " S0 ", " S1 "\n v_fmac_f32 v76, " S0 ", " S1 "\n v_fmac_f32 v80, " S0 ", " S1 "\n v_fmac_f32 v84, " S0 ", " S1 "\n" \
    "v_fmac_f32 v88, " S0 ", " S1 "\n v_fmac_f32 v92, " S0 ", " S1 "\n v_fmac_f32 v96, " S0 ", " S1 "\n v_fmac_f32 v100, " S0 ", " S1 "\n"
// v_mul_f32 into the accumulators (two sources only)
#define MUL16(S0, S1)                                                                                       \
    "v_mul_f32 v40, " S0 ", " S1 "\n v_mul_f32 v44, " S0 ", " S1 "\n v_mul_f32 v48, " S0 ", " S1 "\n v_mul_f32 v52, " S0 ", " S1 "\n" \
    "v_mul_f32 v56, " S0 ", " S1 "\n v_mul_f32 v60, " S0 ", " S1 "\n v_mul_f32 v64, " S0 ", " S1 "\n v_mul_f32 v68, " S0 ", " S1 "\n" \
    "v_mul_f32 v72, " S0 ", " S1 "\n v_mul_f32 v76, " S0 ", " S1 "\n v_mul_f32 v80, " S0 ", " S1 "\n v_mul_f32 v84, " S0 ", " S1 "\n" \
    "v_mul_f32 v88, " S0 ", " S1 "\n v_mul_f32 v92, " S0 ", " S1 "\n v_mul_f32 v96, " S0 ", " S1 "\n v_mul_f32 v100, " S0 ", " S1 "\n"
#define CLOBBERS "v20", "v21", "v24", "v25", "v26", "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68", "v72", "v76", "v80", "v84", "v88", "v92", "v96", "v100"

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc) {
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.5\n"
                 "v_mov_b32 v40, 0\n v_mov_b32 v44, 0\n v_mov_b32 v48, 0\n v_mov_b32 v52, 0\n v_mov_b32 v56, 0\n v_mov_b32 v60, 0\n"
                 "v_mov_b32 v64, 0\n v_mov_b32 v68, 0\n v_mov_b32 v72, 0\n v_mov_b32 v76, 0\n v_mov_b32 v80, 0\n v_mov_b32 v84, 0\n"
                 "v_mov_b32 v88, 0\n v_mov_b32 v92, 0\n v_mov_b32 v96, 0\n v_mov_b32 v100, 0\n" ::: CLOBBERS);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) asm volatile(FMA16("v20", "v24") ::: CLOBBERS);      // banks 0, 0, 0
        if (KIND == 1) asm volatile(FMA16("v21", "v26") ::: CLOBBERS);      // banks 1, 2, 0
        if (KIND == 2) asm volatile(FMA16("v20", "v26") ::: CLOBBERS);      // banks 0, 2, 0
        if (KIND == 3) asm volatile(FMA16("v21", "v21") ::: CLOBBERS);      // banks 1, 1, 0 (one register twice)
        if (KIND == 4) asm volatile(FMA16("v21", "v25") ::: CLOBBERS);      // banks 1, 1, 0 (two registers)
        if (KIND == 5) asm volatile(FMAC16("v21", "v26") ::: CLOBBERS);     // fmac: banks 1, 2, accumulator 0
        if (KIND == 6) asm volatile(FMAC16("v20", "v26") ::: CLOBBERS);     // fmac: banks 0, 2, accumulator 0
        if (KIND == 7) asm volatile(FMAC16("v21", "v24") ::: CLOBBERS);     // fmac: banks 1, 0, accumulator 0
        if (KIND == 8) asm volatile(FMAC16("v21", "v25") ::: CLOBBERS);     // fmac: banks 1, 1, accumulator 0
        if (KIND == 9) asm volatile(MUL16("v21", "v26") ::: CLOBBERS);      // mul: banks 1, 2
        if (KIND == 10) asm volatile(MUL16("v21", "v25") ::: CLOBBERS);     // mul: banks 1, 1
        if (KIND == 11) asm volatile(MUL16("v20", "v26") ::: CLOBBERS);     // mul: banks 0, 2 into bank 0 (destination only)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s;
    asm volatile("v_add_f32 %0, v40, v100" : "=v"(s) :: CLOBBERS);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
int run(const char* name, float* d, unsigned long long* dc, int ncu) {
    const int blocks = ncu * 4;              // four blocks of four waves per CU: four waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double wave_instr = (double)blocks * 4 * ITERS * 16;
    printf("%-34s %.3f ms  %.2f cyc/wave-instr/SIMD  lane-ops %.2f T/s\n", name, best, med / ((double)ITERS * 16 * 4),
           wave_instr * 64 / (best * 1e-3) * 1e-12);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float* d; unsigned long long* dc;
    CHECK(hipMalloc(&d, (size_t)ncu * 4 * 256 * 4));
    CHECK(hipMalloc(&dc, (size_t)ncu * 4 * 4 * 8));
    for (int rep = 0; rep < 1; ++rep) {
        run<0>("sources in banks 0, 0, 0", d, dc, ncu);
        run<1>("sources in banks 1, 2, 0", d, dc, ncu);
        run<2>("sources in banks 0, 2, 0", d, dc, ncu);
        run<3>("one register twice + bank 0", d, dc, ncu);
        run<4>("sources in banks 1, 1, 0", d, dc, ncu);
        run<5>("fmac: banks 1, 2, acc 0", d, dc, ncu);
        run<6>("fmac: banks 0, 2, acc 0", d, dc, ncu);
        run<7>("fmac: banks 1, 0, acc 0", d, dc, ncu);
        run<8>("fmac: banks 1, 1, acc 0", d, dc, ncu);
        run<9>("mul: banks 1, 2", d, dc, ncu);
        run<10>("mul: banks 1, 1", d, dc, ncu);
        run<11>("mul: banks 0, 2 (destination in 0)", d, dc, ncu);
    }
    return 0;
}
