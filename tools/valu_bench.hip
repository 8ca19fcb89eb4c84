// valu_bench.hip -- measures the sustained issue rate of FP32 / integer VALU
// instructions on gfx950, to price the render kernel's inner loop.  Reports
// shader cycles per wave-instruction per SIMD (s_memtime) and wall time.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 32768;
constexpr int UNROLL = 16;   // independent chains per lane

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float a, float b) {
    float v[UNROLL];
    float2 p[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { v[i] = threadIdx.x * 0.001f + i; p[i] = make_float2(v[i], v[i] + 1); }
    float2 ab = make_float2(a, b);
    float c = a + b;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(ab), "v"(ab));
            if (KIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(ab));
            if (KIND == 4) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "s"(a), "v"(b));
            if (KIND == 6) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 8) asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 9) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 10) asm volatile("v_min_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 11) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 12) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 13) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[i]) : "s"(a));
            if (KIND == 14) asm volatile("v_min_u32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 15) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(a), "v"(b), "v"(c));
            if (KIND == 16) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 17) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 18) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 19) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "s"(a), "v"(b));
            if (KIND == 20) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 21) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 22) asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 23) asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 24) asm volatile("v_pk_add_i16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 25) asm volatile("v_pk_min_i16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 26) asm volatile("v_pk_max_i16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 27) asm volatile("v_or_b32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 28) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 29) asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 30) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 31) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 32) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(a), "v"(b));
            if (KIND == 33) asm volatile("v_min_f16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 34) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 35) asm volatile("v_max_i32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            if (KIND == 36) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(v[i]) : "v"(a));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < UNROLL; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
int run(const char* name, int blocks_per_cu, float* d, unsigned long long* dc, int ncu) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int blocks = ncu * blocks_per_cu;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 0.9999f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 0.9999f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    // each wave issued ITERS*UNROLL instructions in `med` cycles while sharing its SIMD with blocks_per_cu waves
    double cyc_per_instr = med / ((double)ITERS * UNROLL * blocks_per_cu);
    double wave_instr = (double)blocks * 4 * ITERS * UNROLL;
    printf("%-24s waves/SIMD=%d  %.3f ms  %.2f cyc/wave-instr/SIMD  eff clock %.2f GHz  lane-ops %.2f T/s\n",
           name, blocks_per_cu, best, cyc_per_instr, med / (best * 1e-3) * 1e-9, wave_instr * 64 / (best * 1e-3) * 1e-12);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz\n", prop.name, ncu, prop.clockRate);
    float* d; unsigned long long* dc;
    CHECK(hipMalloc(&d, (size_t)ncu * 8 * 256 * 4));
    CHECK(hipMalloc(&dc, (size_t)ncu * 8 * 4 * 8));
    for (int occ : {4}) {
        run<20>("v_pk_add_f16", occ, d, dc, ncu);
        run<31>("v_pk_add_f16 2src", occ, d, dc, ncu);
        run<21>("v_pk_fma_f16", occ, d, dc, ncu);
        run<29>("v_pk_mul_f16", occ, d, dc, ncu);
        run<22>("v_pk_min_f16", occ, d, dc, ncu);
        run<23>("v_pk_max_f16", occ, d, dc, ncu);
        run<33>("v_min_f16", occ, d, dc, ncu);
        run<24>("v_pk_add_i16", occ, d, dc, ncu);
        run<36>("v_pk_add_u16", occ, d, dc, ncu);
        run<25>("v_pk_min_i16", occ, d, dc, ncu);
        run<26>("v_pk_max_i16", occ, d, dc, ncu);
        run<27>("v_or_b32", occ, d, dc, ncu);
        run<34>("v_xor_b32", occ, d, dc, ncu);
        run<28>("v_add_u32", occ, d, dc, ncu);
        run<35>("v_max_i32", occ, d, dc, ncu);
        run<30>("v_cvt_pkrtz_f16_f32", occ, d, dc, ncu);
        run<32>("v_add_f32 2src", occ, d, dc, ncu);
    }
    for (int occ : {4}) {
        run<0>("v_fma_f32 d=acc", occ, d, dc, ncu);
        run<15>("v_fma_f32 3 src", occ, d, dc, ncu);
        run<5>("v_fma_f32 sgpr", occ, d, dc, ncu);
        run<7>("v_fmac_f32", occ, d, dc, ncu);
        run<19>("v_fmac_f32 sgpr", occ, d, dc, ncu);
        run<1>("v_pk_fma_f32", occ, d, dc, ncu);
        run<2>("v_add_f32", occ, d, dc, ncu);
        run<13>("v_add_f32 sgpr", occ, d, dc, ncu);
        run<11>("v_mul_f32", occ, d, dc, ncu);
        run<3>("v_pk_add_f32", occ, d, dc, ncu);
        run<10>("v_min_f32", occ, d, dc, ncu);
        run<6>("v_max_f32", occ, d, dc, ncu);
        run<4>("v_min3_f32", occ, d, dc, ncu);
        run<12>("v_max3_f32", occ, d, dc, ncu);
        run<18>("v_med3_f32", occ, d, dc, ncu);
        run<14>("v_min_u32", occ, d, dc, ncu);
        run<9>("v_and_b32", occ, d, dc, ncu);
        run<8>("v_or3_b32", occ, d, dc, ncu);
        run<16>("v_and_or_b32", occ, d, dc, ncu);
        run<17>("v_add3_u32", occ, d, dc, ncu);
    }
    return 0;
}
