// valu_dep_bench.hip -- how many independent v_fmac_f32 chains does a wave need on gfx950?
// Four waves per SIMD, every CU busy; each wave issues a stream of v_fmac_f32 in which an
// instruction depends on the one CHAINS instructions before it.  Prints shader cycles per wave
// instruction per SIMD for CHAINS = 1, 2, 3, 4, 6, 8, 16.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_dep_bench.hip -o tools/valu_dep_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 8192;
constexpr int PER_ITER = 48;     // instructions per loop iteration (a multiple of every CHAINS)

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float a, float b) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    asm volatile("" : "+v"(a), "+v"(b));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < PER_ITER; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j % CHAINS]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS>
int run(float* d, unsigned long long* dc, int ncu, int blocks_per_cu) {
    const int blocks = ncu * blocks_per_cu;
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 0.9999f);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 0.9999f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("waves/SIMD %d  chains %2d: %.2f cyc/wave-instr/SIMD  (%.3f ms)\n", blocks_per_cu, CHAINS,
           med / ((double)ITERS * PER_ITER * blocks_per_cu), ms);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float* d; unsigned long long* dc;
    CHECK(hipMalloc(&d, (size_t)ncu * 8 * 256 * 4));
    CHECK(hipMalloc(&dc, (size_t)ncu * 8 * 4 * 8));
    for (int occ : {1, 2, 4, 6}) {
        run<1>(d, dc, ncu, occ); run<2>(d, dc, ncu, occ); run<3>(d, dc, ncu, occ); run<4>(d, dc, ncu, occ);
        run<6>(d, dc, ncu, occ); run<8>(d, dc, ncu, occ); run<16>(d, dc, ncu, occ);
    }
    return 0;
}
