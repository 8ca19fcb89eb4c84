// filter_loop_bench.hip -- the filter loop of general_intersect_kernel on its own, built up in
// steps, to see which part keeps it from issuing an instruction every two cycles.
// A lane holds 8 rays (direction and moment), 256-thread blocks at four waves per SIMD walk 4096
// records staged in LDS (20 floats each, as the kernel's); per ray and record
//   MODE 0: the 15 fused multiply-adds (three dot-product chains)
//   MODE 1: + W = det - Nb - Ng (two subtractions)
//   MODE 2: + the verdict (two v_max3_f32, one v_sub, one v_or) and the or-tree / vote per record
//   MODE 3: MODE 2 with the records in registers instead of LDS (one record, perturbed per iteration)
//   MODE 4: MODE 2 with the kernel's LDS footprint; MODE 5: on random mantissas; MODE 6: ray after ray;
//   MODE 7: MODE 5 with the two multiplicands of every multiply-add exchanged
// Prints wall time, VALU instructions per (ray, record) counted by hand, and cycles per wave
// instruction per SIMD at an assumed 2.1 GHz (the clock under load was not measured).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/filter_loop_bench.hip -o tools/filter_loop_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kRays = 8, kRecords = 4096, kChunk = 256;

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz, float start) {
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, __builtin_fmaf(ax, bx, start)));
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void loop_bench(const float4* __restrict__ rec, float* __restrict__ out, int n_rec) {
    __shared__ float4 s_rec[kChunk * 5];
    // MODE 4: MODE 2 with the kernel's 38 KB of LDS per block, i.e. four blocks per CU, four waves per SIMD
    __shared__ float4 s_pad[MODE >= 4 ? 1152 : 1];
    const int tid = threadIdx.x;
    if (MODE >= 4 && tid == 0) s_pad[blockIdx.x % 1152] = make_float4(0.f, 0.f, 0.f, 0.f);
    float dx[kRays], dy[kRays], dz[kRays], mx[kRays], my[kRays], mz[kRays];
#pragma unroll
    for (int s = 0; s < kRays; ++s) {
        if (MODE == 5 || MODE == 7) {
            // MODE 5: MODE 4 on operands with random mantissas and signs (what the arithmetic units switch per
            // instruction, hence the power they draw, depends on the bits)
            uint32_t h = (uint32_t)(tid + 256 * s + 2048 * blockIdx.x) * 2654435761u;
            auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return __uint_as_float((h & 0x807FFFFFu) | 0x3F000000u); };
            dx[s] = rnd(); dy[s] = rnd(); dz[s] = rnd(); mx[s] = rnd(); my[s] = rnd(); mz[s] = rnd();
        } else {
            const float t = (float)(tid + 256 * s + blockIdx.x) * 1e-3f;
            dx[s] = t; dy[s] = 1.0f - t; dz[s] = 0.5f + t; mx[s] = t * 0.25f; my[s] = 0.125f - t; mz[s] = t * t;
        }
    }
    uint32_t seen = 0;
    for (int base = 0; base < n_rec; base += kChunk) {
        __syncthreads();
        for (int k = 0; k < 5; ++k) s_rec[tid + k * 256] = rec[(size_t)base * 5 + tid + k * 256];
        __syncthreads();
        for (int g = 0; g < kChunk; ++g) {
            float4 q0, q1, q2, q3, q4;
            if (MODE == 3) {
                q0 = s_rec[0]; q1 = s_rec[1]; q2 = s_rec[2]; q3 = s_rec[3]; q4 = s_rec[4];
                q0.x = __uint_as_float(__float_as_uint(q0.x) ^ (uint32_t)g);
                q1.x = __uint_as_float(__float_as_uint(q1.x) ^ (uint32_t)g);
                q4.x = __uint_as_float(__float_as_uint(q4.x) ^ (uint32_t)g);
            } else {
                q0 = s_rec[g * 5 + 0]; q1 = s_rec[g * 5 + 1]; q2 = s_rec[g * 5 + 2]; q3 = s_rec[g * 5 + 3]; q4 = s_rec[g * 5 + 4];
            }
            uint32_t any = 0;
#pragma unroll
            for (int s = 0; s < kRays; ++s) {
                float xb, xg, xd;
                if (MODE == 7) {
                    // MODE 7: MODE 5 with the multiplicands exchanged: the ray's component (different in every lane)
                    // is the first source operand, the record's (the same in all 64 lanes) the second
                    xb = dot3(mx[s], my[s], mz[s], q2.x, q2.y, q2.z, dot3(dx[s], dy[s], dz[s], q0.x, q0.y, q0.z, q0.w));
                    xg = dot3(mx[s], my[s], mz[s], q3.x, q3.y, q3.z, dot3(dx[s], dy[s], dz[s], q1.x, q1.y, q1.z, q0.w));
                    xd = dot3(dx[s], dy[s], dz[s], q4.x, q4.y, q4.z, q1.w);
                } else {
                    xb = dot3(q2.x, q2.y, q2.z, mx[s], my[s], mz[s], dot3(q0.x, q0.y, q0.z, dx[s], dy[s], dz[s], q0.w));
                    xg = dot3(q3.x, q3.y, q3.z, mx[s], my[s], mz[s], dot3(q1.x, q1.y, q1.z, dx[s], dy[s], dz[s], q0.w));
                    xd = dot3(q4.x, q4.y, q4.z, dx[s], dy[s], dz[s], q1.w);
                }
                if (MODE == 0) {
                    asm volatile("" :: "v"(xb), "v"(xg), "v"(xd));
                } else {
                    const float xw = (xd - xb) - xg;
                    if (MODE == 1) {
                        asm volatile("" :: "v"(xw));
                    } else {
                        const float nlo = __builtin_fmaxf(__builtin_fmaxf(-xb, -xg), -xw);
                        const float hi = __builtin_fmaxf(__builtin_fmaxf(xb, xg), xw);
                        any |= __float_as_uint(nlo) | __float_as_uint(hi - q2.w);
                    }
                }
                // MODE 6: MODE 4 with one ray finished before the next begins (the order the compiler
                // gives the kernel's loop, where more values are live)
                if (MODE == 6) __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE >= 2) {
                if (__builtin_amdgcn_sicmp((int32_t)any, 0, 40 /* ICMP_SLT */) == 0) continue;
                seen += 1;          // (rare with the records this program makes)
            }
        }
    }
    out[blockIdx.x * 256 + tid] = (float)seen + dx[0];
}

template <int MODE>
int run(const char* name, const float4* d_rec_plain, float* d_out, int ncu, double valu_per_pair, const float4* d_rec_random = nullptr) {
    const float4* d_rec = d_rec_random ? d_rec_random : d_rec_plain;
    const int blocks = ncu * 4 * 6;          // six rounds of what the GPU holds, as the kernel's launches
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(loop_bench<MODE>, dim3(blocks), dim3(256), 0, 0, d_rec, d_out, kRecords);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(loop_bench<MODE>, dim3(blocks), dim3(256), 0, 0, d_rec, d_out, kRecords);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double pairs = (double)blocks * 256 * kRays * kRecords;
    const double wave_instr_per_simd = pairs * valu_per_pair / 64.0 / (ncu * 4);
    printf("%-44s %.3f ms  %.3e pairs/s  %.1f VALU/pair (by hand)  %.2f cycles per wave instruction per SIMD at 2.1 GHz\n", name, best,
           pairs / (best * 1e-3), valu_per_pair, best * 1e-3 * 2.1e9 / wave_instr_per_simd);
    return 0;
}

int main(int argc, char** argv) {
    // filter_loop_bench MODE SECONDS: only that mode (4 or 5), again and again for that long -- for
    // tools/sample_clocks.sh to read the clock beside it
    const int only_mode = argc > 2 ? atoi(argv[1]) : -1;
    const double seconds = argc > 2 ? atof(argv[2]) : 0.0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    std::vector<float4> h(kRecords * 5);
    for (int k = 0; k < kRecords; ++k) {
        // vectors that make det, Nb, Ng of mixed signs for nearly every ray: the vote almost never passes
        h[k * 5 + 0] = make_float4(1.0f + k * 1e-4f, -2.0f, 0.5f, 1e-6f);
        h[k * 5 + 1] = make_float4(-1.5f, 0.25f + k * 1e-4f, 1.0f, 3e-6f);
        h[k * 5 + 2] = make_float4(0.5f, 1.0f, -1.0f, 2e-6f);
        h[k * 5 + 3] = make_float4(-0.5f, 2.0f, 0.25f, 0.0f);
        h[k * 5 + 4] = make_float4(1.0f, 1.0f, -1.0f - k * 1e-4f, 1e-6f);
    }
    float4* d_rec; float* d_out;
    CHECK(hipMalloc(&d_rec, h.size() * sizeof(float4)));
    CHECK(hipMemcpy(d_rec, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)ncu * 4 * 6 * 256 * sizeof(float)));
    // records with random mantissas and signs, margins as before
    std::vector<float4> hr(h);
    uint32_t x = 12345u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; uint32_t b = (x & 0x807FFFFFu) | 0x3F000000u; float f; memcpy(&f, &b, 4); return f; };
    for (int k = 0; k < kRecords * 5; ++k) { hr[k].x = rnd(); hr[k].y = rnd(); hr[k].z = rnd(); }
    float4* d_rec_random;
    CHECK(hipMalloc(&d_rec_random, hr.size() * sizeof(float4)));
    CHECK(hipMemcpy(d_rec_random, hr.data(), hr.size() * sizeof(float4), hipMemcpyHostToDevice));
    if (only_mode >= 0) {
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            if (only_mode == 7) run<7>("random operands, multiplicands exchanged", d_rec, d_out, ncu, 21.0 + 0.75, d_rec_random);
            else if (only_mode == 5) run<5>("random mantissas and signs", d_rec, d_out, ncu, 21.0 + 0.75, d_rec_random);
            else run<4>("plain operands", d_rec, d_out, ncu, 21.0 + 0.75);
        }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("15 fma", d_rec, d_out, ncu, 15.0 + 0.2);
        run<1>("15 fma + 2 sub", d_rec, d_out, ncu, 17.0 + 0.2);
        run<2>("15 fma + 2 sub + verdict, records in LDS", d_rec, d_out, ncu, 21.0 + 0.75);
        run<3>("... records in registers", d_rec, d_out, ncu, 21.0 + 0.75 + 0.4);
        run<4>("... records in LDS, four waves per SIMD", d_rec, d_out, ncu, 21.0 + 0.75);
        run<5>("... the same on random mantissas and signs", d_rec, d_out, ncu, 21.0 + 0.75, d_rec_random);
        run<6>("... four waves per SIMD, ray after ray", d_rec, d_out, ncu, 21.0 + 0.75);
        run<7>("... random operands, multiplicands exchanged", d_rec, d_out, ncu, 21.0 + 0.75, d_rec_random);
        run<5>("... random operands, as before (again)", d_rec, d_out, ncu, 21.0 + 0.75, d_rec_random);
    }
    return 0;
}
