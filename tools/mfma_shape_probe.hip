// mfma_shape_probe.hip — does the chip hold a higher clock on v_mfma_f32_16x16x32_f16 than on v_mfma_f32_32x32x16_f16?
// (MI355X_MICROARCH.md, "DVFS give-back" item 7, measured the bf16 pair: 1.15x the FLOP/s at equal cycles per FLOP on random
// data.)  Accumulate chains on RANDOM f16 operands held in registers, the same flops per wave and the same number of
// accumulator registers (64) for both shapes, long launches (the clock needs ~0.1 s to settle), 1 and 2 waves per SIMD; the
// in-kernel clock from s_memtime / s_memrealtime.
// Build (in tools/): hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_shape_probe mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// SHAPE 0: 32x32x16, 4 accumulators of 16 registers; per iteration 16 MFMAs = 16 x 32768 flop
// SHAPE 1: 16x16x32, 16 accumulators of 4 registers; per iteration 32 MFMAs = 32 x 16384 flop
template <int SHAPE, int WPS>
__global__ __launch_bounds__(256, WPS) void probe(const h8 *__restrict__ in, float *__restrict__ out, int iters,
                                                  unsigned long long *__restrict__ stamps)
{
    const int lane = threadIdx.x & 63;
    h8 q[16], a[2];
    for (int t = 0; t < 16; ++t)
        q[t] = in[(t * 64 + lane + blockIdx.x * 7) & 4095];
    a[0] = in[(16 * 64 + lane) & 4095];
    a[1] = in[(17 * 64 + lane) & 4095];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    if (SHAPE == 0) {
        f16v e[4];
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 16; ++i)
                e[j][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                e[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 1], q[t], e[t & 3], 0, 0, 0);
        }
        sink = e[0][0] + e[1][1] + e[2][2] + e[3][3];
    } else {
        f4v e[16];
        for (int j = 0; j < 16; ++j)
            e[j] = (f4v){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 32; ++t)
                e[t & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t & 1], q[t & 15], e[t & 15], 0, 0, 0);
        }
        for (int j = 0; j < 16; ++j)
            sink += e[j][j & 3];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sink;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int SHAPE, int WPS>
static int run(const char *name, const h8 *in, float *out, unsigned long long *stamps, int cus, hipEvent_t a, hipEvent_t b)
{
    const int iters = 400000;   // ~0.1-0.2 s per launch
    const int blocks = cus * WPS;
    float best = 1e30f;
    double mhz = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((probe<SHAPE, WPS>), dim3(blocks), dim3(256), 0, 0, in, out, iters, stamps);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) {
            best = ms;
            unsigned long long h[2 * 512];
            CHK(hipMemcpy(h, stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
            double s = 0;
            for (int i = 0; i < blocks; ++i)
                s += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
            mhz = s / blocks;
        }
    }
    const double flop = (double)blocks * 4 * iters * 16 * 32768.0;
    printf("%-22s waves/SIMD %d: %8.2f ms  %7.1f TFLOP/s  in-kernel clock %.0f MHz\n", name, WPS, best, flop / (best * 1e-3) / 1e12, mhz);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    h8 *in;
    float *out;
    unsigned long long *stamps;
    _Float16 *host = (_Float16 *)malloc(4096 * 16);
    srand(7);
    for (int i = 0; i < 4096 * 8; ++i)
        host[i] = (_Float16)((rand() / (double)RAND_MAX - 0.5) * 1.0);
    CHK(hipMalloc(&in, 4096 * 16));
    CHK(hipMemcpy(in, host, 4096 * 16, hipMemcpyHostToDevice));
    CHK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    CHK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * cus * 8));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    for (int round = 0; round < 2; ++round) {
        if (run<0, 1>("32x32x16 f16 chains", in, out, stamps, cus, a, b)) return 1;
        if (run<1, 1>("16x16x32 f16 chains", in, out, stamps, cus, a, b)) return 1;
        if (run<0, 2>("32x32x16 f16 chains", in, out, stamps, cus, a, b)) return 1;
        if (run<1, 2>("16x16x32 f16 chains", in, out, stamps, cus, a, b)) return 1;
    }
    return 0;
}
