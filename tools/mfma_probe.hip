// mfma_probe.hip — ceiling of the filter's instruction mix on gfx950: per 32x32 tile one
// v_mfma_f32_32x32x16_f16 plus an 8-op v_min3_f32 tree (+ compare), operands in registers.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// MODE 0: MFMA only; 1: MFMA + min3 tree + compare/branch per tile; 2: min3 tree only;
// 3: MFMA + min3 tree folded into a running minimum (no compare, no branch);
// 4: MFMA + min3 tree + (acc |= mn ^ thr), one compare/branch per 4 tiles
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void probe(const h8 *__restrict__ in, float *__restrict__ out, int iters, float thr)
{
    const int lane = threadIdx.x & 63;
    h8 q[16];
    for (int t = 0; t < 16; ++t) q[t] = in[t * 64 + lane];
    h8 a = in[16 * 64 + lane];
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = (float)i * 0.01f;
    float um = 1e30f;
    unsigned hits = 0, acc = 0;
    f16v d[2];
    d[0] = c; d[1] = c;
    if (MODE == 7 || MODE == 8) {   // vdst rotates over 4 tuples, srcC = a fixed VGPR tile (7) or literal zero (8)
        f16v e[4];
        e[0] = c; e[1] = c; e[2] = c; e[3] = c;
        f16v z;
        for (int i = 0; i < 16; ++i) z[i] = 0.0f;
        float s = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                e[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, q[t], MODE == 7 ? c : z, 0, 0, 0);
                if ((t & 3) == 3)
                    asm volatile("" ::"v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]));
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = s + e[0][0] + e[1][1] + e[2][2] + e[3][3];
        return;
    }
    if (MODE == 5 || MODE == 6) {   // pure accumulate chains: 4 (MODE 5) or 2 (MODE 6) rotating accumulators
        f16v e[4];
        e[0] = c; e[1] = c; e[2] = c; e[3] = c;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = MODE == 5 ? (t & 3) : (t & 1);
                e[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, q[t], e[j], 0, 0, 0);
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = e[0][0] + e[1][1] + e[2][2] + e[3][3];
        return;
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE != 2)
            d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, q[0], c, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (MODE != 2 && t + 1 < 16)
                d[(t + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, q[t + 1], c, 0, 0, 0);
            if (MODE != 0) {
                f16v &x = d[t & 1];
                if (MODE == 2) {
                    asm volatile("" : "+v"(x));   // keep the tree from being hoisted out of the loop
                }
                const float m0 = min3f(x[0], x[1], x[2]);
                const float m1 = min3f(x[3], x[4], x[5]);
                const float m2 = min3f(x[6], x[7], x[8]);
                const float m3 = min3f(x[9], x[10], x[11]);
                const float m4 = min3f(x[12], x[13], x[14]);
                const float m5 = min3f(m0, m1, m2);
                const float m6 = min3f(m3, m4, x[15]);
                if (MODE == 3) {
                    um = min3f(m5, m6, um);
                } else if (MODE == 4) {
                    const float mn = min3f(m5, m6, thr);
                    acc |= __float_as_uint(mn) ^ __float_as_uint(thr);
                    if ((t & 3) == 3) {
                        if (__builtin_expect(acc != 0u, 0)) { ++hits; um = mn; }
                        acc = 0u;
                    }
                } else {
                    const float mn = min3f(m5, m6, thr);
                    if (__builtin_expect(mn < thr, 0)) { ++hits; um = mn; }
                }
            } else {
                asm volatile("" :: "v"(d[t & 1]));
            }
        }
        a[0] += (_Float16)0.001f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = um + hits + d[0][3] + d[1][5];
}

template <int MODE, int WPS>
static int run(const char *name, const h8 *in, float *out, int cus, hipEvent_t a, hipEvent_t b)
{
    const int iters = 4000;
    const int blocks = cus * WPS;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((probe<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, in, out, iters, -1e30f);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double tiles = (double)blocks * 4 * iters * 16;
    const double cyc_per_tile_per_simd = best * 1e-3 * 2.4e9 / (tiles / (cus * 4.0));
    printf("%-28s waves/SIMD %d: %8.3f ms  %7.1f TFLOP/s-equivalent  %6.1f cycles@2.4GHz per tile per SIMD\n", name, WPS, best,
           tiles * 32768.0 / (best * 1e-3) / 1e12, cyc_per_tile_per_simd);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    h8 *in;
    float *out;
    CHK(hipMalloc(&in, 17 * 64 * 16));
    CHK(hipMemset(in, 0x3c, 17 * 64 * 16));
    CHK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    if (run<0, 1>("mfma only", in, out, cus, a, b)) return 1;
    if (run<0, 2>("mfma only", in, out, cus, a, b)) return 1;
    if (run<1, 1>("mfma + min3 tree + cmp", in, out, cus, a, b)) return 1;
    if (run<1, 2>("mfma + min3 tree + cmp", in, out, cus, a, b)) return 1;
    if (run<2, 1>("min3 tree + cmp only", in, out, cus, a, b)) return 1;
    if (run<2, 2>("min3 tree + cmp only", in, out, cus, a, b)) return 1;
    if (run<5, 1>("mfma chains x4 accumulators", in, out, cus, a, b)) return 1;
    if (run<5, 2>("mfma chains x4 accumulators", in, out, cus, a, b)) return 1;
    if (run<6, 1>("mfma chains x2 accumulators", in, out, cus, a, b)) return 1;
    if (run<6, 2>("mfma chains x2 accumulators", in, out, cus, a, b)) return 1;
    if (run<7, 1>("mfma srcC = fixed VGPR tile", in, out, cus, a, b)) return 1;
    if (run<7, 2>("mfma srcC = fixed VGPR tile", in, out, cus, a, b)) return 1;
    if (run<8, 1>("mfma srcC = 0", in, out, cus, a, b)) return 1;
    if (run<8, 2>("mfma srcC = 0", in, out, cus, a, b)) return 1;
    if (run<3, 1>("mfma + min3 running min", in, out, cus, a, b)) return 1;
    if (run<3, 2>("mfma + min3 running min", in, out, cus, a, b)) return 1;
    if (run<4, 1>("mfma + min3 + xor/or, cmp/4", in, out, cus, a, b)) return 1;
    if (run<4, 2>("mfma + min3 + xor/or, cmp/4", in, out, cus, a, b)) return 1;
    return 0;
}
