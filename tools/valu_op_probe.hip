// valu_op_probe.hip — throughput of candidate reduction instructions on gfx950 (8 x 3-operand
// tree or 16 x 2-operand tree over 16 registers), 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define TREE3(OP) \
    OP " %0, %1, %2, %3\n\t" OP " %1, %4, %5, %6\n\t" OP " %2, %7, %8, %9\n\t" OP " %3, %10, %11, %12\n\t" \
    OP " %4, %13, %14, %15\n\t" OP " %0, %0, %1, %2\n\t" OP " %1, %3, %4, %16\n\t" OP " %0, %0, %1, %17\n\t"
#define TREE2(OP) \
    OP " %0, %1, %2\n\t" OP " %1, %3, %4\n\t" OP " %2, %5, %6\n\t" OP " %3, %7, %8\n\t" OP " %4, %9, %10\n\t" \
    OP " %5, %11, %12\n\t" OP " %6, %13, %14\n\t" OP " %7, %15, %16\n\t" OP " %0, %0, %1\n\t" OP " %2, %2, %3\n\t" \
    OP " %4, %4, %5\n\t" OP " %6, %6, %7\n\t" OP " %0, %0, %2\n\t" OP " %4, %4, %6\n\t" OP " %0, %0, %4\n\t" \
    OP " %0, %0, %17\n\t"

#define DEF3(NAME, OP) \
__global__ __launch_bounds__(256) void NAME(float *__restrict__ out, int iters, float thr, float seed) { \
    f16v x; for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x + i; unsigned hits = 0; \
    for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int u = 0; u < 16; ++u) { float mn; \
        asm volatile(TREE3(OP) : "=&v"(mn), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]) \
            : "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), \
              "v"(x[13]), "v"(x[14]), "v"(x[15]), "v"(x[0]), "v"(thr)); \
        if (__builtin_expect(__float_as_int(mn) < __float_as_int(thr), 0)) ++hits; } } \
    float s = hits; for (int i = 0; i < 16; ++i) s += x[i]; out[blockIdx.x * 256 + threadIdx.x] = s; }
#define DEF2(NAME, OP) \
__global__ __launch_bounds__(256) void NAME(float *__restrict__ out, int iters, float thr, float seed) { \
    f16v x; for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x + i; unsigned hits = 0; \
    for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int u = 0; u < 16; ++u) { float mn; \
        asm volatile(TREE2(OP) : "=&v"(mn), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
            : "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]), \
              "v"(x[0]), "v"(thr)); \
        if (__builtin_expect(__float_as_int(mn) < __float_as_int(thr), 0)) ++hits; } } \
    float s = hits; for (int i = 0; i < 16; ++i) s += x[i]; out[blockIdx.x * 256 + threadIdx.x] = s; }

DEF3(k_min3_f32, "v_min3_f32")
DEF3(k_min3_i32, "v_min3_i32")
DEF3(k_min3_u32, "v_min3_u32")
DEF3(k_or3_b32, "v_or3_b32")
DEF3(k_max3_f32, "v_max3_f32")
DEF3(k_minimum3_f32, "v_minimum3_f32")
DEF3(k_med3_f32, "v_med3_f32")
DEF3(k_add3_u32, "v_add3_u32")
DEF2(k_min_f32, "v_min_f32")
DEF2(k_min_i32, "v_min_i32")
DEF2(k_min_u32, "v_min_u32")
DEF2(k_and_b32, "v_and_b32")
DEF2(k_add_f32, "v_add_f32")
DEF2(k_max_f32, "v_max_f32")

typedef void (*kern_t)(float *, int, float, float);
static int run(const char *name, kern_t k, float *out, int cus, hipEvent_t a, hipEvent_t b)
{
    const int iters = 3000;
    printf("%-18s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        const int blocks = cus * wps;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, -1e30f, 1.0f);
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double tiles = (double)blocks * 4 * iters * 16;
        printf("  w%d: %6.1f", wps, best * 1e-3 * 2.4e9 / (tiles / (cus * 4.0)));
    }
    printf("   cycles@2.4GHz per 16-register reduction per SIMD\n");
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out;
    CHK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
#define RUN(K) if (run(#K, K, out, cus, a, b)) return 1;
    RUN(k_min3_f32) RUN(k_min3_i32) RUN(k_min3_u32) RUN(k_or3_b32) RUN(k_max3_f32) RUN(k_minimum3_f32) RUN(k_med3_f32) RUN(k_add3_u32)
    RUN(k_min_f32) RUN(k_min_i32) RUN(k_min_u32) RUN(k_and_b32) RUN(k_add_f32) RUN(k_max_f32)
    return 0;
}
