// valu_probe.hip — measures what the exact kernels' arithmetic can reach on gfx950:
// lane-ops/s of plain v_add/v_mul_f32 chains vs packed v_pk_add/v_pk_mul_f32 chains (no FMA),
// at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 8 independent chains of (sub, mul, add) per lane, scalar fp32: 24 lane-ops per iteration.
__global__ __launch_bounds__(256) void probe_scalar(float *out, float q0, float r0, int iters)
{
    float q[8], acc[8];
    for (int i = 0; i < 8; ++i) { q[i] = q0 + threadIdx.x * 1e-3f + i; acc[i] = 0.f; }
    float r = r0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float d = q[i] - r;
                float p = d * d;
                acc[i] = acc[i] + p;
            }
            r += 1.0f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 8 independent packed chains: 48 lane-ops per iteration-unit.
__global__ __launch_bounds__(256) void probe_packed(float *out, float q0, float r0, int iters)
{
    f2 q[8], acc[8];
    for (int i = 0; i < 8; ++i) { q[i].x = q0 + threadIdx.x * 1e-3f + i; q[i].y = q[i].x + 0.5f; acc[i] = (f2){0.f, 0.f}; }
    float r = r0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f2 rr = {r, r};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f2 d = q[i] - rr;
                f2 p = d * d;
                acc[i] = acc[i] + p;
            }
            r += 1.0f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// plain streaming read: sum of a big buffer with 16-B loads (the HBM ceiling on this box)
__global__ __launch_bounds__(256) void probe_stream(const float4 *__restrict__ in, long long n4, float *out)
{
    float s = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = in[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float *out;
    CHK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {           // waves per SIMD = blocks of 256 per CU
        for (int packed = 0; packed < 2; ++packed) {
            const int blocks = cus * wps;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipEventRecord(a));
                if (packed)
                    hipLaunchKernelGGL(probe_packed, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.25f, iters);
                else
                    hipLaunchKernelGGL(probe_scalar, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.25f, iters);
                CHK(hipEventRecord(b));
                CHK(hipEventSynchronize(b));
                float ms;
                CHK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            const double lane_ops = (double)blocks * 256 * iters * 4 * 8 * 3 * (packed ? 2 : 1);
            printf("waves/SIMD %d  %s : %8.3f ms  %7.2f T lane-ops/s\n", wps, packed ? "v_pk_*_f32" : "v_*_f32   ",
                   best, lane_ops / (best * 1e-3) / 1e12);
        }
    }
    // HBM streaming-read ceiling
    const long long bytes = 4ll << 30;
    float4 *buf;
    CHK(hipMalloc(&buf, bytes));
    CHK(hipMemset(buf, 0, bytes));
    for (int bpc = 4; bpc <= 16; bpc *= 2) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(a));
            hipLaunchKernelGGL(probe_stream, dim3(cus * bpc), dim3(256), 0, 0, buf, bytes / 16, out);
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("stream read 4 GiB, %2d blocks/CU: %7.3f ms  %7.1f GB/s\n", bpc, best, bytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
