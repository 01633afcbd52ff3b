// cvt_probe.hip — what v_cvt_scalef32_2xpk16_bf6_f32 does with 32 fp32 values per lane (gfx950), and
// what it costs.  Idea under test: a filter step only needs to know whether ANY of a lane's 16 scores
// is under the lane's threshold.  With the threshold folded into the MFMA (score - thr comes out of
// the matrix core), that is "any sign bit set", and this instruction squeezes 32 accumulator
// registers into 6 (one 6-bit field per value, sign on top) in ONE vector instruction where the
// v_min3_f32 tree needs 16.
// Build: hipcc --offload-arch=gfx950 -O3 -o cvt_probe cvt_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u6v __attribute__((ext_vector_type(6)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#if defined(__HIP_DEVICE_COMPILE__)
#define KEEP1(X) asm volatile("" ::"v"(X))
#define KEEP2(X, Y) asm volatile("" ::"v"(X), "v"(Y))
#else
#define KEEP1(X) (void)(X)
#define KEEP2(X, Y) ((void)(X), (void)(Y))
#endif
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- 1. semantics: every lane converts in[lane][0..31] with scale[lane] ----
__global__ void k_semantics(const float *__restrict__ in, const float *__restrict__ scale, unsigned *__restrict__ out)
{
    const int lane = threadIdx.x;
    f16v a, b;
    for (int i = 0; i < 16; ++i) {
        a[i] = in[lane * 32 + i];
        b[i] = in[lane * 32 + 16 + i];
    }
    const float s = scale[lane];
    u6v r;
    asm volatile("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(s));
    for (int i = 0; i < 6; ++i)
        out[lane * 6 + i] = r[i];
}

// ---- 2. cost: MODE 0 = cvt only (4 independent destinations), 1 = per pair of MFMAs one cvt + three
// v_or3_b32 (the candidate hot loop), 2 = the two MFMAs alone, 3 = per pair of MFMAs two 8-op min3 trees
// (today's loop), all one wave per SIMD, operands in registers. ----
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_cost(const h8 *__restrict__ in, unsigned *__restrict__ out, int iters,
                                                 unsigned long long *__restrict__ stamps)
{
    const int lane = threadIdx.x & 63;
    h8 q[8];
    for (int t = 0; t < 8; ++t)
        q[t] = in[t * 64 + lane];
    const h8 a = in[8 * 64 + lane];
    f16v c;
    for (int i = 0; i < 16; ++i)
        c[i] = (float)(i + lane) * 0.01f;
    f16v d[4];
    for (int j = 0; j < 4; ++j)
        d[j] = c;
    unsigned acc0 = 0, acc1 = 0, acc2 = 0;
    float run[8];
    for (int t = 0; t < 8; ++t)
        run[t] = 3e38f;
    float tmp[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
    const float s = 0x1p-60f;
    u6v r[2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 6; ++i)
            r[j][i] = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            if (MODE == 0) {
                asm volatile("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, %3" : "=&v"(r[(t >> 1) & 1]) : "v"(d[0]), "v"(d[1]), "v"(s));
                KEEP1(r[((t >> 1) + 1) & 1]);
            } else if (MODE == 1) {
                // two MFMAs (their results are consumed 2 pairs later), one cvt of the pair issued 2 pairs ago, 3 or3
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %7\n\t"
                             "v_mfma_f32_32x32x16_f16 %1, %4, %6, %7\n\t"
                             "v_cvt_scalef32_2xpk16_bf6_f32 %2, %8, %9, %10\n\t"
                             : "=&v"(d[t & 3]), "=&v"(d[(t + 1) & 3]), "=&v"(r[(t >> 1) & 1]), "+v"(acc0)
                             : "v"(a), "v"(q[t]), "v"(q[t + 1]), "v"(c), "v"(d[(t + 2) & 3]), "v"(d[(t + 3) & 3]), "v"(s));
                asm volatile("v_or3_b32 %0, %0, %3, %4\n\t"
                             "v_or3_b32 %1, %1, %5, %6\n\t"
                             "v_or3_b32 %2, %2, %7, %8"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2)
                             : "v"(r[((t >> 1) + 1) & 1][0]), "v"(r[((t >> 1) + 1) & 1][1]), "v"(r[((t >> 1) + 1) & 1][2]),
                               "v"(r[((t >> 1) + 1) & 1][3]), "v"(r[((t >> 1) + 1) & 1][4]), "v"(r[((t >> 1) + 1) & 1][5]));
            } else if (MODE == 2) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %5\n\t"
                             "v_mfma_f32_32x32x16_f16 %1, %2, %4, %5"
                             : "=&v"(d[t & 3]), "=&v"(d[(t + 1) & 3])
                             : "v"(a), "v"(q[t]), "v"(q[t + 1]), "v"(c));
                KEEP2(d[(t + 2) & 3], d[(t + 3) & 3]);
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f16v &x = d[(t + 2 + u) & 3];
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %7, %8, %9\n\t"
                                 "v_min3_f32 %1, %10, %11, %12\n\t" "v_min3_f32 %2, %13, %14, %15\n\t" "v_min3_f32 %3, %16, %17, %18\n\t"
                                 "v_min3_f32 %4, %19, %20, %21\n\t" "v_min3_f32 %5, %22, %23, %24\n\t" "v_min3_f32 %1, %1, %2, %3\n\t"
                                 "v_min3_f32 %4, %4, %5, %25\n\t" "v_min3_f32 %6, %1, %4, %6"
                                 : "=&v"(d[(t + u) & 3]), "+v"(tmp[u][0]), "+v"(tmp[u][1]), "+v"(tmp[u][2]), "+v"(tmp[u][3]),
                                   "+v"(tmp[u][4]), "+v"(run[t + u])
                                 : "v"(a), "v"(q[t + u]), "v"(c), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]),
                                   "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]),
                                   "v"(x[14]), "v"(x[15]));
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    unsigned sum = acc0 ^ acc1 ^ acc2;
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 6; ++i)
            sum ^= r[j][i];
    float f = 0.f;
    for (int t = 0; t < 8; ++t)
        f += run[t];
    for (int j = 0; j < 4; ++j)
        f += d[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum + (unsigned)f;
    if (lane == 0)
        stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int MODE>
static int cost(const char *name, const h8 *in, unsigned *out, unsigned long long *stamps, int units_per_iter)
{
    const int iters = 20000, blocks = 256;
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(k_cost<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters, stamps);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    std::vector<unsigned long long> hs(blocks * 4);
    CHK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (unsigned long long v : hs)
        cyc += (double)v;
    cyc /= hs.size();
    printf("%-64s %8.3f ms   %7.1f shader cycles per %s\n", name, best, cyc / ((double)iters * units_per_iter), MODE == 0 ? "cvt" : "MFMA (one tile pair)");
    return 0;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    // semantics
    std::vector<float> hin(64 * 32), hscale(64);
    const float probe[16] = {1.0f, -1.0f, 0.0f, -0.0f, 1e-3f, -1e-3f, 1e-30f, -1e-30f, 28.0f, -28.0f, 1e30f, -1e30f,
                             INFINITY, -INFINITY, NAN, 0.3f};
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < 32; ++i)
            hin[l * 32 + i] = l < 32 ? ((i == l) ? -1.0f : 1.0f)      // lanes 0..31: one negative value at position = lane
                                     : probe[(i + l) % 16];
        hscale[l] = l < 32 ? 1.0f : (l < 48 ? 1.0f : 0x1p-60f);
    }
    float *din, *dscale;
    unsigned *dout;
    CHK(hipMalloc(&din, hin.size() * 4));
    CHK(hipMalloc(&dscale, 64 * 4));
    CHK(hipMalloc(&dout, 64 * 6 * 4 + 256 * 256 * 4));
    CHK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dscale, hscale.data(), 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, din, dscale, dout);
    std::vector<unsigned> hout(64 * 6);
    CHK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
    printf("field layout (lanes 0..31 have -1.0 at input position = lane, +1.0 elsewhere, scale 1.0):\n");
    for (int l = 0; l < 32; ++l) {
        printf("  input %2d negative ->", l);
        for (int i = 0; i < 6; ++i)
            printf(" %08x", hout[l * 6 + i]);
        // which 6-bit field differs from the all-(+1.0) pattern?
        int where = -1;
        for (int f = 0; f < 32; ++f) {
            const int bit = f * 6;
            unsigned long long w = (unsigned long long)hout[l * 6 + bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? hout[l * 6 + bit / 32 + 1] : 0) << 32);
            const unsigned v = (unsigned)(w >> (bit % 32)) & 63u;
            if (v & 32u)
                where = f;
        }
        printf("   sign bit found in field %d\n", where);
    }
    printf("special values (lane 32+: inputs rotate through {1,-1,0,-0,1e-3,-1e-3,1e-30,-1e-30,28,-28,1e30,-1e30,inf,-inf,nan,0.3}):\n");
    for (int l = 32; l < 64; l += 15) {
        printf("  lane %d scale %g:", l, hscale[l]);
        for (int f = 0; f < 32; ++f) {
            const int bit = f * 6;
            unsigned long long w = (unsigned long long)hout[l * 6 + bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? hout[l * 6 + bit / 32 + 1] : 0) << 32);
            printf(" %g->%02x", hin[l * 32 + f], (unsigned)(w >> (bit % 32)) & 63u);
        }
        printf("\n");
    }
    // cost
    h8 *in;
    unsigned long long *stamps;
    CHK(hipMalloc(&in, 9 * 64 * 16));
    CHK(hipMemset(in, 0x3c, 9 * 64 * 16));
    CHK(hipMalloc(&stamps, 256 * 4 * 8));
    for (int round = 0; round < 2; ++round) {
        if (cost<0>("cvt_scalef32_2xpk16_bf6_f32 alone", in, dout + 64 * 6, stamps, 4)) return 1;
        if (cost<2>("2 MFMAs (fresh C in VGPRs)", in, dout + 64 * 6, stamps, 8)) return 1;
        if (cost<1>("per 2 MFMAs: 1 cvt + 3 v_or3_b32", in, dout + 64 * 6, stamps, 8)) return 1;
        if (cost<3>("per MFMA: 8 v_min3_f32 (today)", in, dout + 64 * 6, stamps, 8)) return 1;
    }
    return 0;
}
