// read_probe2.hip — which part of the cell scan's memory behaviour costs the 15-20 us over the plain stream?
// Variants of the 9-KiB-per-cell reader of read_probe.hip with the scan kernel's extras added one at a time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define T 9

template <int WAVES, bool LDSFILL, bool SMALL, bool META>
__global__ __launch_bounds__(64 * WAVES) void cells_read(const f4 *__restrict__ buf, const f4 *__restrict__ nrm,
                                                          const unsigned short *__restrict__ lists, const unsigned *__restrict__ cnts,
                                                          const f4 *__restrict__ qf, long long ncells, float *out)
{
    extern __shared__ f4 s_q[];
    const int lane = threadIdx.x & 63;
    if (LDSFILL) {
        for (int i = threadIdx.x; i < 2304; i += 64 * WAVES)   // 36 KiB
            s_q[i] = qf[i];
        __syncthreads();
    }
    const long long wave = (long long)blockIdx.x * WAVES + (threadIdx.x >> 6), nw = (long long)gridDim.x * WAVES;
    f4 acc = {0, 0, 0, 0};
    unsigned meta = 1;
    if (META) {
        const long long mine = (long long)lane * nw + wave;
        meta = mine < ncells ? cnts[mine] : 0u;
    }
    for (long long c = wave; c < ncells; c += nw) {
        f4 v[T];
#pragma unroll
        for (int p = 0; p < T; ++p)
            v[p] = __builtin_nontemporal_load(&buf[(c * T + p) * 64 + lane]);
        f4 n0 = {0, 0, 0, 0}, n1 = {0, 0, 0, 0};
        unsigned l0 = 0;
        if (SMALL) {
            n0 = __builtin_nontemporal_load(&nrm[c * T * 8 + lane]);
            if (lane < T * 8 - 64)
                n1 = __builtin_nontemporal_load(&nrm[c * T * 8 + 64 + lane]);
            l0 = lists[c * 128 + (lane & 31)];
        }
#pragma unroll
        for (int p = 0; p < T; ++p)
            acc += v[p];
        acc += n0 + n1;
        acc[0] += (float)l0 + (float)meta;
        if (LDSFILL)
            acc[1] += s_q[(l0 * 2) % 2304][0];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f)
        out[0] = acc[0];
}

int main()
{
    const size_t bytes = 512ull << 20;
    const long long ncells = (long long)(bytes / (1024ull * T));
    f4 *buf, *nrm, *qf; float *out; unsigned short *lists; unsigned *cnts;
    hipMalloc(&buf, bytes); hipMalloc(&nrm, ncells * T * 128); hipMalloc(&lists, ncells * 256); hipMalloc(&cnts, ncells * 4);
    hipMalloc(&qf, 36864); hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes); hipMemset(nrm, 0, ncells * T * 128); hipMemset(lists, 0, ncells * 256); hipMemset(cnts, 0, ncells * 4); hipMemset(qf, 0, 36864);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto kernel, int waves, int blocks, size_t lds, double mb) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * waves), lds, 0, buf, nrm, lists, cnts, qf, ncells, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-64s %.1f us  %.2f TB/s\n", name, best * 1e3, mb * 1048576.0 / (best * 1e-3) / 1e12);
    };
    const double mb_big = 512.0, mb_all = 512.0 + 64.0 + 4.0;
    run("tiles only, 4-wave blocks x 1536", cells_read<4, false, false, false>, 4, 1536, 0, mb_big);
    run("tiles only, 12-wave blocks x 512", cells_read<12, false, false, false>, 12, 512, 0, mb_big);
    run("+ 36 KiB LDS fill per block", cells_read<12, true, false, false>, 12, 512, 36864, mb_big);
    run("+ norms (1152 B) and list (64 B) per cell", cells_read<12, true, true, false>, 12, 512, 36864, mb_all);
    run("+ strided metadata gather first", cells_read<12, true, true, true>, 12, 512, 36864, mb_all);
    run("norms + list, no LDS fill", cells_read<12, false, true, false>, 12, 512, 0, mb_all);
    return 0;
}
