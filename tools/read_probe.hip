// read_probe.hip — what HBM read rate does the cell scan's access pattern allow?  Every wave reads whole
// "cells" (T KiB contiguous as T wave-wide 16-byte loads issued together, then waits for all of them), wave w
// taking cells w, w + W, ... of a 512 MiB buffer; W = waves per CU x 256.  The grid-stride stream (one load in
// flight per wave, many waves) is the reference line.
//   hipcc -O3 --offload-arch=gfx950 tools/read_probe.hip -o tools/read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int T>
__global__ __launch_bounds__(256) void cells_read(const f4 *__restrict__ buf, long long ncells, float *out)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    f4 acc = {0, 0, 0, 0};
    for (long long c = wave; c < ncells; c += nw) {
        f4 v[T];
#pragma unroll
        for (int p = 0; p < T; ++p)
            v[p] = __builtin_nontemporal_load(&buf[(c * T + p) * 64 + lane]);
#pragma unroll
        for (int p = 0; p < T; ++p)
            acc += v[p];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f)
        out[0] = acc[0];
}

int main()
{
    const size_t bytes = 576ull << 20;
    f4 *buf; float *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto kernel, int T, int blocks) {
        const long long ncells = (long long)(bytes / (1024ull * T));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, buf, ncells, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-34s blocks %5d: %.1f us  %.2f TB/s\n", name, blocks, best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
    for (int bpc : {2, 4, 6, 8}) {
        run("1 KiB per wave request", cells_read<1>, 1, 256 * bpc);
        run("4 KiB per wave request", cells_read<4>, 4, 256 * bpc);
        run("9 KiB per wave request", cells_read<9>, 9, 256 * bpc);
        run("16 KiB per wave request", cells_read<16>, 16, 256 * bpc);
    }
    return 0;
}
