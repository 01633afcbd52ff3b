// ta_harness.cpp — a TA-style harness written for this repository's tests (it is NOT the
// reference's main.cu): it drives the operator boundary exactly the way the reference's harness
// does — a function pointer of the cudaCallback type chosen by the CALLBACKn macros, host
// malloc'd inputs freed by the caller, results malloc'd by the callee and freed by the caller —
// over the same eight sample shapes and seed, and prints the same two kinds of line
// ("CallbackN, k, m, n, ms" and "errors/total w.r.t. baseline: e/m") so logs are comparable with
// the reference's screen.log.  It also writes a file in the layout of the reference's results.csv:
// per sample one line of nearest indices and one line of "%.3f" distances.
//
// v0 (CALLBACK1, the CPU baseline) comes from the CPU oracle: test infrastructure, never the product.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "../../include/ta_compat.h"
#include "../../oracle/knn_oracle.h"

int divup(int n, int m) { return (n + m - 1) / m; }   // the harness owns this symbol

namespace v0 {
void cudaCallback(KNN_TA_SIGNATURE)
{
    int *out = (int *)malloc(sizeof(int) * (size_t)m);
    knn_oracle_v0(k, m, n, searchPoints, referencePoints, out);
    *results = out;
}
}  // namespace v0

typedef void (*callback_t)(int, int, int, float *, float *, int **);

static const int kShapes[][3] = {{3, 1, 2},      {3, 2, 8},       {3, 1, 1024},     {3, 1, 65536},
                                 {16, 1, 65536}, {3, 1024, 1024}, {3, 1024, 65536}, {16, 1024, 65536}};
static const int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);
static int *g_baseline[kNumShapes];

static double now_ms(void)
{
    struct timespec ts;
    timespec_get(&ts, TIME_UTC);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

static float euclid(int k, const float *q, const float *r)
{
    float s = 0.f;
    for (int i = 0; i < k; ++i) {
        const float d = q[i] - r[i];
        s += d * d;
    }
    return sqrtf(s);
}

static int run_all(int slot, callback_t fn, FILE *csv)
{
    int total_errors = 0;
    ta_srand(1000);
    for (int i = 0; i < kNumShapes; ++i) {
        const int k = kShapes[i][0], m = kShapes[i][1], n = kShapes[i][2];
        float *q = (float *)malloc(sizeof(float) * (size_t)k * m);
        float *r = (float *)malloc(sizeof(float) * (size_t)k * n);
        ta_get_sample(k, m, n, q, r);
        int *res = NULL;
        const double t0 = now_ms();
        fn(k, m, n, q, r, &res);
        const double t1 = now_ms();
        printf("Callback%d, %2d, %4d, %5d, %10.3fms\n", slot, k, m, n, t1 - t0);
        if (!g_baseline[i]) {
            g_baseline[i] = res;
        } else {
            int errors = 0;
            for (int j = 0; j < m; ++j) {
                if (g_baseline[i][j] == res[j])
                    continue;
                const float d1 = euclid(k, q + (size_t)j * k, r + (size_t)g_baseline[i][j] * k);
                const float d2 = euclid(k, q + (size_t)j * k, r + (size_t)res[j] * k);
                if (fabsf(d1 - d2) > 1e-3f)
                    ++errors;
            }
            printf("errors/total w.r.t. baseline: %d/%d\n\n", errors, m);
            total_errors += errors;
            if (csv) {   // results.csv layout: one line of indices, one line of "%.3f" distances per sample
                for (int j = 0; j < m; ++j)
                    fprintf(csv, "%d,", res[j]);
                fprintf(csv, "\n");
                for (int j = 0; j < m; ++j)
                    fprintf(csv, "%.3f,", euclid(k, q + (size_t)j * k, r + (size_t)res[j] * k));
                fprintf(csv, "\n");
            }
            free(res);
        }
        free(q);   // inputs are released only after the comparison (the reference frees them
        free(r);   // before it, main.cu:76-77 — a latent use-after-free this harness does not copy)
    }
    return total_errors;
}

int main(int argc, char **argv)
{
    FILE *csv = argc > 1 ? fopen(argv[1], "w") : NULL;
    int bad = 0;
    bad += run_all(1, &CALLBACK1, NULL);    // baseline
    bad += run_all(10, &CALLBACK10, csv);   // v8 -> libknn_mi355x.so
    bad += run_all(2, &CALLBACK2, NULL);    // any other slot forwards to the same path
    if (csv)
        fclose(csv);
    for (int i = 0; i < kNumShapes; ++i)
        free(g_baseline[i]);
    printf("total errors: %d\n", bad);
    return bad ? 1 : 0;
}
