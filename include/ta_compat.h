/*
 * ta_compat.h — source compatibility with the reference's operator header for TA-style
 * harnesses.  A harness written against the reference selects an
 * implementation through the macros CALLBACK1..CALLBACK10, each naming a function
 *     void vN::cudaCallback(int k, int m, int n, float *searchPoints, float *referencePoints, int **results)
 * (reference sources/src/core.h:12-21, used by sources/src/main.cu:113-171).  This header
 * declares the same names without any CUDA/HIP/thrust include; libknn_mi355x.so defines v1..v9
 * (C++ linkage) as forwarders to its MI355X path.  v0 — the serial CPU baseline a
 * harness compares against (core.cu:27-62) — is deliberately NOT defined by the product: the
 * harness supplies its own (tests/harness/ links the CPU oracle for it).
 *
 * divup(int,int) is only declared, as in the reference (core.h:74): the harness owns its
 * definition (utils.h:11) and the product exports no such symbol.
 */
#ifndef KNN_TA_COMPAT_H
#define KNN_TA_COMPAT_H

#include <stdio.h>
#include <stdlib.h>

#define KNN_TA_SIGNATURE int k, int m, int n, float *searchPoints, float *referencePoints, int **results
#define KNN_TA_NAMESPACE(ns) \
    namespace ns {           \
    void cudaCallback(KNN_TA_SIGNATURE); \
    }

KNN_TA_NAMESPACE(v0)
KNN_TA_NAMESPACE(v1)
KNN_TA_NAMESPACE(v2)
KNN_TA_NAMESPACE(v3)
KNN_TA_NAMESPACE(v4)
KNN_TA_NAMESPACE(v5)
KNN_TA_NAMESPACE(v6)
KNN_TA_NAMESPACE(v7)
KNN_TA_NAMESPACE(v8)
KNN_TA_NAMESPACE(v9)

extern "C" void cudaCallback(KNN_TA_SIGNATURE); /* libknn_mi355x.so (include/knn_mi355x.h) */

/* the reference's numbering: 1 = baseline, 10 = "the best" */
#define CALLBACK1 v0::cudaCallback
#define CALLBACK2 v1::cudaCallback
#define CALLBACK3 v2::cudaCallback
#define CALLBACK4 v3::cudaCallback
#define CALLBACK5 v4::cudaCallback
#define CALLBACK6 v5::cudaCallback
#define CALLBACK7 v6::cudaCallback
#define CALLBACK8 v7::cudaCallback
#define CALLBACK9 v9::cudaCallback
#define CALLBACK10 v8::cudaCallback

int divup(int n, int m);

/* error-check macro with the reference's print-and-exit behaviour (core.h:77-87), for HIP */
#define CHECK(call)                                                                   \
    do {                                                                              \
        const int knn_ta_err_ = (int)(call);                                          \
        if (knn_ta_err_ != 0) {                                                       \
            printf("Error: %s:%d, code:%d \n", __FILE__, __LINE__, knn_ta_err_);      \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

#endif
