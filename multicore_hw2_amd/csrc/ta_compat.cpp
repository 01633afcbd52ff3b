// ta_compat.cpp — v1..v9 of include/ta_compat.h as forwarders to the MI355X path.  In the
// reference these are ten different implementations of one operator (core.cu:64-1050); a harness
// built against this library gets the same answers from every slot.  v0 is not defined here.
#include "../../include/ta_compat.h"

#define KNN_TA_FORWARD(ns)                                                   \
    namespace ns {                                                           \
    void cudaCallback(KNN_TA_SIGNATURE)                                      \
    {                                                                        \
        ::cudaCallback(k, m, n, searchPoints, referencePoints, results);     \
    }                                                                        \
    }

KNN_TA_FORWARD(v1)
KNN_TA_FORWARD(v2)
KNN_TA_FORWARD(v3)
KNN_TA_FORWARD(v4)
KNN_TA_FORWARD(v5)
KNN_TA_FORWARD(v6)
KNN_TA_FORWARD(v7)
KNN_TA_FORWARD(v8)
KNN_TA_FORWARD(v9)
