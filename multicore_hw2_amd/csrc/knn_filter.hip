// knn_filter.hip — MFMA low-precision filter (placeholder until the filter path lands).
#include "knn_common.h"
