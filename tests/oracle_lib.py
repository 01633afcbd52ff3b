"""ctypes binding of oracle/libknn_oracle.so — the CPU checker (see oracle/knn_oracle.h).
Test infrastructure only."""
import ctypes

import numpy as np

TA_SAMPLES = [(3, 1, 2), (3, 2, 8), (3, 1, 1024), (3, 1, 65536), (16, 1, 65536),
              (3, 1024, 1024), (3, 1024, 65536), (16, 1024, 65536)]  # reference main.cu:28-39
TA_SEED = 1000  # reference main.cu:43


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Oracle:
    def __init__(self, path):
        L = ctypes.CDLL(path)
        c_int, c_ll, c_vp = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p
        L.knn_oracle_v0.argtypes = [c_int, c_int, c_ll, c_vp, c_vp, c_vp]
        L.knn_oracle_v0_range.argtypes = [c_int, c_int, c_int, c_ll, c_vp, c_vp, c_vp, c_int]
        L.knn_oracle_v0_range.restype = c_int
        L.knn_oracle_v0_keys.argtypes = [c_int, c_int, c_ll, c_vp, c_vp, c_ll, c_vp, c_int]
        L.knn_oracle_dist2.argtypes = [c_int, c_vp, c_vp]
        L.knn_oracle_dist2.restype = ctypes.c_float
        L.ta_srand.argtypes = [ctypes.c_uint]
        L.ta_rand.restype = c_int
        L.ta_get_sample.argtypes = [c_int, c_int, c_int, c_vp, c_vp]
        L.knn_synth_fill.argtypes = [c_vp, c_ll, ctypes.c_ulonglong, c_ll]
        self.L = L

    # -- v0 ---------------------------------------------------------------
    def v0(self, k, Q, R, threads=0):
        """Nearest index per query, v0 arithmetic; OpenMP over queries (bit-identical)."""
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1)
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(-1)
        m, n = Q.size // k, R.size // k
        out = np.empty(m, dtype=np.int32)
        self.L.knn_oracle_v0_range(k, 0, m, n, _p(Q), _p(R), _p(out), threads)
        return out

    def v0_serial(self, k, Q, R):
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1)
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(-1)
        m, n = Q.size // k, R.size // k
        out = np.empty(m, dtype=np.int32)
        self.L.knn_oracle_v0(k, m, n, _p(Q), _p(R), _p(out))
        return out

    def v0_keys(self, k, Q, R_shard, base=0, threads=0):
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1)
        R = np.ascontiguousarray(R_shard, dtype=np.float32).reshape(-1)
        m, n = Q.size // k, R.size // k
        keys = np.empty(m, dtype=np.uint64)
        self.L.knn_oracle_v0_keys(k, m, n, _p(Q), _p(R), base, _p(keys), threads)
        return keys

    def dist2(self, q, r):
        q = np.ascontiguousarray(q, dtype=np.float32)
        r = np.ascontiguousarray(r, dtype=np.float32)
        return float(self.L.knn_oracle_dist2(q.size, _p(q), _p(r)))

    # -- inputs -----------------------------------------------------------
    def ta_samples(self, upto=len(TA_SAMPLES)):
        """The TA samples of main.cu:28-39 drawn sequentially from one rand() stream seeded
        with 1000 (generator.h:32-50): yields (k, m, n, Q, R)."""
        self.L.ta_srand(TA_SEED)
        for (k, m, n) in TA_SAMPLES[:upto]:
            Q = np.empty(k * m, dtype=np.float32)
            R = np.empty(k * n, dtype=np.float32)
            self.L.ta_get_sample(k, m, n, _p(Q), _p(R))
            yield k, m, n, Q, R

    def synth(self, count, seed, first=0):
        x = np.empty(count, dtype=np.float32)
        self.L.knn_synth_fill(_p(x), count, seed, first)
        return x
