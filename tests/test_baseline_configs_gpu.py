"""BASELINE.json's configurations at their FULL shapes, through the C-ABI, on the GPU (C3 lives in
test_parity_gpu.py::test_headline_shape_properties; C1 is TA sample 2 of the golden file):

  C2  k=3   m=1024   n=2^20   every query against the full oracle
  C4  k=16  m=1024   n=2^27   on ONE GPU as 8 shards of 2^24 folded into one key array (the 8-rank
                              partition of bench.py / core.cu:873-883): sampled oracle + planted copies
  C5  k=128 m=n=65536         2048 seeded queries against the OpenMP oracle + planted copies for all
  and the N > 1 exchange step of bench.py (grouped RCCL all-reduce) rehearsed with a 1-rank nccl group.
Bar: bit-exact nearest indices."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch  # before libknn_mi355x.so is dlopen'ed: one HIP runtime per process

import multicore_hw2_amd as pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THREADS = 16   # a one-GPU box's CPU share


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert os.path.exists(pkg.lib_path), "libknn_mi355x.so not built (no CPU fallback exists)"
    assert pkg.device_count() >= 1, "no GPU visible to libknn_mi355x.so"
    yield
    for name in ("path", "shards", "stream"):
        pkg.set_option(name, 0)
    pkg.trim()


def _resident_query(k, m, q_d, shards, n, r_d, stream):
    """Indices from device-resident shards folding into one key array (bench.py's data path)."""
    dev = q_d.device
    keys = torch.empty(m, dtype=torch.int64, device=dev)
    out = torch.empty(m, dtype=torch.int32, device=dev)
    pkg.keys_init(keys.data_ptr(), m, stream=stream)
    taken = []
    for lo, hi in pkg.shard_bounds(n, shards):
        ix = pkg.KnnIndex(k, r_d.data_ptr() + lo * k * 4, n_local=hi - lo, base_index=lo, refs_on_device=True,
                          stream=stream)
        ix.query_keys(m, q_d.data_ptr(), keys.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        taken.append(ix.last_stats()[0])
        ix.close()
    pkg.keys_to_indices(keys.data_ptr(), m, out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    return out.cpu().numpy().copy(), taken


@pytest.mark.parametrize("path", [1, 2, 0], ids=["exact", "filter", "auto"])
def test_c2_full_shape_every_query_against_the_oracle(oracle, path):
    k, m, n = 3, 1024, 1 << 20
    Q, R = oracle.synth(m * k, 1000), oracle.synth(n * k, 1001)
    want = oracle.v0(k, Q, R, threads=THREADS)
    pkg.set_option("path", path)
    try:
        np.testing.assert_array_equal(pkg.cudaCallback(k, m, n, Q, R), want, err_msg="drop-in call")
        dev = torch.device("cuda:0")
        q_d, r_d = torch.from_numpy(Q).to(dev), torch.from_numpy(R).to(dev)
        got, taken = _resident_query(k, m, q_d, 1, n, r_d, torch.cuda.current_stream().cuda_stream)
        np.testing.assert_array_equal(got, want, err_msg="resident index")
        if path:
            assert taken == [4 if path == 2 else path]              # 4 = the filter's cell-pruned form (resident index)
    finally:
        pkg.set_option("path", 0)


def test_c5_full_shape_sampled_oracle_and_planted_copies(oracle):
    k, m, n = 128, 65536, 65536
    Q, R = oracle.synth(m * k, 1000), oracle.synth(n * k, 1001)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    q_d, r_d = torch.from_numpy(Q).to(dev), torch.from_numpy(R).to(dev)
    got, taken = _resident_query(k, m, q_d, 1, n, r_d, stream)
    assert taken == [2]                               # the deep-K MFMA filter is the path under test
    sel = np.random.default_rng(5).choice(m, 2048, replace=False)
    want = oracle.v0(k, Q.reshape(m, k)[sel], R, threads=THREADS)
    np.testing.assert_array_equal(got[sel], want)
    # the drop-in entry on the same data (host arrays; one shard per visible GPU)
    np.testing.assert_array_equal(pkg.cudaCallback(k, m, n, Q, R)[sel], want)
    # every query gets an exact copy somewhere in the reference set: that row must be its answer
    pos = np.random.default_rng(6).permutation(n)[:m]
    r_d.view(n, k)[torch.from_numpy(pos).to(dev)] = q_d.view(m, k)
    planted, _ = _resident_query(k, m, q_d, 1, n, r_d, stream)
    np.testing.assert_array_equal(planted, pos.astype(np.int32))


def test_c4_full_shape_on_one_gpu_as_eight_folded_shards(oracle):
    k, m, n, shards = 16, 1024, 1 << 27, 8
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    r_d = torch.empty(n * k, dtype=torch.float32, device=dev)       # 8 GiB
    q_d = torch.empty(m * k, dtype=torch.float32, device=dev)
    pkg.synth_fill_device(r_d.data_ptr(), n * k, 1001, stream=stream)
    pkg.synth_fill_device(q_d.data_ptr(), m * k, 1000, stream=stream)
    got, taken = _resident_query(k, m, q_d, shards, n, r_d, stream)
    assert taken == [4] * shards                         # every shard of 2^24 rows: the cell-pruned filter
    # 16 seeded queries against all 2^27 references on the host (3.4e10 multiply-adds)
    R = r_d.cpu().numpy()
    Q = q_d.cpu().numpy().reshape(m, k)
    sel = np.random.default_rng(7).choice(m, 16, replace=False)
    want = oracle.v0(k, Q[sel], R, threads=THREADS)
    del R
    np.testing.assert_array_equal(got[sel], want)
    assert len(set((want >> 24).tolist())) > 1        # the sampled answers really come from several shards
    # planted copies, spread over all eight shards (global indices beyond 2^24 .. 2^27)
    pos = np.sort(np.random.default_rng(8).choice(n, m, replace=False))
    r_d.view(n, k)[torch.from_numpy(pos).to(dev)] = q_d.view(m, k)
    planted, _ = _resident_query(k, m, q_d, shards, n, r_d, stream)
    np.testing.assert_array_equal(planted, pos.astype(np.int32))
    assert pos.max() >= (7 << 24)
    # the same set as ONE shard of 2^27 rows (64-bit offsets: k n = 2^31 floats)
    whole, _ = _resident_query(k, m, q_d, 1, n, r_d, stream)
    np.testing.assert_array_equal(whole, planted)


def test_bench_exchange_step_rehearsed_with_a_one_rank_rccl_group():
    """bench.py's N > 1 code (nccl process group, grouped all-reduce MIN of the batches' packed keys on
    its own stream, unpack, parity spot check against the oracle) run as ONE rank: everything but a
    second GPU.  The reduced indices must pass the oracle check and the line must say so."""
    env = dict(os.environ, KNN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(name, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "16,1024,2097152", "--steps", "30",
                        "--warmup", "3", "--cpu-queries", "8"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    doc = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert doc["n_gpus"] == 1 and doc["value"] > 0
    assert "rccl all_reduce(min)" in doc["config"]["collective"]
    assert doc["parity_spot_check"].startswith("8/8 sampled queries identical")


@pytest.mark.parametrize("shard", ["index", "cells"])
def test_two_rank_bench_flow_rehearsed_on_one_gpu(shard):
    """The N = 2 flow of bench.py end to end — self-launch, one rank per shard of the SAME global set, four batches in flight
    per rank, grouped MIN all-reduce of the packed keys, unpack, oracle spot check of the reduced indices — with both ranks
    on GPU 0 and gloo as the transport (RCCL refuses two ranks on one device).  What the 1-rank rehearsal cannot see:
    index: shard offsets (rank 1's index carries base_index = n/2) and a real reduction;
    cells: the gathered sample -> one grid on both ranks, the all-to-all that moves every row (and its global number) to the
           rank whose cell range holds it, the gathered seed layer, keys that carry global numbers when the batch ends."""
    env = dict(os.environ, KNN_BENCH_REHEARSE_ON_ONE_GPU="1")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KNN_BENCH_FORCE_DIST"):
        env.pop(name, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "16,1024,2097152",
                        "--steps", "20", "--warmup", "2", "--cpu-queries", "8", "--shard", shard], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["scaling"] == "strong"
    if shard == "index":
        assert doc["config"]["n_per_gpu"] == 1 << 20 and "index ranges" in doc["config"]["shards"]
    else:
        assert abs(doc["config"]["n_per_gpu"] - (1 << 20)) < (1 << 20) * 0.05         # the grid's two halves: equal shares, nearly
        assert "cell ranges of one global grid of 2^13 cells" in doc["config"]["shards"]["partition"]
    assert "all_reduce(min)" in doc["config"]["collective"]
    assert doc["parity_spot_check"].startswith("8/8 sampled queries identical")


def test_two_rank_flow_falls_back_to_index_ranges_when_one_rank_fails_before_the_exchange():
    """the guarded fallback from cell-range to index-range shards must work when ONE rank fails — its peers
    must not already be waiting in the all-to-all.  The partition's local phase (owner of every row, sort by destination)
    is agreed on with a MIN all-reduce BEFORE any rank enters the all-to-alls; a rank that fails there (test hook) takes
    every rank to the index-range shards it already holds, and the run ends with the same answers."""
    env = dict(os.environ, KNN_BENCH_REHEARSE_ON_ONE_GPU="1", KNN_BENCH_TEST_PARTITION_FAIL_RANK="1")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KNN_BENCH_FORCE_DIST"):
        env.pop(name, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "16,1024,2097152",
                        "--steps", "8", "--warmup", "2", "--cpu-queries", "8"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    doc = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "index ranges" in doc["config"]["shards"] and doc["config"]["n_per_gpu"] == 1 << 20
    assert "index-range shards instead" in r.stderr
    assert doc["parity_spot_check"].startswith("8/8 sampled queries identical")
