#!/usr/bin/env python3
"""bench.py — queries/sec of the brute-force 1-NN hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: m queries (already in
HBM) against the device-resident reference set -> int32 nearest indices in HBM.  With N GPUs the
reference set of the SAME workload is split into N contiguous shards (one rank per GPU, the
scheme of reference core.cu:873-883); each rank scans its shard and the packed (distance, index)
keys are min-reduced with one RCCL all-reduce — strong scaling at the metric's fixed n.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|k,m,n]

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the
dominant kernel (HIP events recorded by the library around that kernel, on the stream it is
launched on) and `cpu_baseline` (the serial CPU oracle on a bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs: name -> (k, m, n)
    "c2": (3, 1024, 1 << 20),
    "c3": (16, 1024, 1 << 24),   # the configuration the metric is quoted on
    "c4": (16, 1024, 1 << 27),
    "c5": (128, 65536, 65536),   # dense -2 Q R^T contraction: MFMA utilisation is the figure of merit
}
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
SETUP_STEPS = 30
SETUP_MS = 40.0         # untimed steady work in front of the warm-up steps (see main)
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense bf16/f16 MFMA
VALU_LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD32 x 2.4 GHz = 78.6e12 lane-ops/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", help="c2 | c3 | c4 | c5 | k,m,n")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 exact VALU only, 2 force MFMA filter")
    ap.add_argument("--filter-qt", type=int, default=0, help="tuning: query tiles per filter wave (0 auto)")
    ap.add_argument("--filter-rounds", type=int, default=0, help="tuning: filter blocks per resident slot (0 default)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="batches in flight (1..8), each on its own stream / workspace slot; 0 = 2 for full scans of "
                         ">= 16M references (chained), 4 on the cell-pruned path, 3 otherwise (free to overlap)")
    ap.add_argument("--filter-chain", type=int, default=0, help="0 auto, 1 scans of different slots chained, 2 free")
    ap.add_argument("--time-every", type=int, default=8,
                    help="bracket every N-th launch of the dominant kernel with HIP events (roofline.kernel_avg_ms)")
    ap.add_argument("--cells", type=int, default=0, help="A/B: cell-sorted layouts, 0 = library policy, 1 = always (k <= 16), 2 = never")
    ap.add_argument("--separate-init", action="store_true",
                    help="A/B: start the keys with a knn_keys_init launch per step instead of KNN_QUERY_INIT_KEYS")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B: any other library option (knn_set_option), e.g. --opt scan_deal=2 --opt scan_blocks=1")
    ap.add_argument("--serial", action="store_true", help="one batch in flight (no overlap of consecutive steps)")
    ap.add_argument("--shard", choices=["auto", "cells", "index"], default="auto",
                    help="N > 1: how the reference set is split over the ranks.  index = contiguous index ranges (reference "
                         "core.cu:875-883: every rank grids its own n / N rows); cells = contiguous ranges of the cell codes of ONE "
                         "global grid (knn_geom_*: the rows move to their ranks by an all-to-all at index build, the ranks' scans "
                         "add up to one GPU's); auto = cells where the pruned path serves the shape (k <= 16, enough rows per rank)")
    ap.add_argument("--seed-tiles", type=int, default=0, help="cell-range shards: tiles of every cell in the replicated seed layer (0 = library default)")
    ap.add_argument("--emulate", default="", metavar="N[:r]",
                    help="one GPU plays rank r (default 0) of an N-rank cell-range sharded run: the whole set is generated here, "
                         "all N shards are built (their seed layer is needed), rank r's step is timed — no collective.  For the "
                         "per-rank projections under profiles/ when no multi-GPU node is at hand")
    ap.add_argument("--group", type=int, default=0,
                    help="N > 1: batches whose keys ONE all-reduce folds (0 = 16, or the batches in flight when that is more): "
                         "the collective's host cost (~50 us per call through torch.distributed) is shared by the group")
    ap.add_argument("--scan-stamps", default="", metavar="FILE.npz",
                    help="development: with a library built by tools/build_timeline_lib.sh (KNN_MI355X_LIB), save the "
                         "per-wave wall-clock stamps of the last pruned-scan launch, pipelined and one batch at a time")
    ap.add_argument("--cpu-queries", type=int, default=-1,
                    help="queries in the cpu_baseline sample (-1: sized for ~15 s, 0: skip)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="launcher rehearsal without GPUs: the ranks only form a gloo group, min-reduce packed keys "
                         "and rank 0 prints one JSON line (tests/test_host_logic.py)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 outside torchrun: start one rank per GPU with
    torch.distributed.run as a CHILD process (the parent never touches a GPU and never execs after
    one has been initialised), relay rank 0's single JSON line, return the launcher's exit code."""
    import socket
    import subprocess
    import torch
    if not args.selftest_launcher and not os.environ.get("KNN_BENCH_REHEARSE_ON_ONE_GPU"):
        have = torch.cuda.device_count()      # counting devices does not initialise the GPU
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this node: nothing was run\n"
                             % (args.gpus, have))
            return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith("{"):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write("bench.py: a rank failed (torch.distributed.run exit code %d)\n" % rc)
        return rc
    if line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        return 4
    print(line, flush=True)
    return 0


def selftest_launcher():
    """What the N > 1 path does around its kernels, on the CPU: gloo group, MIN all-reduce of packed
    (distance bits << 32 | index) keys viewed as int64, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("KNN_BENCH_SELFTEST_FAIL_RANK") == str(rank):
        raise SystemExit(7)
    # rank r holds distance r + 1 for query 0 and distance world - r for query 1
    keys = torch.tensor([((rank + 1) << 32) | (100 + rank), ((world - rank) << 32) | (200 + rank)], dtype=torch.int64)
    dist.all_reduce(keys, op=dist.ReduceOp.MIN)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "keys": keys.tolist()}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse_args()
    # Batches in flight run on streams of their own; the HIP runtime maps streams onto 4 hardware queues by default, and
    # with four batch streams plus the default stream two of them then share a queue and serialise (4 in flight: 0.053 ms
    # per step at n_local 2^21 against 0.044 for 3).  Eight queues: 4 in flight 0.042.  Must be set before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # (ranks started by an outside torchrun: the host driver only supports dmabuf IPC, RCCL fails without this)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.selftest_launcher:
        return selftest_launcher()
    import numpy as np
    import torch
    import multicore_hw2_amd as pkg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world     # launched by torchrun: the world size is the number of GPUs
    if args.workload in WORKLOADS:
        k, m, n = WORKLOADS[args.workload]
        wname = args.workload.upper()
    else:
        k, m, n = (int(t) for t in args.workload.split(","))
        wname = "custom"

    if not torch.cuda.is_available() or pkg.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # KNN_BENCH_REHEARSE_ON_ONE_GPU=1 (tests): every rank uses GPU 0 and the collective goes through gloo — RCCL
    # refuses two ranks on one device.  Everything else of the N > 1 flow (shard bounds and base indices per rank,
    # grouped all-reduce of the batches' keys, unpack, parity check of the reduced result) is the real code.
    rehearse = os.environ.get("KNN_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Native libraries (RCCL prints a version banner at communicator creation) write to fd 1:
    # keep stdout clean for the ONE JSON line by pointing fd 1 at stderr until it is printed.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1 or os.environ.get("KNN_BENCH_FORCE_DIST") == "1":   # the env hook rehearses the N>1 code on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    pkg.set_option("path", args.path)
    for item in args.opt:
        name, _, value = item.partition("=")
        pkg.set_option(name, int(value))
    if args.filter_qt:
        pkg.set_option("filter_qt", args.filter_qt)
    if args.filter_rounds:
        pkg.set_option("filter_rounds", args.filter_rounds)
    if args.filter_chain:
        pkg.set_option("filter_chain", args.filter_chain)
    if args.cells:
        pkg.set_option("cells", args.cells)

    stream = torch.cuda.current_stream().cuda_stream
    emu_n, emu_r = 0, 0
    if args.emulate:
        parts = args.emulate.split(":")
        emu_n, emu_r = int(parts[0]), int(parts[1]) if len(parts) > 1 else 0
        assert world == 1 and 1 <= emu_n <= 64 and 0 <= emu_r < emu_n, "--emulate N[:r] runs on ONE GPU"
    nshards = emu_n if emu_n else world
    lo, hi = pkg.shard_bounds(n, world)[rank] if rank < len(pkg.shard_bounds(n, world)) else (n, n)
    n_local = hi - lo
    # synthetic uniform [0,1) inputs generated on the device (counter-based: each rank fills its
    # own slice of the same global reference set); same values as oracle/knn_synth_fill
    r_d = torch.empty(max(n_local, 1) * k, dtype=torch.float32, device=dev)
    q_d = torch.empty(m * k, dtype=torch.float32, device=dev)
    pkg.synth_fill_device(r_d.data_ptr(), n_local * k, 1001, first=lo * k, device=local_rank, stream=stream)
    pkg.synth_fill_device(q_d.data_ptr(), m * k, 1000, device=local_rank, stream=stream)
    # ---- N > 1 (or --emulate): cell-range shards.  Every rank holds rows [lo, hi) of the global set as generated; they move
    # to the rank whose range of the global grid's cell codes they fall into (index build: untimed, like the layouts).
    shard_mode = "index"
    cells_ok = k <= 16 and pkg.get_option("path") in (0, 2) and pkg.get_option("cells") != 2
    if nshards > 1 and args.shard != "index" and cells_ok:
        shard_mode = "cells"
    geom, gids_d, layer_d, shard_note = None, None, None, None
    if shard_mode == "cells":
        torch.cuda.synchronize()
        t_part = time.perf_counter()
        rows2d = r_d[: n_local * k].reshape(n_local, k)
        # 1. ONE grid from a sample of the global set: every rank's strided sample, gathered (identical input on every rank)
        per_rank = 8192 if emu_n == 0 else 8192 * emu_n
        samp = rows2d[:: max(1, n_local // per_rank)][:per_rank].contiguous()
        if dist is not None:
            gathered = [torch.empty_like(samp) for _ in range(world)]
            if rehearse:
                gl = [torch.empty(samp.shape, dtype=samp.dtype) for _ in range(world)]
                dist.all_gather(gl, samp.cpu())
                gathered = gl
            else:
                dist.all_gather(gathered, samp)
            samp = torch.cat([g_.to("cpu") for g_ in gathered])
        try:
            geom = pkg.KnnGeom(k, n, nshards, samp.cpu().numpy(), args.seed_tiles)
        except pkg.KnnError as exc:
            if args.shard == "cells":
                raise
            sys.stderr.write("bench.py: %s\nbench.py: falling back to index-range shards\n" % exc)
            shard_mode = "index"
    def partition_local():
        """2a. LOCAL work only (no collective inside: a rank that fails here has not left its peers blocked in an all-to-all —
        ADVICE r04): where every row belongs; rows and their global numbers sorted by destination (stable: ascending per
        destination).  -> what partition_exchange needs"""
        owner = torch.empty(n_local, dtype=torch.int32, device=dev)
        geom.assign(rows2d.data_ptr(), n_local, owner.data_ptr(), device=local_rank, stream=stream)
        torch.cuda.synchronize()
        counts = torch.bincount(owner, minlength=nshards).to(torch.int64)
        if emu_n:
            return owner, counts, None, None
        order = torch.argsort(owner, stable=True)
        send_rows = rows2d[order].contiguous()
        send_gids = (order + lo).to(torch.int32)
        return owner, counts, send_rows, send_gids

    def partition_exchange(owner, counts, send_rows, send_gids):
        """2b. the rows move to their ranks by one all-to-all (entered only when EVERY rank got through 2a).
        -> (this rank's rows, their global numbers, note)"""
        nonlocal layer_d

        def all_to_all(send, in_split, out_split):
            """rows of `send` (sorted by destination) -> the rows every rank sends here, in source order"""
            out = torch.empty((int(sum(out_split)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
            if rehearse:      # gloo moves CPU tensors
                o_cpu, s_cpu = out.cpu(), send.cpu()
                dist.all_to_all_single(o_cpu, s_cpu, output_split_sizes=out_split, input_split_sizes=in_split)
                return o_cpu.to(send.device)
            dist.all_to_all_single(out, send, output_split_sizes=out_split, input_split_sizes=in_split)
            return out

        if emu_n:
            # one GPU plays every rank in turn: the others' shards exist only long enough to export their seed tiles
            layer_d = torch.zeros(geom.layer_bytes, dtype=torch.uint8, device=dev)
            mine_rows = mine_gids = None
            for r_ in range(emu_n):
                g_ = torch.nonzero(owner == r_).reshape(-1)
                rows_r = rows2d[g_].contiguous()
                gids_r = (g_ + lo).to(torch.int32)
                ix_ = pkg.KnnIndex.sharded(geom, r_, rows_r.data_ptr(), gids_r.data_ptr(), rows_r.shape[0], device=local_rank, stream=stream,
                                           owners=(rows_r, gids_r))
                ix_.seed_export(layer_d.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                ix_.close()
                if r_ == emu_r:
                    mine_rows, mine_gids = rows_r, gids_r
                del rows_r, gids_r, g_
            rows_d, gids_d = mine_rows, mine_gids
        else:
            cnt_out = torch.empty(world, dtype=torch.int64, device=dev)
            if rehearse:
                c_cpu = torch.empty(world, dtype=torch.int64)
                dist.all_to_all_single(c_cpu, counts.cpu())
                cnt_out = c_cpu.to(dev)
            else:
                dist.all_to_all_single(cnt_out, counts)
            in_split, out_split = [int(c_) for c_ in counts.tolist()], [int(c_) for c_ in cnt_out.tolist()]
            rows_d = all_to_all(send_rows, in_split, out_split)
            gids_d = all_to_all(send_gids, in_split, out_split)
        torch.cuda.synchronize()
        note = {"partition": "cell ranges of one global grid of 2^%d cells (%d per rank at most), seed layer %d tile(s) per cell = "
                             "%.0f MB replicated" % (geom.bits, geom.cells_per_rank, geom.seed_tiles, geom.layer_bytes / 1e6),
                "partition_ms": (time.perf_counter() - t_part) * 1e3, "rows_this_rank": int(rows_d.shape[0])}
        if emu_n:
            note["emulated"] = "one GPU as rank %d of %d (no collective in the step)" % (emu_r, emu_n)
        return rows_d, gids_d, note

    if shard_mode == "cells":
        assert not args.separate_init, "a cell-range shard starts its keys itself (KNN_QUERY_INIT_KEYS)"
        # Two phases.  A LOCAL failure — a geometry a rank refuses, an allocation — must not take the whole run
        # down: every rank reports after the local phase, and if any failed all of them keep the index-range shards they
        # already hold.  The collectives are entered only when every rank is ready for them; an error INSIDE a collective is
        # not recoverable (the peers are blocked in it) and ends the run through the process group's timeout.
        local, why = None, ""
        try:
            local = partition_local()
        except Exception as exc:     # noqa: BLE001
            why = "%s: %s" % (type(exc).__name__, exc)
        if os.environ.get("KNN_BENCH_TEST_PARTITION_FAIL_RANK") == str(rank):    # (test hook: tests/test_shards_logic.py)
            local, why = None, "KNN_BENCH_TEST_PARTITION_FAIL_RANK"
        ok = 1 if local is not None else 0
        if dist is not None:
            flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if rehearse else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            rows_d, gids_d, shard_note = partition_exchange(*local)
            del local
            r_d = rows_d.reshape(-1)            # this rank's rows from here on
            n_local = rows_d.shape[0]
            if emu_n:
                args.cpu_queries = 0     # (one rank's answers are not the set's: parity of the shards is tests/test_shards_gpu.py's job)
        else:
            if args.shard == "cells" or emu_n:
                raise SystemExit("bench.py: cell-range shards failed on some rank (%s)" % (why or "another rank"))
            sys.stderr.write("bench.py: cell-range shards failed on some rank (%s): index-range shards instead\n" % (why or "another rank"))
            shard_mode, geom = "index", None
    # Two batches in flight: step i runs on stream i&1 with the index's query workspace i&1 and its
    # own key/result buffers.  The small latency-bound kernels of step i+1 (query fragments, sample
    # pass, thresholds) and — with N > 1 — the all-reduce of step i overlap the other step's scan.
    # Every step is still a complete pass (init, scan, [reduce], unpack) inside the timed region.
    # (the library chains the scans of different slots for shards of >= 16M rows: two in flight are enough then)
    long_scan = n_local >= (1 << 24)
    cells_from = {0: (1 << 19) if k <= 12 else (1 << 20), 1: 1 << 17}.get(pkg.get_option("cells"))   # the library's policy
    cells_expected = (k <= 16 and cells_from is not None and n_local >= cells_from and pkg.get_option("path") in (0, 2)) or shard_mode == "cells"
    inflight = args.inflight if args.inflight > 0 else (2 if long_scan and not cells_expected else 4 if cells_expected and k > 4 else 3)
    nbuf = 1 if args.serial else max(1, min(8, inflight))
    # Key / result buffers: [group][batch]; without a collective only group 0 is used.
    # gsize = batches per exchange step (N > 1).  Round 4: a group was the nbuf batches in flight; with a rank of eight at 26 us of
    # GPU work per batch the host became the limit — one all-reduce through torch.distributed costs the host ~50 us, 12 us per
    # step at four batches per group (one rank, n_local 2^21: 0.048 ms per step against 0.037 without a collective).  Sixteen
    # batches per group: the same 8 KiB-per-batch exchange, one call per 16 steps.  The batches still run nbuf at a time on
    # nbuf streams / workspace slots; a batch's indices are final when its group's reduction has run (inside the timed region:
    # the last group is flushed before the closing fence).
    gsize = nbuf if dist is None or args.serial else max(nbuf, args.group if args.group > 0 else 16)
    keys_all = torch.empty((2, gsize, m), dtype=torch.int64, device=dev)
    outs_all = torch.empty((2, gsize, m), dtype=torch.int32, device=dev)
    keys = [keys_all[0, b] for b in range(nbuf)]
    outs = [outs_all[0, b] for b in range(nbuf)]
    # index build: the first one of a process also loads the code objects and fills the library's buffer pool (13 ms at C3
    # against 4.7 for every later one): built twice, both reported
    def make_index():
        if shard_mode == "cells":
            return pkg.KnnIndex.sharded(geom, emu_r if emu_n else rank, r_d.data_ptr(), gids_d.data_ptr(), n_local,
                                        device=local_rank, stream=stream, owners=(r_d, gids_d))
        return pkg.KnnIndex(k, r_d.data_ptr(), n_local=n_local, device=local_rank, base_index=lo, refs_on_device=True, stream=stream,
                            owners=(r_d,))

    t0 = time.perf_counter()
    index = make_index()
    torch.cuda.synchronize()
    prep_first_ms = (time.perf_counter() - t0) * 1e3
    index.close()
    t0 = time.perf_counter()
    index = make_index()
    torch.cuda.synchronize()
    prep_ms = (time.perf_counter() - t0) * 1e3
    if shard_mode == "cells":
        # the seed layer: every rank's part (the first tiles of each of its cells), gathered; replicated on every rank
        t0 = time.perf_counter()
        if not emu_n:
            layer_d = torch.zeros(geom.layer_bytes, dtype=torch.uint8, device=dev)
            index.seed_export(layer_d.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            part = layer_d[rank * geom.part_bytes:(rank + 1) * geom.part_bytes].clone()
            if rehearse:
                gl = [torch.empty(geom.part_bytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(gl, part.cpu())
                layer_d.copy_(torch.cat(gl).to(dev))
            else:
                dist.all_gather_into_tensor(layer_d, part)
        index.seed_attach(layer_d.data_ptr(), owner=layer_d)
        torch.cuda.synchronize()
        shard_note["seed_layer_ms"] = (time.perf_counter() - t0) * 1e3
    nstreams = nbuf
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    # N > 1: the exchange step.  The nbuf batches in flight form a group whose packed keys are
    # min-reduced over the ranks by ONE all-reduce (nbuf x m keys) on a stream of its own, followed by
    # one unpack; the next group's scans run meanwhile on the batch streams (two groups of buffers).
    # A collective per batch cost ~10 us of stream round trips even on one rank, comparable to the
    # 85 us step of a 2M-row shard.
    reduce_stream = torch.cuda.Stream(device=dev) if dist is not None else None
    group_done = [None, None]      # event: group g's buffers hold final indices and may be reused
    state = {"filled": 0, "group": 0, "last": (0, 0)}

    # (everything the distributed hot loop needs per step is made here once: the host side of a step was ~25 us of torch stream
    # contexts, tensor views and fresh events where a rank of eight's GPU work is 26 us)
    key_ptrs_all = [[keys_all[g_, b_].data_ptr() for b_ in range(gsize)] for g_ in range(2)]
    stream_events = [[torch.cuda.Event() for _ in range(nstreams)] for _ in range(2)]   # "this stream's batches of the group are done"
    group_events = [torch.cuda.Event(), torch.cuda.Event()]
    group_views = [[keys_all[g_, :f_] for f_ in range(gsize + 1)] for g_ in range(2)]
    group_key_ptr = [keys_all[g_].data_ptr() for g_ in range(2)]
    group_out_ptr = [outs_all[g_].data_ptr() for g_ in range(2)]

    def flush_group():
        """All-reduce + unpack the batches enqueued so far in the current group."""
        g, filled = state["group"], state["filled"]
        if dist is None or filled == 0:
            return
        for s_ in range(min(filled, nstreams)):   # every stream that carried batches of the group: one event each, recorded now
            stream_events[g][s_].record(streams[s_])
            reduce_stream.wait_event(stream_events[g][s_])
        with torch.cuda.stream(reduce_stream):      # (torch.distributed launches on torch's current stream)
            # keys < 2^63 (distance bits of a non-negative float): int64 MIN == unsigned MIN
            dist.all_reduce(group_views[g][filled], op=dist.ReduceOp.MIN)
        pkg.keys_to_indices(group_key_ptr[g], filled * m, group_out_ptr[g], device=local_rank, stream=reduce_stream.cuda_stream)
        group_events[g].record(reduce_stream)
        group_done[g] = group_events[g]
        state["group"], state["filled"] = g ^ 1, 0

    raw_streams = [st_.cuda_stream for st_ in streams]
    q_ptr = q_d.data_ptr()
    key_ptrs = [kb_.data_ptr() for kb_ in keys]
    out_ptrs = [ob_.data_ptr() for ob_ in outs]

    def step(i):
        if dist is None and not args.separate_init:
            # the N = 1 hot loop: two library calls with explicit streams and nothing else (a `with torch.cuda.stream(...)`
            # block per step cost the host ~15 us: 0.3 ms of a 20-step timed region whose GPU work is 2.5 ms)
            # (round 4: ONE library call — the int32 indices are written by the batch's last kernel, knn_index_query)
            b = i % nbuf
            index.query_keys(m, q_ptr, key_ptrs[b], stream=raw_streams[b % nstreams], slot=b, init_keys=True,
                             indices_dev=out_ptrs[b])
            state["last"] = (0, b)
            return
        if dist is not None and not args.separate_init:
            # the N > 1 hot loop: wait for the buffers' previous use, one library call, one event — on explicit streams
            b, g = state["filled"], state["group"]
            slot = b % nstreams
            st = streams[slot]
            if b < nstreams and group_done[g] is not None:
                st.wait_event(group_done[g])     # the previous use of these buffers is complete (later batches: stream order)
            index.query_keys(m, q_ptr, key_ptrs_all[g][b], stream=raw_streams[slot], slot=slot, init_keys=True)
            state["last"] = (g, b)
            state["filled"] = b + 1
            if b + 1 == gsize:
                flush_group()
            return
        b = state["filled"] if dist is not None else i % nbuf
        slot = b % nstreams
        st = streams[slot]
        with torch.cuda.stream(st):
            if dist is not None:
                g = state["group"]
                if b < nstreams and group_done[g] is not None:
                    st.wait_event(group_done[g])     # the previous use of these buffers is complete
                kb, ob = keys_all[g, b], outs_all[g, b]
            else:
                g, kb, ob = 0, keys[b], outs[b]
            if args.separate_init:
                pkg.keys_init(kb.data_ptr(), m, device=local_rank, stream=st.cuda_stream)
            index.query_keys(m, q_d.data_ptr(), kb.data_ptr(), stream=st.cuda_stream, slot=slot,
                             init_keys=not args.separate_init)
            if dist is None:
                pkg.keys_to_indices(kb.data_ptr(), m, ob.data_ptr(), device=local_rank, stream=st.cuda_stream)
        state["last"] = (g, b)
        if dist is not None:
            state["filled"] += 1
            if state["filled"] == gsize:
                flush_group()

    def drain():
        flush_group()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, before the W warm-up steps: the first queries of an index allocate its per-slot
    # workspaces (records, partial minima) and load the kernels' code objects; SETUP_STEPS untimed
    # steps get that out of the way even when the caller asks for W = 0 — and the GPU is then kept at
    # this work for SETUP_MS more: a 20-step C3 run started 35 steps (4 ms) after the index build read
    # 0.140 ms per step, started after 100+ steps 0.126 (what 200 timed steps read either way):
    # the chip is not at its working clocks a few milliseconds after a quiet spell.  Every rank does the
    # same number of steps (the exchange step is a collective): rank 0 decides how many.
    setup_steps = SETUP_STEPS
    for i in range(SETUP_STEPS):
        step(i)
    drain()
    fence()
    t_setup = time.perf_counter()
    for i in range(SETUP_STEPS):
        step(i)
    drain()
    fence()
    per_step = max((time.perf_counter() - t_setup) / SETUP_STEPS, 1e-6)
    extra = int(min(4000, max(0.0, SETUP_MS * 1e-3 / per_step)))
    if dist is not None:
        t_extra = torch.tensor([extra], dtype=torch.int64, device=dev)
        dist.broadcast(t_extra, src=0)
        extra = int(t_extra.item())
    extra = (extra + gsize - 1) // gsize * gsize
    for i in range(extra):
        step(i)
    drain()
    fence()
    setup_steps = 2 * SETUP_STEPS + extra
    for i in range(args.warmup):
        step(i)
    drain()
    fence()
    index.timing(max(1, args.time_every))
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_issued = time.perf_counter() - t0      # the host is done enqueueing (launches are asynchronous): host-bound if ~ elapsed
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    launches, kern_ms = index.timing_read()
    # outside the timed region: the dominant kernel alone on the GPU (one batch in flight), so the
    # roofline block can show both the pipelined-phase duration and the kernel's own
    torch.cuda.synchronize()
    alone_n, alone_ms = 0, 0.0
    index.timing(1)
    serial_step_ms = None
    if True:
        # (one launch at a time, but alternating between the slots the timed region used: the library picks the kernel's
        # shape from the caller's last eight calls — with one slot only it would switch to the one-batch-at-a-time shapes
        # and this would no longer be the kernel that ran in the timed region)
        for i_ in range(20):
            b_ = i_ % min(nbuf, 2)
            index.query_keys(m, q_d.data_ptr(), keys[b_].data_ptr(), stream=streams[b_].cuda_stream, slot=b_, init_keys=True)
            torch.cuda.synchronize()
        alone_n, alone_ms = index.timing_read()
        # What a pair of events adds to the launch it brackets: pairs with NOTHING between them on the same idle stream
        # (4.5-4.8 us on MI355X / ROCm 7.2, tools/event_overhead.py).  rocprofv3's kernel durations do not carry it, so
        # it is taken off the bracketed time below; both figures go into the line.
        pair_us = []
        for _ in range(40):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(streams[0]):
                ea.record()
                eb.record()
            torch.cuda.synchronize()
            pair_us.append(ea.elapsed_time(eb) * 1e3)
        event_pair_ms = sorted(pair_us)[len(pair_us) // 2] * 1e-3
        # the whole chain of one batch with nothing else in flight (what a latency-bound caller sees)
        index.timing(False)
        for _ in range(10):     # (untimed: the library's shape policy settles on what a one-batch-at-a-time caller gets)
            index.query_keys(m, q_d.data_ptr(), keys[0].data_ptr(), stream=streams[0].cuda_stream, slot=0, init_keys=True,
                             indices_dev=outs[0].data_ptr())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(50):
            with torch.cuda.stream(streams[0]):
                index.query_keys(m, q_d.data_ptr(), keys[0].data_ptr(), stream=streams[0].cuda_stream, slot=0, init_keys=True,
                                 indices_dev=outs[0].data_ptr())
        torch.cuda.synchronize()
        serial_step_ms = (time.perf_counter() - t1) * 1e3 / 50
    index.timing(False)
    # the exchange step by itself (N > 1): one group's all-reduce + unpack on the reduction stream, nothing else running —
    # what a rank's step carries on top of its scan when the collective does not hide behind the next group
    collective_ms = None
    if dist is not None:
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        with torch.cuda.stream(reduce_stream):
            ev0.record()
            for _ in range(reps):
                dist.all_reduce(keys_all[0, :gsize], op=dist.ReduceOp.MIN)
                pkg.keys_to_indices(keys_all[0].data_ptr(), gsize * m, outs_all[0].data_ptr(), device=local_rank,
                                    stream=reduce_stream.cuda_stream)
            ev1.record()
        torch.cuda.synchronize()
        collective_ms = ev0.elapsed_time(ev1) / reps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    qps = m * args.steps / elapsed
    stats = index.last_stats()
    result_idx = outs_all[state["last"][0], state["last"][1]].cpu().numpy()

    if rank == 0:
        stats_path = int(stats[0])
        roof = roofline_block(k=k, m=m, n_local=n_local, path_taken=stats_path, kern_ms=kern_ms, launches=launches,
                              alone_ms=alone_ms, alone_n=alone_n, event_pair_ms=event_pair_ms, ms_per_step=ms_per_step,
                              serial_step_ms=serial_step_ms,
                              pmc_key=pmc_workload_key(k, m, n_local, shard_mode, emu_n, emu_r, world),
                              source_sha=kernel_source_sha())
        path_taken = stats_path

        cpu = None
        cpu_all = None
        parity = None
        if dist is None and args.cpu_queries != 0:
            cpu = cpu_baseline(k, m, n, args.cpu_queries, result_idx)
            cpu_all = cpu_baseline_all_cores(k, m, n, cpu["value"], result_idx)
        elif dist is not None and args.cpu_queries != 0:
            # no CPU baseline at N > 1, but never report a number for wrong answers: the reduced
            # result of the last step must match the oracle on a few queries of the full set
            parity = parity_spot_check(k, m, n, 8, result_idx)

        line = {
            "metric": "queries/sec (exact 1-NN, bit-exact vs v0), m=%d n=%d k=%d" % (m, n, k),
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # SURVEY 8(d) defines queries/s as m / t for ONE call (reference main.cu:69-73 times a call): `value` is the
            # throughput with batches in flight, this is one batch at a time with nothing else on the GPU
            "value_one_call": (m / (serial_step_ms * 1e-3)) if (serial_step_ms and dist is None) else None,
            "dtype": "f32 results (f16 MFMA filter + f32 exact re-rank)" if path_taken in (2, 4) else "f32", "data": "synthetic",
            "config": {"workload": "%s: k=%d m=%d n=%d uniform[0,1) fp32, refs resident in HBM, sharded over n" %
                                   (wname, k, m, n),
                       "n_per_gpu": n_local, "shards": shard_note if shard_note else ("index ranges (reference core.cu:875-883)" if world > 1 else None),
                       "path": {1: "exact", 2: "mfma_filter+exact_rerank", 3: "grid_index",
                                                     4: "cell_pruned_mfma_filter+exact_rerank"}.get(path_taken),
                       "rerank_candidates": int(stats[1]), "index_prep_ms": prep_ms,
                       "index_prep_first_in_process_ms": prep_first_ms,
                       "batches_in_flight": nstreams, "setup_steps": setup_steps,
                       "host_enqueue_ms_per_step": t_issued * 1e3 / args.steps,
                       "keys_init": "separate launch" if args.separate_init else "inside the query (KNN_QUERY_INIT_KEYS)",
                       "collective": ("%s all_reduce(min) of %d x %d packed keys per %d batches" %
                                      ("gloo (one-GPU rehearsal)" if rehearse else "rccl", gsize, m, gsize))
                       if dist is not None else None,
                       "collective_ms_per_group_alone": collective_ms,
                       "collective_ms_per_step_alone": collective_ms / gsize if collective_ms is not None else None},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if cpu_all is not None:
            line["cpu_baseline_16_threads"] = cpu_all
        if parity:
            line["parity_spot_check"] = parity
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if args.scan_stamps and dist is None:
        import ctypes
        import numpy as np
        fn = pkg.lib().knn_debug_scan_stamps   # (AttributeError: not the stamped build)
        fn.argtypes = [ctypes.c_void_p]
        grabbed = {}
        for tag in ("pipe", "serial"):
            for i in range(48):
                if tag == "pipe":
                    step(i)
                else:
                    index.query_keys(m, q_ptr, key_ptrs[0], stream=raw_streams[0], slot=0, init_keys=True, indices_dev=out_ptrs[0])
            torch.cuda.synchronize()
            buf = np.zeros(8192 * 5, dtype=np.uint64)
            assert fn(buf.ctypes.data) == 0
            grabbed[tag] = buf.reshape(8192, 5)
        np.savez(args.scan_stamps, **grabbed)
    index.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_workload_key(k, m, n_local, shard_mode, emu_n, emu_r, world):
    """What a PMC traffic file must have been taken on to be quoted for this run: the shape the dominant kernel saw (k, m, the
    rows of THIS rank's shard), how the shard was cut, and which rank of how many it is.  (Round 4 quoted the whole-set C3 file
    under --emulate 8:3, where the shard is an eighth: a committed roofline.frac of 2.95.)"""
    key = "k%d_m%d_n%d_%s" % (k, m, n_local, shard_mode if (emu_n or world > 1) else "whole")
    if emu_n:
        key += "_rank%dof%d" % (emu_r, emu_n)
    elif world > 1:
        key += "_world%d" % world
    return key


def find_pmc_profile(pmc_key, source_sha, profiles_dir=None):
    """The newest profiles/*_pmc_traffic.json taken on this workload key -> (name, doc or None, why-not)."""
    import glob
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    stale = None
    for path in sorted(glob.glob(os.path.join(profiles_dir, "*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            continue
        if doc.get("workload_key") != pmc_key:
            continue
        if doc.get("kernel_source_sha256") == source_sha:
            return os.path.basename(path), doc, None
        stale = stale or os.path.basename(path)
    if stale:
        return stale, None, "profiles/%s is from another build of the kernels: not quoted" % stale
    return None, None, "no PMC pass under profiles/ for workload key %s" % pmc_key


READER_CEILING_GBPS = 6500.0    # a bare reader of the scan's access pattern (tools/read_probe2.hip, profiles/r02_cells_probes.txt)


def roofline_block(k, m, n_local, path_taken, kern_ms, launches, alone_ms, alone_n, event_pair_ms, ms_per_step, serial_step_ms,
                   pmc_key, source_sha, profiles_dir=None):
    """Roofline of the dominant kernel (pure: tests/test_host_logic.py calls it on the CPU).  Basis: the
    bytes (or flops) the kernel has to move, divided by the duration of a SINGLE launch with the GPU to itself (HIP events on its
    stream, 20 launches after the timed region) — reproducible from the serial kernel trace under profiles/.  The north_star's
    figure — SURVEY 8(d)'s algorithmic bytes over the step time — is kept under its own name.
    Rules: PMC bytes are quoted only from a file taken on THIS workload key and THIS kernel source; a
    physical fraction above 1, or a byte MODEL that would put the kernel above what a bare reader of its pattern reaches, is not
    printed — `frac` is null and `frac_withheld` says why."""
    kern_avg_ms = kern_ms / max(launches, 1)                  # a launch inside the pipelined timed region
    # (the headline fraction stays on the directly measured bracket; the figure with the event pair's own
    # cost taken off — what rocprofv3 reports as AverageNs — is published under its own keys)
    kern_ms_alone = alone_ms / max(alone_n, 1)                # the same kernel, nothing else on the GPU, between two events
    kern_ms_minus_pair = max(kern_ms_alone - event_pair_ms, 0.5 * kern_ms_alone)
    alg_bytes = 4.0 * k * n_local + 4.0 * k * m + 8.0 * m      # the fp32 rows once, queries, keys
    roof = {}
    if path_taken == 2:
        flops = 2.0 * k * m * n_local
        ach = flops / (kern_ms_alone * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_F16_PEAK_TFLOPS, "traffic": None,
                "kernel": "knn_filter (f16 MFMA 32x32x16 over every pair + exact re-rank)",
                "flops_per_launch": flops}
    elif path_taken == 4:
        # cell-pruned scan: ~4 % of the pairs are scored; the kernel reads the cell-sorted fp16 layout once —
        # 32 B of fragment (k <= 16; 64 B for 16 < k <= 32) + 4 B of norm per position, cells padded to whole tiles (~4 %);
        # 16 < k <= 30: the norm rides in the fragment's free K-slots (KNN_NIF_MAX_K, knn_cells.hip) — no norm array is read
        per_pos = 32.0 * ((k + 15) // 16) + (0.0 if 16 < k <= 30 else 4.0)
        phys = per_pos * n_local * 1.04
        roof = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": None,
                "kernel": "knn_cells_scan (f16 MFMA 32x32x16 over the cells each query could not rule out)",
                "bytes_per_launch": phys, "bytes_source": "layout size (%d B per position, 4 %% padding)" % per_pos}
    else:
        roof = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": None,
                "kernel": "knn_grid_query (uniform-grid ring search, one wave per query: latency-bound, touches a few "
                          "hundred rows per query instead of n)" if path_taken == 3 else
                          "knn_exact (fp32 VALU scan of the rows; HBM-bound only for m <~ 10)",
                "bytes_per_launch": alg_bytes if path_taken != 3 else None,
                "bytes_source": "the fp32 rows once (SURVEY 8d)" if path_taken != 3 else
                                "not a streaming kernel: no byte model (see kernel_ms)"}
        if path_taken == 1:
            lane_ops = (3.0 * k + 3.0) * m * n_local
            roof["valu_frac"] = lane_ops / (kern_ms_alone * 1e-3) / VALU_LANE_OPS_PEAK
    # HBM bytes per launch from PMC counters: collected in separate rocprofv3 --pmc passes of this
    # same command (tools/pmc_traffic.py -> profiles/*_pmc_traffic.json).  Quoted only when the profile was taken on this
    # workload key (shape of the rank's shard, shard mode, rank) AND on the kernel source that is running now.
    roof["pmc_key"] = pmc_key
    pmc_name, pmc_doc, why_not = find_pmc_profile(pmc_key, source_sha, profiles_dir)
    if pmc_doc is not None:
        kname = {2: ("_Z17knn_filter", "void knn_filter_kernel"),
                 4: ("_Z21knn_cells_scan", "void knn_cells_scan_kernel")
                 }.get(path_taken, ("void knn_exact_qreg<16, 2>",))
        best = None   # the variant of the kernel with the most launches in the profile: the timed region's (a rank of eight also
        for name, ent in pmc_doc["kernels"].items():   # runs the self-listing variant, in the one-batch-at-a-time measurement)
            if name.startswith(kname) and ent["hbm_bytes_per_launch"] > 1e6:
                if best is None or (ent.get("launches", 0), ent["hbm_bytes_per_launch"]) > (best.get("launches", 0), best["hbm_bytes_per_launch"]):
                    best = ent
        if best is not None:
            roof["traffic"] = best["hbm_bytes_per_launch"]
            roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; " \
                                     "read bytes = 2 x FETCH_SIZE KiB, gfx950); same kernel source, same workload key" % pmc_name
            if path_taken == 4:
                roof["bytes_per_launch"] = best["hbm_bytes_per_launch"]
                roof["bytes_source"] = "PMC (traffic)"
    else:
        roof["traffic_source"] = why_not
    if path_taken == 4 and m < 512 and roof.get("traffic") is None:
        # a small batch lists only some of the cells: the kernel reads a fraction of the layout and there is no byte model
        # for it short of a PMC pass (m = 1 at C3 reads its ~1600 cells: 2.5 % of the layout)
        roof["bytes_per_launch"] = None
        roof["bytes_source"] = "batch too small to touch every cell: the layout size is not what the kernel reads"
    if roof.get("bound") == "hbm":
        bpl = roof.get("bytes_per_launch")
        roof["achieved"] = bpl / (kern_ms_alone * 1e-3) / 1e9 if bpl else None
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBPS if bpl else None
    roof["kernel_ms"] = kern_ms_alone
    roof["kernel_ms_basis"] = ("HIP events around the kernel on its stream, %d single launches, nothing else on the GPU "
                               "(the bracket as measured; an EMPTY event pair on the same stream reads event_pair_ms)" % alone_n)
    roof["event_pair_ms"] = event_pair_ms
    roof["kernel_ms_minus_event_pair"] = kern_ms_minus_pair   # ~ rocprofv3's AverageNs of the same launches
    if roof.get("bound") == "hbm" and roof.get("bytes_per_launch"):
        roof["frac_minus_event_pair"] = roof["bytes_per_launch"] / (kern_ms_minus_pair * 1e-3) / 1e9 / HBM_PEAK_GBPS
    # ---- nothing above what the wires can carry leaves this function
    if roof.get("bound") == "hbm" and roof.get("frac") is not None:
        modelled = roof.get("bytes_source", "").startswith("layout size")
        limit = READER_CEILING_GBPS / HBM_PEAK_GBPS if modelled else 1.0
        worst = max(roof["frac"], roof.get("frac_minus_event_pair") or 0.0)
        if worst > limit:
            roof["frac_withheld"] = (
                "%.3f of 8 TB/s from %s is above %s: these are not the bytes this launch moved (a shard whose batch lists only "
                "some of its cells reads less than its layout) — take a PMC pass for workload key %s (tools/pmc_traffic.py)"
                % (worst, roof.get("bytes_source"), "what a bare reader of the pattern reaches (6.5 TB/s)" if modelled else "the peak", pmc_key))
            roof["achieved"] = None
            roof["frac"] = None
            roof.pop("frac_minus_event_pair", None)
    elif roof.get("frac") is not None and roof["frac"] > 1.0:
        roof["frac_withheld"] = "%.3f of the peak: not a physical figure" % roof["frac"]
        roof["achieved"] = None
        roof["frac"] = None
    roof["kernel_in_pipeline_ms"] = kern_avg_ms           # a launch that shares the GPU with the other batches in flight
    roof["kernel_launches_timed_in_pipeline"] = launches
    roof["algorithmic_bytes_per_launch"] = alg_bytes
    roof["algorithmic_frac_per_step"] = alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS
    roof["algorithmic_frac_note"] = "SURVEY 8(d) bytes (the fp32 rows once) / ms_per_step / 8 TB/s: the north_star's figure; " \
                                    "can exceed what the wires carry because the scan reads the fp16 layout (36 B per row, not 64)"
    roof["serial_step_ms"] = serial_step_ms
    if path_taken == 2 and k <= 16:
        pairs_per_simd = (n_local / 32.0) * ((m + 31) // 32) / (256 * 4)
        roof["ceiling"] = {
            "what": "vector-issue floor of 'score every pair' at k <= 16: 1 MFMA + 8 v_min3_f32 per 32x32 tile pair",
            "cycles_per_tile_pair": 50.0, "tile_pairs_per_simd": pairs_per_simd,
            "ms_at_2.4GHz": pairs_per_simd * 50.0 / 2.4e9 * 1e3,
            "ms_at_measured_clock_1.8GHz": pairs_per_simd * 50.0 / 1.8e9 * 1e3,
            "source": "profiles/r02_filter_probe.txt (variants 21-39), profiles/r02_cvt_probe.txt"}
    if path_taken == 4:
        roof["ceiling"] = {
            "what": "HBM read rate a bare reader of the same access pattern reaches on this chip",
            "GBps": READER_CEILING_GBPS,
            "ms_for_this_launch": (roof["bytes_per_launch"] / (READER_CEILING_GBPS * 1e9) * 1e3) if roof.get("bytes_per_launch") and not roof.get("frac_withheld") else None,
            "source": "tools/read_probe2.hip, tools/read_probe.hip (profiles/r02_cells_probes.txt), DESIGN 4.5"}
    return roof


def kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    for name in ("knn_filter.hip", "knn_cells.hip", "knn_filter_dev.h", "knn_exact.hip", "knn_exact_dev.h", "knn_common.h"):
        with open(os.path.join(ROOT, "multicore_hw2_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def parity_spot_check(k, m, n, nq, gpu_idx):
    """Oracle (OpenMP over queries) on the first nq queries against the whole reference set."""
    import subprocess
    from tests.oracle_lib import Oracle
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    o = Oracle(os.path.join(ROOT, "oracle", "libknn_oracle.so"))
    Q = o.synth(nq * k, 1000)
    R = o.synth(n * k, 1001)
    want = o.v0(k, Q, R, threads=16)
    if not (want == gpu_idx[:nq]).all():
        raise SystemExit("PARITY FAILURE: reduced multi-GPU indices differ from the CPU oracle")
    return "%d/%d sampled queries identical to the CPU oracle over all %d refs" % (nq, nq, n)


def cpu_baseline(k, m, n, cpu_queries, gpu_idx):
    """Serial CPU oracle (restates the reference's v0, core.cu:27-62) on a bounded sample of the
    same workload: the first q queries against all n references, 1 thread.  Also asserts the GPU
    answer for those queries is identical."""
    import numpy as np
    import subprocess
    from tests.oracle_lib import Oracle
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    o = Oracle(os.path.join(ROOT, "oracle", "libknn_oracle.so"))
    Q = o.synth(m * k, 1000)
    R = o.synth(n * k, 1001)
    if cpu_queries < 0:          # size the sample for ~15 s from a 2-query probe
        t0 = time.perf_counter()
        o.v0_serial(k, Q[:2 * k], R)
        per_query = (time.perf_counter() - t0) / 2
        cpu_queries = int(max(1, min(m, 15.0 / max(per_query, 1e-9))))
    t0 = time.perf_counter()
    want = o.v0_serial(k, Q[:cpu_queries * k], R)
    dt = time.perf_counter() - t0
    if not (want == gpu_idx[:cpu_queries]).all():
        raise SystemExit("PARITY FAILURE: GPU indices differ from the CPU oracle on the baseline sample")
    return {"value": cpu_queries / dt, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "first %d of %d queries against all %d refs, serial oracle (v0 restatement), %.1f s; "
                      "GPU indices identical on the sample" % (cpu_queries, m, n, dt),
            "host_cpus": os.cpu_count()}


def cpu_baseline_all_cores(k, m, n, serial_qps, gpu_idx):
    """'v0 x P cores' (BASELINE.md §3): the same oracle with OpenMP over queries (the per-query
    loop is untouched, so results are bit-identical) on all host cores, on a sample sized for ~10 s from
    the serial rate; also a parity check of that many more queries."""
    from tests.oracle_lib import Oracle
    o = Oracle(os.path.join(ROOT, "oracle", "libknn_oracle.so"))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))      # a one-GPU box's CPU share is 16 threads, whatever the host has
    q = int(max(cores, min(m, 10.0 * serial_qps * cores)))
    q = min(m, q)
    Q = o.synth(m * k, 1000)
    R = o.synth(n * k, 1001)
    t0 = time.perf_counter()
    want = o.v0(k, Q[:q * k], R, threads=cores)
    dt = time.perf_counter() - t0
    if not (want == gpu_idx[:q]).all():
        raise SystemExit("PARITY FAILURE: GPU indices differ from the CPU oracle on the all-cores sample")
    return {"value": q / dt, "unit": "queries/s", "cores": cores, "kind": "port", "host_cpus": os.cpu_count(),
            "note": "threads = min(affinity, 16): a one-GPU box's CPU share, not every core of the host",
            "sample": "first %d of %d queries against all %d refs, oracle with OpenMP over queries on %d threads, %.1f s; "
                      "GPU indices identical on the sample" % (q, m, n, cores, dt)}


if __name__ == "__main__":
    main()
