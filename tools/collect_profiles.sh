#!/bin/bash
# Round-end evidence run on the GPU box (everything lands under gpurun_out/r03_final_k/ — a fresh directory: gpurun MERGES into
# gpurun_out/, and an earlier collection's files would mix with this one's; tools/pmc_traffic.py and the
# copy into profiles/ happen afterwards in the build container).  usage: gpurun -- 'bash tools/collect_profiles.sh'
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final_k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
step "bench lines"
timeout -k 10 300 python3 $R/bench.py > $O/c3_bench.json 2> $O/c3_bench.err || exit 1
timeout -k 10 300 python3 $R/bench.py --cells 2 --cpu-queries 0 > $O/c3_fullscan_bench.json 2>> $O/c3_bench.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload c2 > $O/c2_bench.json 2>> $O/c3_bench.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload c5 --cpu-queries 2048 > $O/c5_bench.json 2>> $O/c3_bench.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload c5 --cpu-queries 0 --deepk 3 > $O/c5_one_tile_per_barrier_bench.json 2>> $O/c3_bench.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload c4 --cpu-queries 0 > $O/c4_1gpu_bench.json 2>> $O/c3_bench.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload 16,1,16777216 --cpu-queries 0 > $O/16_1_16777216_bench.json 2>> $O/c3_bench.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload 16,64,16777216 --cpu-queries 0 > $O/16_64_16777216_bench.json 2>> $O/c3_bench.err || exit 1
step "small-shard rows (the per-rank shards of N = 2, 4, 8), default chain and the round-2 chain on the same box"
for n in 8388608 4194304 2097152; do
  timeout -k 10 200 python3 $R/bench.py --workload 16,1024,$n --cpu-queries 0 > $O/16_1024_${n}_bench.json 2>> $O/c3_bench.err || exit 1
  timeout -k 10 200 python3 $R/bench.py --workload 16,1024,$n --cpu-queries 0 --cells-variant 1 --separate-init > $O/16_1024_${n}_r02chain_bench.json 2>> $O/c3_bench.err || exit 1
done
step "the bench's distributed code path on one rank (RCCL all-reduce of a group of batches), per-rank shard of N = 8"
KNN_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 $R/bench.py --workload 16,1024,2097152 --cpu-queries 0 > $O/16_1024_2097152_dist1_bench.json 2>> $O/c3_bench.err || exit 1
step "kernel traces: pipelined (the default bench) and --serial (one batch at a time: single-launch durations)"
for w in c3 c2 c5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 $R/bench.py --workload $w --cpu-queries 0 > /dev/null 2>&1 || exit 1
done
# (--serial from a fresh process would take the single-slot shapes — items from a block counter, two blocks per CU; the traces the
# roofline block is checked against must show the kernel of the TIMED region: the fixed deal, one block per CU on small shards)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3_serial -- python3 $R/bench.py --cpu-queries 0 --serial --opt scan_deal=1 --opt scan_blocks=2 > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3_serial_block_counter -- python3 $R/bench.py --cpu-queries 0 --serial > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2097152_serial -- python3 $R/bench.py --workload 16,1024,2097152 --cpu-queries 0 --serial --opt scan_deal=1 --opt scan_blocks=1 > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2097152_r02chain_serial -- python3 $R/bench.py --workload 16,1024,2097152 --cpu-queries 0 --serial --cells-variant 1 --separate-init --opt scan_deal=1 --opt scan_blocks=1 > /dev/null 2>&1 || exit 1
step "pmc passes (C3)"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 --serial > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 --serial > /dev/null 2>&1 || exit 1
step "one batch at a time (single-slot callers: block-counter deal, two blocks per CU)"
for n in 16777216 4194304 2097152; do
  timeout -k 10 200 python3 $R/bench.py --workload 16,1024,$n --cpu-queries 0 --serial > $O/16_1024_${n}_serial_bench.json 2>> $O/c3_bench.err || exit 1
done
step "128 < k <= 4096 on the MFMA filter"
timeout -k 10 200 python3 $R/bench.py --workload 256,65536,65536 --cpu-queries 0 --steps 20 --warmup 3 > $O/256_65536_65536_bench.json 2>> $O/c3_bench.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload 512,65536,65536 --cpu-queries 0 --steps 20 --warmup 3 > $O/512_65536_65536_bench.json 2>> $O/c3_bench.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload 1024,65536,65536 --cpu-queries 0 --steps 10 --warmup 2 > $O/1024_65536_65536_bench.json 2>> $O/c3_bench.err || exit 1
step "off the uniform cube"
( cd $R && timeout -k 10 300 python3 tools/distribution_check.py 2>&1 | grep -v amdgpu.ids > $O/distribution_check.txt ) || exit 1
for c in clusters64 heavy_tail; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -- python3 $R/tools/distribution_check.py $c > /dev/null 2>&1 || exit 1
done
step "index build stages"
cd $R
KNN_MI355X_TRACE_BUILD=1 timeout -k 10 120 python3 tools/build_trace.py > $O/build_trace.txt 2>&1 || exit 1
step "list lengths"
for n in 2097152 4194304 16777216; do timeout -k 10 60 python3 tools/cells_trace.py $n 2>&1 | grep "knn cells" >> $O/cells_trace.txt; done
step "drop-in timing"
timeout -k 10 300 python3 tools/dropin_timing.py > $O/dropin_timing.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/ingest_timing.py > $O/ingest_timing.txt 2>&1 || exit 1
echo done
