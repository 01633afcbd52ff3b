#!/bin/bash
# Round-end evidence run on the GPU box (everything lands under gpurun_out/r04_final/ — a fresh directory: gpurun MERGES into
# gpurun_out/, and an earlier collection's files would mix with this one's; tools/publish_profiles.py copies into profiles/
# afterwards in the build container).  Two calls (each within gpurun's 20 minutes):
#   gpurun -- 'bash tools/collect_profiles.sh a'    bench lines, kernel traces, PMC passes
#   gpurun -- 'bash tools/collect_profiles.sh b'    emulated ranks, deep K, distributions, build / ingest / drop-in records
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
B() { out=$1; shift; timeout -k 10 300 python3 $R/bench.py "$@" > $O/${out}_bench.json 2>> $O/bench.err || { echo "bench $out failed"; tail -5 $O/bench.err; exit 1; }; }
KT() { out=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$out -- python3 $R/bench.py --cpu-queries 0 "$@" > $O/kt_$out.json 2> $O/kt_$out.err || { echo "trace $out failed"; tail -5 $O/kt_$out.err; exit 1; }; }
PMC() { out=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$out -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 $PMC_ARGS > /dev/null 2> $O/pmc_$out.err || { echo "pmc $out failed"; tail -5 $O/pmc_$out.err; exit 1; }; }
if [ "$1" = a ]; then
step "bench lines"
B c3
B c3_fullscan --cells 2 --cpu-queries 0
B c2 --workload c2
B c5 --workload c5 --cpu-queries 2048
B c4_1gpu --workload c4 --cpu-queries 0
B 16_1_16777216 --workload 16,1,16777216 --cpu-queries 0
B 16_64_16777216 --workload 16,64,16777216 --cpu-queries 0
step "index-range shard sizes of N = 2, 4, 8 (what --shard index gives a rank)"
for n in 8388608 4194304 2097152; do B 16_1024_$n --workload 16,1024,$n --cpu-queries 0; done
step "the bench's distributed code path on one rank (RCCL all-reduce of a group of batches), per-rank shard of N = 8"
KNN_BENCH_FORCE_DIST=1 B 16_1024_2097152_dist1 --workload 16,1024,2097152 --cpu-queries 0
step "one batch at a time"
for n in 16777216 4194304 2097152; do B 16_1024_${n}_serial --workload 16,1024,$n --cpu-queries 0 --serial; done
step "kernel traces: pipelined (the default bench) and --serial (one batch at a time: single-launch durations)"
KT c3
KT c2 --workload c2
KT c5 --workload c5
KT c3_serial --serial
KT 2097152_serial --workload 16,1024,2097152 --serial
step "pmc passes (C3)"
PMC_ARGS=""
PMC fetch FETCH_SIZE
PMC write WRITE_SIZE
PMC l2 TCC_HIT_sum TCC_MISS_sum
PMC_ARGS="--serial"
PMC sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES
PMC sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
echo done a
fi
if [ "$1" = b ]; then
step "emulated ranks of a cell-range sharded C3 (one GPU plays rank r of N; no collective in the step)"
for e in 2:0 2:1 4:0 4:1 4:3 8:0 8:3 8:7; do B c3_rank_${e#*:}_of_${e%:*} --emulate $e --cpu-queries 0; done
for e in 8:0 8:3; do B c3_rank_${e#*:}_of_${e%:*}_serial --emulate $e --cpu-queries 0 --serial; done
B c3_rank_0_of_8_index_shards --workload 16,1024,2097152 --cpu-queries 0
step "a rank of 8 of C4 (n = 2^27: the rank's 2^24 rows on the global 2^19-cell grid)"
B c4_rank_0_of_8 --workload c4 --emulate 8:0 --cpu-queries 0
KT c3_rank_0_of_8 --emulate 8:0
KT c3_rank_0_of_8_serial --emulate 8:0 --serial
step "128 < k <= 4096 on the MFMA filter"
B 64_65536_65536 --workload 64,65536,65536 --cpu-queries 0 --steps 20 --warmup 3
B 256_65536_65536 --workload 256,65536,65536 --cpu-queries 0 --steps 20 --warmup 3
B 512_65536_65536 --workload 512,65536,65536 --cpu-queries 0 --steps 20 --warmup 3
B 640_16384_65536 --workload 640,16384,65536 --cpu-queries 0 --steps 20 --warmup 3
B 1024_65536_65536 --workload 1024,65536,65536 --cpu-queries 0 --steps 10 --warmup 2
B 2048_8192_32768 --workload 2048,8192,32768 --cpu-queries 0 --steps 20 --warmup 3
KT k1024 --workload 1024,16384,65536 --steps 10 --warmup 2
step "SQ counters of the deep-K scans"
bash $R/tools/pmc_sq_deepk.sh gpurun_out/r04_final c5 c5 > $O/deepk_sq_counters.txt 2>&1 || exit 1
bash $R/tools/pmc_sq_deepk.sh gpurun_out/r04_final 1024,16384,65536 k1024 >> $O/deepk_sq_counters.txt 2>&1 || exit 1
cd /tmp
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
step "per-wave timeline of the scan (stamped build of the library)"
KNN_MI355X_LIB=$R/tools/libknn_timeline.so timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --cpu-queries 0 --scan-stamps $O/stamps_c3.npz > /dev/null 2>> $O/bench.err || exit 1
KNN_MI355X_LIB=$R/tools/libknn_timeline.so timeout -k 10 200 python3 bench.py --emulate 8:0 --steps 100 --warmup 20 --cpu-queries 0 --scan-stamps $O/stamps_rank_0_of_8.npz > /dev/null 2>> $O/bench.err || exit 1
{ echo "== C3"; python3 tools/scan_timeline.py $O/stamps_c3.npz; echo "== rank 0 of 8 (cell-range shard of C3)"; python3 tools/scan_timeline.py $O/stamps_rank_0_of_8.npz; } > $O/scan_timeline_final.txt
echo done b
fi
