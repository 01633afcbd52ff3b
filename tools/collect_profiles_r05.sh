#!/bin/bash
# Round-5 evidence run on the GPU box (everything lands under gpurun_out/r05_final/ — a fresh directory per part: gpurun MERGES
# into gpurun_out/; tools/publish_profiles_r05.py copies into profiles/ afterwards in the build container).  Three calls:
#   gpurun -- 'bash tools/collect_profiles_r05.sh a'    bench lines, kernel traces, PMC passes keyed by workload (C3, rank 0 of 8, C4)
#   gpurun -- 'bash tools/collect_profiles_r05.sh b'    emulated ranks, pipeline gaps, list makers, deep K, k 17-20
#   gpurun -- 'bash tools/collect_profiles_r05.sh c'    distributions, build / ingest / drop-in records, per-wave timeline
#   gpurun -- 'bash tools/collect_profiles_r05.sh d'    (after tools/publish_profiles_r05.py has made the PMC files of part a) the bench
#                                                       lines of the three workloads with a PMC file again, so that their roofline
#                                                       blocks quote the counter bytes — what a later `python bench.py` prints
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
B() { out=$1; shift; timeout -k 10 300 python3 $R/bench.py "$@" > $O/${out}_bench.json 2>> $O/bench.err || { echo "bench $out failed"; tail -5 $O/bench.err; exit 1; }; }
KT() { out=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$out -- python3 $R/bench.py --cpu-queries 0 "$@" > $O/kt_$out.json 2> $O/kt_$out.err || { echo "trace $out failed"; tail -5 $O/kt_$out.err; exit 1; }; }
PMCSET() { # tag, bench args: FETCH / WRITE / L2 passes -> profiles/r05_<tag>_pmc_traffic.json (keyed by workload)
  tag=$1; shift
  for c in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
    set -- "$@"; name=${c%% *}; ctrs=${c#* }
    timeout -k 10 400 rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc_${tag}_$name -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 "$@" > $O/pmc_${tag}_$name.json 2> $O/pmc_${tag}_$name.err || { echo "pmc $tag $name failed"; tail -5 $O/pmc_${tag}_$name.err; exit 1; }
  done
}
SQ() { out=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$out -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-queries 0 --serial > /dev/null 2> $O/pmc_$out.err || { echo "pmc $out failed"; tail -5 $O/pmc_$out.err; exit 1; }; }
if [ "$1" = a ]; then
step "bench lines"
B c3
B c3_fullscan --cells 2 --cpu-queries 0
B c2 --workload c2
B c5 --workload c5 --cpu-queries 2048
B c5_fixed_thresholds --workload c5 --cpu-queries 0 --opt run_thresholds=2
B c4_1gpu --workload c4 --cpu-queries 0
B 16_1_16777216 --workload 16,1,16777216 --cpu-queries 0
B 16_64_16777216 --workload 16,64,16777216 --cpu-queries 0
step "index-range shard sizes of N = 2, 4, 8"
for n in 8388608 4194304 2097152; do B 16_1024_$n --workload 16,1024,$n --cpu-queries 0; done
step "the bench's distributed code path on one rank"
KNN_BENCH_FORCE_DIST=1 B 16_1024_2097152_dist1 --workload 16,1024,2097152 --cpu-queries 0
step "one batch at a time"
for n in 16777216 4194304 2097152; do B 16_1024_${n}_serial --workload 16,1024,$n --cpu-queries 0 --serial; done
step "kernel traces: pipelined and --serial"
KT c3
KT c2 --workload c2
KT c5 --workload c5
KT c3_serial --serial
KT 2097152_serial --workload 16,1024,2097152 --serial
step "pmc passes keyed by workload: C3, rank 0 of 8 of C3, C4 on one GPU"
PMCSET c3
PMCSET c3_rank_0_of_8 --emulate 8:0
PMCSET c4_1gpu --workload c4
PMCSET k20 --workload 20,1024,16777216
step "SQ counters (C3, one batch at a time)"
SQ sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES
SQ sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
echo done a
fi
if [ "$1" = b ]; then
step "emulated ranks of a cell-range sharded C3"
for e in 2:0 2:1 4:0 4:1 4:3 8:0 8:3 8:7; do B c3_rank_${e#*:}_of_${e%:*} --emulate $e --cpu-queries 0; done
for e in 8:0 8:3 4:0; do B c3_rank_${e#*:}_of_${e%:*}_serial --emulate $e --cpu-queries 0 --serial; done
step "who lists the cells' queries (rank 0 of 8, one batch at a time and pipelined)"
B c3_rank_0_of_8_serial_match --emulate 8:0 --cpu-queries 0 --serial --opt cells_lists=1
B c3_rank_0_of_8_serial_self --emulate 8:0 --cpu-queries 0 --serial --opt cells_lists=2
B c3_rank_0_of_8_self --emulate 8:0 --cpu-queries 0 --opt cells_lists=2
B c3_self --cpu-queries 0 --opt cells_lists=2
step "a rank of 8 of C4"
B c4_rank_0_of_8 --workload c4 --emulate 8:0 --cpu-queries 0
KT c3_rank_0_of_8 --emulate 8:0
KT c3_rank_0_of_8_serial --emulate 8:0 --serial
step "kernel timeline of the pipelined rank of 8 (gaps, concurrency)"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/ktt_rank_0_of_8 -- python3 $R/bench.py --emulate 8:0 --cpu-queries 0 --steps 200 > /dev/null 2> $O/ktt.err || exit 1
( cd $R && python3 tools/pipeline_gaps.py $O/ktt_rank_0_of_8 1200 > $O/pipeline_gaps_rank_0_of_8.txt ) || exit 1
step "batches in flight (rank 0 of 8)"
for f in 3 4 6 8; do B c3_rank_0_of_8_inflight_$f --emulate 8:0 --cpu-queries 0 --inflight $f; done
step "deep K on the MFMA filter"
B 64_65536_65536 --workload 64,65536,65536 --cpu-queries 0 --steps 20 --warmup 3
B 100_65536_65536 --workload 100,65536,65536 --cpu-queries 0 --steps 20 --warmup 3
B 256_65536_65536 --workload 256,65536,65536 --cpu-queries 0 --steps 20 --warmup 3
B 512_65536_65536 --workload 512,65536,65536 --cpu-queries 0 --steps 20 --warmup 3
B 640_16384_65536 --workload 640,16384,65536 --cpu-queries 0 --steps 20 --warmup 3
B 1024_65536_65536 --workload 1024,65536,65536 --cpu-queries 0 --steps 10 --warmup 2
B 2048_8192_32768 --workload 2048,8192,32768 --cpu-queries 0 --steps 20 --warmup 3
step "16 < k <= 32 on the pruned scan against the full scan"
for k in 17 18 20 22 24; do B ${k}_1024_16777216 --workload $k,1024,16777216 --cpu-queries 0 --steps 60; B ${k}_1024_16777216_fullscan --workload $k,1024,16777216 --cpu-queries 0 --steps 60 --cells 2; done
B 26_1024_16777216_pruned --workload 26,1024,16777216 --cpu-queries 0 --steps 60 --cells 1
B 17_1024_4194304 --workload 17,1024,4194304 --cpu-queries 64 --steps 60
B 20_1024_4194304 --workload 20,1024,4194304 --cpu-queries 64 --steps 60
B 22_1024_8388608 --workload 22,1024,8388608 --cpu-queries 0 --steps 60
KT 20_1024_16777216_serial --workload 20,1024,16777216 --serial --steps 40
step "SQ counters of the deep-K scans"
bash $R/tools/pmc_sq_deepk.sh gpurun_out/r05_final c5 c5 > $O/deepk_sq_counters.txt 2>&1 || exit 1
echo done b
fi
if [ "$1" = c ]; then
cd $R
step "off the uniform cube"
( timeout -k 10 300 python3 tools/distribution_check.py 2>&1 | grep -v amdgpu.ids > $O/distribution_check.txt ) || exit 1
( KNN_DC_OPTS=cells_centre=2 timeout -k 10 300 python3 tools/distribution_check.py 2>&1 | grep -v amdgpu.ids > $O/distribution_check_one_frame.txt ) || exit 1
( KNN_DC_OPTS=cells_centre=1 timeout -k 10 300 python3 tools/distribution_check.py 2>&1 | grep -v amdgpu.ids > $O/distribution_check_cell_frames.txt ) || exit 1
for c in clusters64 heavy_tail; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -- python3 $R/tools/distribution_check.py $c > /dev/null 2>&1 || exit 1
done
step "index build stages + kernels"
KNN_MI355X_TRACE_BUILD=1 timeout -k 10 120 python3 tools/build_trace.py > $O/build_trace.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_build -- python3 tools/build_trace.py > /dev/null 2>&1 || exit 1
step "list lengths"
for n in 2097152 4194304 16777216; do timeout -k 10 60 python3 tools/cells_trace.py $n 2>&1 | grep "knn cells" >> $O/cells_trace.txt; done
step "drop-in and ingest timing"
timeout -k 10 300 python3 tools/dropin_timing.py > $O/dropin_timing.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/ingest_timing.py > $O/ingest_timing.txt 2>&1 || exit 1
step "per-wave timeline of the scan (stamped build of the library)"
KNN_MI355X_LIB=$R/tools/libknn_timeline.so timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --cpu-queries 0 --scan-stamps $O/stamps_c3.npz > /dev/null 2>> $O/bench.err || exit 1
KNN_MI355X_LIB=$R/tools/libknn_timeline.so timeout -k 10 200 python3 bench.py --emulate 8:0 --steps 100 --warmup 20 --cpu-queries 0 --scan-stamps $O/stamps_rank_0_of_8.npz > /dev/null 2>> $O/bench.err || exit 1
{ echo "== C3"; python3 tools/scan_timeline.py $O/stamps_c3.npz; echo "== rank 0 of 8 (cell-range shard of C3)"; python3 tools/scan_timeline.py $O/stamps_rank_0_of_8.npz; } > $O/scan_timeline.txt
step "fuzz (bounded)"
{ echo "== drop-in, random shapes / paths / shards"; timeout -k 10 200 python3 tools/fuzz_parity.py 500 50505 2>&1 | tail -3;
  echo "== cell-pruned scan (k <= 32, both list makers, three builds)"; FUZZ_CELLS=1 timeout -k 10 420 python3 tools/fuzz_parity.py 400 27182 2>&1 | tail -3; } > $O/fuzz.txt 2>&1 || true
echo done c
fi
if [ "$1" = k20 ]; then   # (only the k = 20 PMC passes: added after the round's last full collection)
step "pmc passes: k 20"
PMCSET k20 --workload 20,1024,16777216
echo done k20
fi
if [ "$1" = d ]; then
step "bench lines that quote the PMC files of this collection"
B c3
B c4_1gpu --workload c4 --cpu-queries 0
B c3_rank_0_of_8 --emulate 8:0 --cpu-queries 0
B 20_1024_16777216 --workload 20,1024,16777216 --cpu-queries 0 --steps 60
echo done d
fi
