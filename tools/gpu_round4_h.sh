set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_h; mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('$O/$tag.json')); print('%-28s step %.4f kernel %.4f serial %.4f rerank %d' % ('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['serial_step_ms'], d['config']['rerank_candidates']))"; }
for N in 8 4; do
run emu${N}_sd4_t2 --emulate $N:0
run emu${N}_sd4_t1 --emulate $N:0 --seed-tiles 1
KNN_MI355X_SEED_DIMS=3 run emu${N}_sd3_t2 --emulate $N:0
KNN_MI355X_SEED_DIMS=3 run emu${N}_sd3_t4 --emulate $N:0 --seed-tiles 4
KNN_MI355X_SEED_DIMS=2 run emu${N}_sd2_t2 --emulate $N:0
KNN_MI355X_SEED_DIMS=2 run emu${N}_sd2_t4 --emulate $N:0 --seed-tiles 4
done
run c4_emu8 --workload c4 --emulate 8:0
