set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_f; mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('$O/$tag.json')); print('%-28s step %.4f kernel %.4f serial %.4f inflight %d' % ('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['serial_step_ms'], d['config']['batches_in_flight']))"; }
for B in 3 4 6 8; do run emu8_B$B --emulate 8:0 --inflight $B; done
run emu8_B4_blocks2 --emulate 8:0 --inflight 4 --opt scan_blocks=2
run emu8_B8_blocks2 --emulate 8:0 --inflight 8 --opt scan_blocks=2
run emu8_B6_deal2 --emulate 8:0 --inflight 6 --opt scan_deal=2
for B in 4 6 8; do run emu4_B$B --emulate 4:0 --inflight $B; done
for B in 4 6; do run c3_B$B --inflight $B; done
