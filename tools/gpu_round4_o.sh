cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > /tmp/$tag.json 2> /tmp/$tag.err || { echo "$tag failed"; tail -3 /tmp/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('/tmp/$tag.json')); print('%-28s step %.4f kernel %.4f serial %.4f' % ('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['serial_step_ms']))"; }
for rep in 1 2; do
for N in 4 2; do
run emu${N}_auto --emulate $N:0
run emu${N}_b2_d1 --emulate $N:0 --opt scan_blocks=2 --opt scan_deal=1
run emu${N}_b2_d2 --emulate $N:0 --opt scan_blocks=2 --opt scan_deal=2
run emu${N}_b1_d2 --emulate $N:0 --opt scan_blocks=1 --opt scan_deal=2
done
done
