cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > /tmp/$tag.json 2> /tmp/$tag.err || { echo "$tag failed"; tail -3 /tmp/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('/tmp/$tag.json')); print('%-28s step %.4f kernel %.4f serial %.4f' % ('$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['serial_step_ms']))"; }
for rep in 1 2 3; do
run c3_auto
run c3_deal2 --opt scan_deal=2
run c3_deal2_b1 --opt scan_deal=2 --opt scan_blocks=1
run emu8_auto --emulate 8:0
run emu8_deal2 --emulate 8:0 --opt scan_deal=2
run emu8_deal2_b2 --emulate 8:0 --opt scan_deal=2 --opt scan_blocks=2
done
