set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_dk; mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-queries 0 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return 1; }
  python -c "
import json; d=json.load(open('$O/$tag.json')); r=d['roofline']; print('%-20s step %.4f kernel %.4f frac %.3f serial %.4f' % ('$tag', d['ms_per_step'], r['kernel_ms'], r['frac'], r['serial_step_ms']))"; }
for mr in 8 16 32 64 8 16 32 64; do
  run mr${mr}_k1024big --workload 1024,65536,65536 --steps 10 --warmup 2 --opt chunk_blocks=$((mr*256+4))
  run mr${mr}_k2048big --workload 2048,32768,65536 --steps 10 --warmup 2 --opt chunk_blocks=$((mr*256+4))
done
