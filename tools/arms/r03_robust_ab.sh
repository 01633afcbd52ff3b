set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_robust
timeout -k 10 600 python -m pytest tests/test_cells_gpu.py -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/distribution_check.py clusters64 heavy_tail 2>&1 | grep -v amdgpu.ids
SIZES="2097152 16777216" bash tools/ab_prev.sh
