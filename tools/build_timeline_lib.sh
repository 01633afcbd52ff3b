#!/bin/bash
# private build of the library with the scan's wall-clock stamps compiled in (tools/scan_timeline.py); not the product
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/multicore_hw2_amd/csrc; T=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DKNN_SCAN_TIMELINE"
for f in knn_exact knn_filter knn_cells knn_grid; do /opt/rocm/bin/hipcc $F -c $C/$f.hip -o $T/$f.o & done; wait
for f in knn_api knn_rccl ta_compat; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -I$C -c $C/$f.cpp -o $T/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/libknn_timeline.so $T/*.o -lpthread -ldl
rm -rf $T; echo built tools/libknn_timeline.so
