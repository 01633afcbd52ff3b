// cells_arms.hip — the kernels of the cell-pruned path that LOST their A/B in rounds 2-3 and left the product library in
// round 4.  Kept as a record that still compiles against the production sources (the pattern of
// tools/filter_probe.hip); nothing here is linked into libknn_mi355x.so.  Results: profiles/r03_sweep_experiments.txt,
// profiles/r02_cells_policy.txt, DESIGN.md 4.5.  The last commit that shipped them behind `cells_variant` is 3e687cf.
//   knn_cells_seed_kernel   round-2 chain: seeds + thresholds + pruning tables per query, behind knn_frag_kernel
//   knn_cells_sweep_kernel  round 3: match + scan + exact re-rank in one persistent kernel (lists in LDS)
//   (the scan with its norm tile out of an extra MFMA was a template flag of knn_cells_scan_kernel: see 3e687cf)
// Build check (in tools/arms/): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c cells_arms.hip -o /dev/null
#define KNN_NO_POOL
#include "../../multicore_hw2_amd/csrc/knn_cells.hip"
#include <atomic>
#include "cells_arms_body.inc"
