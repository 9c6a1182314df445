#!/bin/bash
# timing-only experiments on the bf16x6 kernels (results are wrong in the NO* variants)
for V in "$@"; do
  make -C marlclassification_amd/csrc clean > /dev/null; make -C marlclassification_amd/csrc -j8 EXTRA="$V" 2>&1 | grep -E "error|Error"
  echo "== variant [$V]"
  python tools/gemm_split_bench.py 2>&1 | grep "split.: 1" | grep "65536, .n.: 256, .k.: 1024\|624\|'nj': 256" | cut -c1-110
done
