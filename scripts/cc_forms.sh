#!/bin/bash
# A/B of the experiment builds of the narrow kernels' row step (make -C bgt_amd/csrc ccform N=..) on the C2 cohort
for n in "" 1 2 5 7; do
    lib=bgt_amd/lib/libbgt_hip${n:+_cc$n}.so
    [ -f "$lib" ] || continue
    echo "== $lib"
    BGT_AMD_LIB=$PWD/$lib python scripts/c2_ab.py 2>&1 | grep -E "ballot|same"
done
