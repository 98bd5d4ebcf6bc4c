#!/bin/bash
# A/B of the experiment builds of the narrow kernels' row step (make -C bgt_amd/csrc ccform N=.. [CMPX=1]) on the C2 cohort
for lib in bgt_amd/lib/libbgt_hip.so bgt_amd/lib/libbgt_hip_cc*.so; do
    [ -f "$lib" ] || continue
    echo "== $lib"
    BGT_AMD_LIB=$PWD/$lib python scripts/c2_ab.py 2>&1 | grep -E "ballot|same"
done
