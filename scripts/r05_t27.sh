cd $GRAFT_REPO_ROOT
export BGT_AMD_LIB=$PWD/bgt_amd/lib/libbgt_hip_ablate.so
python - <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, bgt_amd
for samples, sites, seed in ((100000, 153 * 8192, 4), (32488, 142000, 7)):
    m = 2 * samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed); pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens); del rle
    rd = bgt_amd.HipReader(pbf); bgt_amd.force_kernels(128)
    for skip in (0, 0x200000, 0x10000, 0x210000):
        os.environ['BGTH_DEBUG_SKIP'] = str(skip)
        best = None; pm = None
        for _ in range(2):
            rd.scan(0, sites); t = rd.timing()['scan_ms']; p = rd.path()['producer_ms']
            if best is None or t < best: best, pm = t, p
        print('m=%d skip=%#x: total %.2f ms, producer %.2f ms' % (m, skip, best, pm), flush=True)
    rd.close(); pbf.close()
PY
