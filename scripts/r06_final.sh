#!/bin/bash
# Round-6 closing run on the GPU box: the whole GPU suite, then the bench line (compact last line + detail file), then the
# product's multi-GPU path on two shards of the one device (what the driver's N > 1 run adds to the line).
cd $GRAFT_REPO_ROOT; export O=$PWD/gpurun_out/final6; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q --timeout 2400 > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
( time python bench.py --detail $O/bench_detail.json > $O/bench.out 2> $O/bench.err ) 2>&1 | grep real
tail -1 $O/bench.out > $O/bench_line.json; wc -c $O/bench_line.json; cut -c1-900 $O/bench_line.json
( time python bench.py --steps 5 --warmup 2 --no-secondary --cpu-sample 0 --no-counters --product-devices 0,0 --detail $O/bench_product_detail.json > $O/bench_product.out 2> $O/bench_product.err ) 2>&1 | grep real
tail -1 $O/bench_product.out | cut -c1-3000
