#!/bin/bash
# Round-5 closing run on the GPU box: the whole GPU suite, then the bench line (compact last line + detail file).
cd $GRAFT_REPO_ROOT; export O=$PWD/gpurun_out/final; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
( time python bench.py --detail $O/bench_detail.json > $O/bench.out 2> $O/bench.err ) 2>&1 | grep real
tail -1 $O/bench.out > $O/bench_line.json; wc -c $O/bench_line.json; cut -c1-600 $O/bench_line.json
