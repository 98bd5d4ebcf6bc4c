cd $GRAFT_REPO_ROOT; export O=$PWD/gpurun_out/t18; rm -rf $O; mkdir -p $O; Q=$PWD/scripts/quick_times.py
python -m pytest tests/test_dir_path.py tests/test_full_size.py tests/test_hip_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do python $Q c3 hrcsub c3half hrc c4 2>/dev/null >> $O/times.log; done
cat $O/times.log
