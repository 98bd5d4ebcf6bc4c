cd $GRAFT_REPO_ROOT; export O=$PWD/gpurun_out/t6; rm -rf $O; mkdir -p $O; L=$PWD/bgt_amd/lib; Q=$PWD/scripts/quick_times.py
python -m pytest tests/test_dir_path.py tests/test_full_size.py -m gpu -x -q -k "plane or subsets" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do echo "== base" >> $O/plane.log; BGT_AMD_LIB=$L/libbgt_hip_base.so python $Q c3 hrcsub c3half 2>/dev/null >> $O/plane.log; echo "== new (896 threads at C3)" >> $O/plane.log; python $Q c3 hrcsub c3half 2>/dev/null >> $O/plane.log; done
cat $O/plane.log
export BGT_AMD_LIB=$L/libbgt_hip_ablate.so
for sh in 11 9 8 7; do for v in unsorted sorted; do
  echo "== sub_shift $sh $v" >> $O/subshift.log
  if [ $v = sorted ]; then BGTH_SUB_SHIFT=$sh BGTH_SORTED_SLOTS=1 python $Q c2 small 2>/dev/null >> $O/subshift.log; else BGTH_SUB_SHIFT=$sh python $Q c2 small 2>/dev/null >> $O/subshift.log; fi
done; done
cat $O/subshift.log
