#!/bin/bash
# VERDICT r4 item 4 (profiles/r05_lds): slots in plane-0 rank order per sub-block (profiling build, BGTH_SORTED_SLOTS=1) against the shipped
# slot order: kernel times, then the LDS counters of the same scans.  GPU box: bash scripts/r05_sorted.sh
cd $GRAFT_REPO_ROOT; export O=$PWD/gpurun_out/r05_sorted; rm -rf $O; mkdir -p $O
export BGT_AMD_LIB=$PWD/bgt_amd/lib/libbgt_hip_ablate.so
Q=$PWD/scripts/quick_times.py
for i in 1 2; do
  echo "== unsorted" >> $O/times.log; python $Q c2 hrc small c4 2>/dev/null >> $O/times.log
  echo "== sorted" >> $O/times.log; BGTH_SORTED_SLOTS=1 python $Q c2 hrc small c4 2>/dev/null >> $O/times.log
done
cat $O/times.log
cd /tmp && export TMPDIR=/tmp
for v in unsorted sorted; do
  if [ $v = sorted ]; then export BGTH_SORTED_SLOTS=1; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc_$v -- python $Q c2 hrc c4 > $O/pmc_$v.log 2>&1
done
python - <<'PY' | tee $O/counters.txt
import csv,glob,os,collections
O=os.environ['O']
for v in ('unsorted','sorted'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O,'pmc_'+v,'**','*counter_collection.csv'),recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if ('scan_kernel<1024, 20' in k or 'walk_kernel' in k) :
                acc[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in sorted(acc.items()):
        t={n:sum(sorted(x)[-2:])/len(sorted(x)[-2:]) for n,x in c.items()}
        print('%-9s %-48s LDS_IDX_ACTIVE %.4g  BANK_CONFLICT %.4g (%.2f)  INSTS_LDS %.4g  INSTS_VALU %.4g  BUSY_CYCLES %.4g' % (
            v, k, t['SQ_LDS_IDX_ACTIVE'], t['SQ_LDS_BANK_CONFLICT'], t['SQ_LDS_BANK_CONFLICT']/t['SQ_LDS_IDX_ACTIVE'], t['SQ_INSTS_LDS'], t['SQ_INSTS_VALU'], t['SQ_BUSY_CYCLES']))
PY
rm -rf $O/pmc_unsorted $O/pmc_sorted
