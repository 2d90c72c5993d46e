cd /tmp; export TMPDIR=/tmp
for fs in 0 2; do
  export QP_ATTN_FORCE_SPLIT=$fs
  QP_SHAPES=cfg4 python /root/repo/tools/bench_attn.py 8 8 2>&1 | grep variant
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$c; QP_SHAPES=cfg4 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm_$c -o t -- python /root/repo/tools/bench_attn.py 8 > /dev/null 2>&1
    python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pm_$c/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k='attn' if 'attn_fwd_kernel_s6' in r['Kernel_Name'] else ('combine' if 'attn_combine' in r['Kernel_Name'] else None)
        if k: acc[k].append(float(r['Counter_Value']))
for k,v in acc.items():
    v=sorted(v)[len(v)//2:]
    print('force_split=$fs $c', k, 'median-upper mean KB per launch: %.0f'%(sum(v)/len(v)), 'n', len(v))
PY
  done
done
