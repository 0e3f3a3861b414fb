O=gpurun_out/r02l; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dp -o dp -- python $R/scripts/dp_overhead.py 64 f32 > $R/$O/dp.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02l/prof_dp/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:6]+[r for r in rows if 'ccl' in r['Name'].lower() or 'nccl' in r['Name'].lower() or 'rccl' in r['Name'].lower()]:
    print(r['Name'][:80], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
find $O -name "*kernel_trace.csv" -size +30M -delete; find $O -name "*.db" -delete
tail -2 $O/dp.log
