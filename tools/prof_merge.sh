cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_merge
KMD_MERGE_PATH=fast timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_merge -o m -- python3 tools/kbench_merge.py --iters 3 "$@" > gpurun_out/merge_prof.log 2>&1 < /dev/null
grep "merge S" gpurun_out/merge_prof.log
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_merge/**/*kernel_stats.csv', recursive=True)+glob.glob('gpurun_out/prof_merge/*kernel_stats.csv'):
    for r in list(csv.reader(open(f)))[:5]:
        print(r[0][:50], r[1:4])
    break
PY
