R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp; rm -rf $O/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o run -- python $R/bench.py --config 5 --steps 6 --warmup 2 > $O/prof_c5.log 2>&1
cd $R; python - <<'PY'
import csv, glob, os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
f=glob.glob(os.path.join(O,"prof_c5","**","*kernel_stats.csv"),recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:110], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
rm -rf $O/prof_c5
