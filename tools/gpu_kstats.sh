# rocprofv3 kernel stats of a short bench (GPU box).  usage: bash tools/gpu_kstats.sh TAG [bench args...]
TAG=${1:-kstats}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o x -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline "$@" > $O/bench.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:26]: print(r["Name"][:120], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
