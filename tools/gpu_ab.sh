# A/B of two builds of the library on the same box: per-kernel average durations of the f32 bench, alternating A B A B.
# usage: bash tools/gpu_ab.sh TAG libA.so libB.so [kernel-name regex]
TAG=$1; A=$2; B=$3; PAT=${4:-wino32}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do for L in A B; do
  LIB=$A; [ $L = B ] && LIB=$B
  LASS_HIP_LIB=$R/lass_amd/csrc/$LIB timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$L$i -o x -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --modes none > $O/bench_$L$i.json 2> $O/err_$L$i.log || { tail -5 $O/err_$L$i.log; exit 1; }
done; done
cd $R
python3 - <<PY
import csv, glob, json, re
for tag in ("A1", "B1", "A2", "B2"):
    f = glob.glob("$O/%s/**/*kernel_stats.csv" % tag, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if re.search("$PAT", r["Name"])]
    rows.sort(key=lambda r: r["Name"])
    v = json.load(open("$O/bench_%s.json" % tag))
    print(tag, "clips/s %.1f" % v["value"], "conv_ms %.2f" % v["roofline"]["class_ms_per_step"], " | ".join("%s %.1f" % (re.sub(r".*wino32_kernel<([^>]*)>.*", r"\1", r["Name"])[:14], float(r["AverageNs"]) / 1e3) for r in rows))
PY
