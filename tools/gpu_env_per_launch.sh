# Per-launch durations of the last step under two values of an environment switch, side by side (rocprofv3 --kernel-trace, LASS_SPLIT=0).
# usage: bash tools/gpu_env_per_launch.sh VAR VALUE_A VALUE_B [bench args, default: f32 headline]
VAR=${1:-LASS_WINO4_NG}; VA=${2:-1}; VB=${3:-2}; shift 3 2>/dev/null
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5u; rm -rf $O; mkdir -p $O
for V in 1 2; do
  VAL=$VA; [ $V = 2 ] && VAL=$VB
  export $VAR=$VAL
  LASS_SPLIT=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t$V -o x -- python3 $R/bench.py --steps 3 --warmup 1 --modes none --no-cpu-baseline "$@" > $O/b$V.json 2> $O/e$V.log || { tail -3 $O/e$V.log; exit 1; }
done
cd $R
python3 - <<PY
import csv, glob, re
def last_step(tag):
    f = glob.glob("$O/t%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
    idx = [i for i, k in enumerate(ks) if "stft2_kernel" in k[0] and "istft" not in k[0]]
    end = [i for i, k in enumerate(ks) if "istft2_kernel" in k[0]]
    s = max(i for i in idx if i < end[-1])
    return ks[s:end[-1] + 1]
a, b = last_step("1"), last_step("2")
for (na, ta), (nb, tb) in zip(a, b):
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", nb).split("(")[0]
    print("%-40s A %8.1f  B %8.1f  %+5.1f %%" % (n[:40], ta, tb, (tb / ta - 1) * 100))
PY
rm -rf $O/t1 $O/t2
