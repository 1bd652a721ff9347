# A/B of two builds of the library on the same box, any compute mode: plain bench runs alternating A B A B (clips/s), then ONE
# rocprofv3 --kernel-trace pass per build with the per-launch durations of the last step side by side.
# usage: bash tools/gpu_ab2.sh TAG libA.so libB.so [bench args, default: --dtype bf16]
TAG=$1; A=$2; B=$3; shift 3
ARGS=${@:---dtype bf16}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for i in 1 2; do for L in A B; do
  LIB=$A; [ $L = B ] && LIB=$B
  LASS_HIP_LIB=$R/lass_amd/csrc/$LIB timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --modes none $ARGS > $O/bench_$L$i.json 2> $O/err_$L$i.log || { tail -5 $O/err_$L$i.log; exit 1; }
  python3 -c "
import json; v = json.load(open('$O/bench_$L$i.json')); print('$L$i $LIB clips/s %.1f ms/step %.3f conv %.3f tconv %.3f' % (v['value'], v['ms_per_step'], v['kernel_ms_per_step']['conv3x3_mfma'], v['kernel_ms_per_step']['tconv_mfma']), flush=True)"
done; done
cd /tmp && export TMPDIR=/tmp
for L in A B; do
  LIB=$A; [ $L = B ] && LIB=$B
  LASS_SPLIT=0 LASS_HIP_LIB=$R/lass_amd/csrc/$LIB timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$L -o x -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --modes none $ARGS > $O/tbench_$L.json 2> $O/terr_$L.log || { tail -5 $O/terr_$L.log; exit 1; }
done
cd $R
python3 - <<PY
import csv, glob, re
def last_step(tag):
    f = glob.glob("$O/trace_%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
    # one step = from the last stft2_kernel launch to the last istft2_kernel launch
    idx = [i for i, k in enumerate(ks) if k[0].startswith("stft2_kernel") or "stft2_kernel" in k[0] and "istft" not in k[0]]
    end = [i for i, k in enumerate(ks) if "istft2_kernel" in k[0]]
    s = max(i for i in idx if i < end[-1])
    return ks[s:end[-1] + 1]
a, b = last_step("A"), last_step("B")
short = lambda n: re.sub(r"^void |\(.*", "", n)[:64]
print("%-66s %9s   %-66s %9s" % ("A", "us", "B", "us"))
for i in range(max(len(a), len(b))):
    ka = a[i] if i < len(a) else ("", 0.0); kb = b[i] if i < len(b) else ("", 0.0)
    print("%-66s %9.1f   %-66s %9.1f" % (short(ka[0]), ka[1], short(kb[0]), kb[1]))
print("sum A %.1f us, sum B %.1f us" % (sum(k[1] for k in a), sum(k[1] for k in b)))
PY
