# Kernel timeline of the evaluator (GPU box): gaps between kernels in the last DCASEEvaluator call of tools/eval_bench.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eval_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -o x -- python3 $R/tools/eval_bench.py 260 > $O/out.log 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
grep clips $O/out.log
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$O/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last evaluator call = host_mixing second call; take the last 17 stft2 launches
st = [i for i, r in enumerate(rows) if "stft2_kernel" in r[2] and "istft2" not in r[2]]
first = st[-17]
seg = rows[first:]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy = sum(r[1] - r[0] for r in seg)
print("last call: %d kernels, span %.1f ms, kernel busy %.1f ms" % (len(seg), (t1 - t0) / 1e6, busy / 1e6))
gaps = sorted(((seg[i + 1][0] - seg[i][1], seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1)), reverse=True)[:12]
for g, a, b in gaps: print("gap %.3f ms after %s before %s" % (g / 1e6, a, b))
# per batch: stft -> next stft
for a, b in zip(st[-17:], st[-16:]): print("batch %.2f ms" % ((rows[b][0] - rows[a][0]) / 1e6), end="; ")
print()
PY
