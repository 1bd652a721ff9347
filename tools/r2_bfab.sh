cd $GRAFT_REPO_ROOT
for N in 2 4 0; do
echo "== LASS_BF16_NPX=$N"
LASS_BF16_NPX=$N timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 --modes none 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('clips/s', round(d['value'],1), 'conv ms', round(d['roofline']['class_ms_per_step'],3), 'tconv', round(d['kernel_ms_per_step']['tconv_mfma'],3))"
done
