O=$GRAFT_REPO_ROOT/gpurun_out/r2d; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "graph" > $O/test.log 2>&1; tail -15 $O/test.log
for G in 1 0; do
LASS_GRAPH=$G timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --modes bf16 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
ks=sum(d['kernel_ms_per_step'].values())
print('LASS_GRAPH=$G clips/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'sum kernel-class ms', round(ks,3), 'gap', round(d['ms_per_step']-ks,3), d['launch']['captures'], d['launch']['replays_in_headline_loops'], '| bf16', round(d['modes']['bf16']['clips_s'],1), round(d['modes']['bf16']['ms_per_step'],3), round(sum(d['modes']['bf16']['kernel_ms_per_step'].values()),3))"
done
