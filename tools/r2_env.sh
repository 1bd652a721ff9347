# usage: VAR=name VALS="0 1" bash tools/r2_env.sh  -> conv_bench per value of env var
cd $GRAFT_REPO_ROOT
for V in $VALS; do echo "== $VAR=$V"; env $VAR=$V timeout -k 10 200 python tools/conv_bench.py --iters 5 $CB_ARGS 2>/dev/null | grep -v "\.up"; done
