cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --skip-configs --skip-fine --skip-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['warm_mvms_per_s'])"; done
