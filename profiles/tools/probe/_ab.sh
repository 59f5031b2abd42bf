R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for m in 6 0 9 6 0; do echo "AMID_WGRAD_SPLIT=$m"; AMID_WGRAD_SPLIT=$m python3 bench.py --no-cpu-baseline --no-stress 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('window_ms_per_step'))"; done
