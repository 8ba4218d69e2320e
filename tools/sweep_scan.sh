R=$GRAFT_REPO_ROOT; cd $R
for l in 8 16 24 32 64; do echo "LDS_KB=$l"; KMAP_SCAN_LDS_KB=$l python tools/bench_scan.py --reps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['scan']['s_per_pass_device'])"; done
