# sweep of the tiled Hamming kernel's rows per block and LDS throttle (blocks per CU) on one MI355X
R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --no-cpu-baseline --e2e none --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%s kernel_ms=%.4f frac=%.3f' % (sys.argv[1], d['roofline']['kernel_ms'], d['roofline']['frac']))" "$1"; }
KMAP_HAMDIST_TILE=0 run "old kernel           "
for r in 2 4 8; do for l in 30 34 40 45 50; do
  KMAP_HAMDIST_TILE_R=$r KMAP_HAMDIST_TILE_LDS_KB=$l run "tile R=$r lds=${l}K"
done; done
