# one gpurun call: launch-time trend of the headline kernel (clock ramp?) + the R x LDS-throttle sweep, one process per variant
cd $GRAFT_REPO_ROOT
echo "== default, back to back"; python tools/hamdist_trend.py 50000 200
echo "== default, after 0.5 s idle"; python tools/hamdist_trend.py 50000 100 0.5
for r in 4 8; do for l in 30 40 50 60 76; do
  echo "== R=$r lds=${l}K"; KMAP_HAMDIST_TILE_R=$r KMAP_HAMDIST_TILE_LDS_KB=$l python tools/hamdist_trend.py 50000 100 | tail -2
done; done
