# one verb in a fresh process, timed part by part (tools/probes/prof_verb.py): bash tools/probes/run_prof.sh [C3|C2]
set -e
CFG=${1:-C3}
python3 - "$CFG" <<'PY'
import sys; sys.path.insert(0,'.')
from kmap_amd import synth
from kmap_amd.e2e import CONFIGS
c=CONFIGS[sys.argv[1]]
seq,b=synth.synth_reads(c['n_reads'],c['read_len'],c['seed'])
over={"kmer_count":{"min_k":6,"max_k":9},"motif_discovery":{"motif_pos_density_flag":False,"motif_co_occurence_flag":False,"gen_hamball_flag":False,"n_total_sample":c["n_total"],"n_motif_sample":c["n_motif"]},"visualization":{"gen_fig_flag":False,"random_seed":7,"n_max_iter":100}}
synth.write_res_dir('/tmp/resx',seq,b,over)
PY
python3 - <<'PY'
import subprocess, time, os, sys
env=dict(os.environ, PYTHONPATH='.')
for verb in ("scan_motif","visualize_kmers"):
    t0=time.time()
    r=subprocess.run([sys.executable,"-X","importtime","tools/probes/prof_verb.py",verb,"/tmp/resx"],env=env,capture_output=True,text=True)
    t1=time.time()
    end=[float(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("END_OF_SCRIPT")][0]
    print("\n".join(r.stdout.strip().splitlines()[:2]))
    print(f"  {verb}: process {t1-t0:.3f} s; start -> end of script {end-t0:.3f} s; teardown {t1-end:.3f} s")
    imp=[l for l in r.stderr.splitlines() if l.startswith("import time:")]
    top=sorted(((int(l.split("|")[1]), l.split("|")[2].strip()) for l in imp[1:] if not l.split("|")[2].startswith("   ")), reverse=True)[:8]
    print("  top-level imports (cumulative us):", top)
PY
rm -rf /tmp/resx
