#!/usr/bin/env python3
"""The FASTA encoder on gzip input against Python's gzip reading the same stream (how much of the time is zlib's)."""
import sys, os, time, subprocess, tempfile, shutil
sys.path.insert(0,'.'); sys.path.insert(0,'tools/probes')
from time_preproc import write_fasta
from kmap_amd import kmer_count as kc
tmp=tempfile.mkdtemp()
try:
    fa=os.path.join(tmp,'r.fa'); size=write_fasta(fa, 2_000_000, 150)
    t=time.perf_counter(); subprocess.run(['gzip','-k','-6',fa],check=True); print(f'gzip -6: {time.perf_counter()-t:.1f} s, {os.path.getsize(fa+".gz")/1e6:.0f} MB from {size/1e6:.0f} MB')
    import gzip
    t=time.perf_counter(); n=0
    with gzip.open(fa+'.gz','rb') as fh:
        while True:
            b=fh.read(1<<22)
            if not b: break
            n+=len(b)
    print(f'python gzip read only: {time.perf_counter()-t:.2f} s')
    for _ in range(2):
        t=time.perf_counter(); a,b=kc.encode_fasta(fa+'.gz'); dt=time.perf_counter()-t
        print(f'encode_fasta(.gz): {dt:.2f} s = {size/1e6/dt:.0f} MB/s of text', a.shape)
    t=time.perf_counter(); a2,b2=kc.encode_fasta(fa); print(f'encode_fasta(plain): {time.perf_counter()-t:.3f} s', (a2==a).all())
finally:
    shutil.rmtree(tmp)
