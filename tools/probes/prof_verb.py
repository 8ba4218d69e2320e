import sys, time
t00=time.perf_counter()
verb, res = sys.argv[1], sys.argv[2]
import numpy as np
t_np=time.perf_counter()
from kmap_amd import _ffi
import kmap_amd.motif_discovery as md, kmap_amd.visualization as vz
t_imp=time.perf_counter()
lib=_ffi.lib()
t_lib=time.perf_counter()
import ctypes as C
n=C.c_int(0)
try:
    lib.kmap_device_count(C.byref(n))
except Exception as e:
    print('devcount', e)
t_dev=time.perf_counter()
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    if verb=='scan_motif':
        np.random.seed(123); md._scan_motif(res)
    else:
        vz._visualize_kmers(res)
t_run=time.perf_counter()
st = dict(md.STAGE_TIMES); st.update({'viz_'+k:v for k,v in vz.STAGE_TIMES.items()})
print(verb, f"numpy {t_np-t00:.3f} imports {t_imp-t_np:.3f} lib {t_lib-t_imp:.3f} hipinit {t_dev-t_lib:.3f} run {t_run-t_dev:.3f}; stage sum {sum(v for k,v in st.items() if not k.startswith('bg_') and k not in ('find_motif',)):.3f}")
print({k:round(v,3) for k,v in st.items()})
print('modules:', 'scipy.stats' in sys.modules, 'pandas' in sys.modules, 'torch' in sys.modules)
import time as _t
print("END_OF_SCRIPT", repr(_t.time()), flush=True)
