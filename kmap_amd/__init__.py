"""kmap_amd -- MI355X (gfx950) implementation of kmap's data-parallel hot path:
k-mer hashing/counting, the all-pairs Hamming matrix of sampled k-mers and the 2-D embedding
loop, as hand-written HIP kernels behind a C ABI (include/kmap_hip.h), with the reference's
`kmap preproc / scan_motif / visualize_kmers` CLI and file contracts on top."""
__version__ = "0.1.0"
from .kmer_count import FileNameDict  # noqa: F401


def main():
    """`kmap <verb> ...`.  The interpreter's normal way out is the default (atexit handlers, library destructors, a preloaded
    profiler's finalisation all run).  KMAP_FAST_EXIT=1 opts in to leaving a SUCCESSFUL verb through os._exit once its output is
    flushed: every file is complete and every writer thread joined by then, and unloading the HIP runtime with gigabytes still
    allocated costs 0.15 - 0.2 s per process (tools/probes/run_prof.sh).  Never under a torch.distributed launch (the process group
    wants its orderly shutdown), never with a tool preloaded (LD_PRELOAD / ROCP_TOOL_LIBRARIES / ROCPROFILER_*: its trace files are
    written at exit), and a flush that fails exits non-zero."""
    import os
    import sys
    from .cli import cli, display_paper_info
    display_paper_info()
    try:
        cli(prog_name="kmap")                       # click's standalone mode: always ends in SystemExit
    except SystemExit as e:
        ok = e.code in (None, 0)
        tooled = any(k == "LD_PRELOAD" or k == "ROCP_TOOL_LIBRARIES" or k.startswith("ROCPROFILER_") for k in os.environ)
        if ok and os.environ.get("KMAP_FAST_EXIT", "0") == "1" and not tooled and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
            try:
                sys.stdout.flush()
                sys.stderr.flush()
            except Exception:   # noqa: BLE001 -- e.g. BrokenPipeError: the output did not arrive, so this is not a success
                os._exit(1)
            os._exit(0)
        raise
