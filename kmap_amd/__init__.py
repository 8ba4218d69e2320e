"""kmap_amd -- MI355X (gfx950) implementation of kmap's data-parallel hot path:
k-mer hashing/counting, the all-pairs Hamming matrix of sampled k-mers and the 2-D embedding
loop, as hand-written HIP kernels behind a C ABI (include/kmap_hip.h), with the reference's
`kmap preproc / scan_motif / visualize_kmers` CLI and file contracts on top."""
__version__ = "0.1.0"
from .kmer_count import FileNameDict  # noqa: F401


def main():
    from .cli import cli, display_paper_info
    display_paper_info()
    cli(prog_name="kmap")
