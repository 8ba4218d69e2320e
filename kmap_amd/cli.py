"""`kmap` command line: the three verbs of the reference's CLI that sit on the GPU hot path
(reference cli.py:9-36, kmer_count.py:70-101, motif_discovery.py:29-53, visualization.py:18-33),
with the same option names, plus `ex_hamball` (motif_discovery.py:74-108; Hamming-ball extraction on the GPU).  The
reference's plotting / alignment / reporting verbs (draw_logo, align_conseq, extract_motif_locations, plot_network,
check_motif_co_occurence) are out of scope here.  `scan_motif` and `visualize_kmers` shard over the
GPUs of a node when launched through `python -m torch.distributed.run --nproc-per-node G -m kmap_amd <verb> ...`."""
import click

from . import __version__


@click.group()
def cli():
    """KMAP on MI355X: k-mer counting, motif scan and 2-D k-mer embedding (HIP kernels)."""


def display_paper_info():
    print()
    print(f"kmap_amd {__version__} -- MI355X/gfx950 implementation of the kmap hot path")
    print("method: KMAP, Fu et al., bioRxiv 2024, doi:10.1101/2024.04.12.589197")
    print()


@cli.command(name="preproc")
@click.option("--fasta_file", type=str, required=True, help="Input fasta file")
@click.option("--res_dir", type=str, default=".", required=False, help="Result directory for storing all outputs")
@click.option("--gpu_mode", type=bool, default=True, required=False, help="kept for CLI compatibility (always GPU)")
@click.option("--debug", type=bool, default=False, required=False, help="display debug information.")
def preproc(fasta_file, res_dir=".", gpu_mode=True, debug=False):
    from .kmer_count import _preproc
    _preproc(fasta_file, res_dir, debug)


@cli.command(name="scan_motif")
@click.option("--res_dir", type=str, required=True, help="Result directory for storing all outputs")
@click.option("--gpu_mode", type=bool, default=True, required=False, help="kept for CLI compatibility (always GPU)")
@click.option("--debug", type=bool, default=False, required=False, help="display debug information.")
def scan_motif(res_dir, gpu_mode=True, debug=False):
    from .motif_discovery import _scan_motif
    _scan_motif(res_dir, debug)


@cli.command(name="visualize_kmers")
@click.option("--res_dir", type=str, required=True, help="Result directory for storing all outputs")
@click.option("--debug", type=bool, default=False, required=False, help="display debug information.")
def visualize_kmers(res_dir, debug=False):
    from .visualization import _visualize_kmers
    _visualize_kmers(res_dir, debug)


@cli.command(name="ex_hamball")
@click.option("--res_dir", type=str, required=True, help="Result directory for storing all outputs")
@click.option("--conseq", type=str, required=True, help="the consensus sequence")
@click.option("--return_type", type=str, required=True, help='output file form, can be ["hash" | "kmer" | "matrix"]')
@click.option("--output_file", type=str, required=True, help="output file name, including the suffix")
@click.option("--max_ham_dist", type=int, default=-1, required=False,
              help="The radius of the Hamming ball. -1 means taking the radius from motif_def_table.csv")
def ex_hamball(res_dir, conseq, return_type, output_file, max_ham_dist=-1):
    from .reports import _ex_hamball
    _ex_hamball(res_dir, conseq, return_type, output_file, max_ham_dist)
