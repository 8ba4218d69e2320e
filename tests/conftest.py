import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(GOLDEN / name, allow_pickle=False))
        return cache[name]

    return load


class MotifDef:
    def __init__(self, k, max_ham_dist, p_uniform, ratio_mu, ratio_std, ratio_cutoff):
        self.kmer_len, self.max_ham_dist, self.p_uniform = k, max_ham_dist, p_uniform
        self.ratio_mu, self.ratio_std, self.ratio_cutoff = ratio_mu, ratio_std, ratio_cutoff


@pytest.fixture(scope="session")
def motif_defs():
    """motif_def_table.csv written by the reference's preproc for tests/test.fa (data fixture)."""
    import csv
    out = {}
    with open(GOLDEN / "scan_testfa" / "motif_def_table.csv") as fh:
        for row in csv.DictReader(fh):
            k = int(row["kmer_len"])
            out[k] = MotifDef(k, int(row["max_ham_dist"]), float(row["p_uniform"]), float(row["ratio_mu"] or "nan"),
                              float(row["ratio_std"] or "nan"), float(row["ratio_cutoff"] or "nan"))
    return out
