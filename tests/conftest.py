import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def weights_pair():
    """(coarse, fine) synthetic state dicts, verified against the digest stored
    when the golden fixtures were generated from the reference."""
    from nerf_projects_amd import synthetic
    sd_c, sd_f = synthetic.synthetic_pair(0)
    dig = load_golden("weights_digest")
    assert synthetic.state_dict_digest(sd_c) == str(dig["digest_c"]), \
        "seeded weights differ from the build container's (fixtures would not apply)"
    assert synthetic.state_dict_digest(sd_f) == str(dig["digest_f"])
    return sd_c, sd_f


def check_sample_pdf(got, want, bins, weights, u, atol=2e-6):
    """Close, within the conditioning bound of inverse-CDF sampling, except at the
    reference's own discontinuities (see _oracle().sample_pdf_tolerance), where the sample
    must still fall inside the neighbouring bins. Returns the flagged fraction."""
    tol, mask, lo, hi = _oracle().sample_pdf_tolerance(bins, weights, u)
    bad = np.abs(got.astype(np.float64) - want) > atol + tol
    assert not np.any(bad & ~mask), (np.argwhere(bad & ~mask)[:5], np.abs(got - want)[bad & ~mask][:5])
    assert np.all((got >= lo - 1e-5) & (got <= hi + 1e-5))
    return mask.mean()



def _oracle():
    from oracle import nerf_oracle
    return nerf_oracle


from oracle.parity import FLIP, check_resampled, ray_errors, resampling_flips  # noqa: E402,F401  (shared with bench.py / smoke())
