"""CPU: host-side logic of the VI driver (spmf_amd/vi.py) -- the plateau /
checkpoint controller against a scripted loss sequence (behaviour evidenced by
the reference notebooks' stdout, notebooks/factorizing_random_noise.ipynb:122-420)
and batch-row counting for the legacy calibrate_advi entry."""
import math

import numpy as np
import pytest
import scipy.sparse as sp


def test_plateau_controller_scripted_sequence():
    from spmf_amd.vi import PlateauController
    c = PlateauController(0.05, rel_tol=1e-4, abs_tol=1e-10, max_decay_steps=3, lr_decay_factor=0.99)
    script = [
        (44.13, "improved"),   # first epoch: best was +inf -> checkpoint ("Saved a checkpoint")
        (43.00, "improved"),
        (43.50, "plateau"),    # worse than best: lr 0.05 -> 0.0495, restore best
        (42.00, "improved"),   # best is still 43.00 after the restore: 42 improves on it
        (42.10, "plateau"),    # lr -> 0.049005  (the notebook's first reported plateau value)
        (42.20, "quit"),       # third reset with max_decay_steps = 3: "We have reset 3 times so quitting"
    ]
    lrs = []
    for loss, want in script:
        assert c.update(loss) == want, (loss, want)
        lrs.append(c.lr)
    assert lrs[:2] == [0.05, 0.05]
    assert abs(lrs[2] - 0.05 * 0.99) < 1e-15
    assert abs(lrs[4] - 0.049005) < 1e-12          # notebooks/factorize_linear_structure.ipynb:216
    assert abs(lrs[5] - 0.05 * 0.99 ** 3) < 1e-15
    assert c.best == 42.00 and c.decays == 3


def test_plateau_controller_tolerance_stop():
    from spmf_amd.vi import PlateauController
    c = PlateauController(0.1, rel_tol=1e-4, abs_tol=1e-10)
    assert c.update(100.0) == "improved"
    assert c.update(99.0) == "improved"
    assert c.update(99.0 - 99.0 * 0.5e-4) == "converged"      # relative gain 0.5e-4 < rel_tol
    d = PlateauController(0.1, rel_tol=0.0, abs_tol=1e-3)
    assert d.update(5.0) == "improved"
    assert d.update(5.0 - 5e-4) == "converged"                # absolute gain < abs_tol
    e = PlateauController(0.1, rel_tol=1e-4, abs_tol=1e-10, max_decay_steps=25)
    e.update(1.0)
    for i in range(24):
        assert e.update(2.0) == "plateau"
    assert e.update(2.0) == "quit" and e.decays == 25         # "We have reset 25 times so quitting"
    assert abs(e.lr - 0.1 * 0.99 ** 25) < 1e-15


class _FakeCounts:
    def __init__(self, n_rows, panel_rows):
        self.n_rows, self.panel_rows = n_rows, panel_rows
        self.n_panels = -(-n_rows // panel_rows)


class _M:
    count_key = "counts"


def test_batch_rows_counts_the_panel_range_not_the_shard():
    """calibrate_advi's dataset size: a {'counts': shard, 'panels': (p0, p1)} batch is
    worth the rows of its panels (round 1 added the whole shard once per batch)."""
    from spmf_amd.vi import batch_rows
    sc = _FakeCounts(1050, 100)
    batches = [{"counts": sc, "panels": (p, p + 2)} for p in range(0, sc.n_panels, 2)]
    assert [batch_rows(_M, b) for b in batches] == [200, 200, 200, 200, 200, 50]
    assert sum(batch_rows(_M, b) for b in batches) == 1050
    assert batch_rows(_M, {"counts": sc}) == 1050
    assert batch_rows(_M, {"counts": sc, "panels": (9, None)}) == 150
    assert batch_rows(_M, {"counts": np.zeros((7, 3))}) == 7
    assert batch_rows(_M, sp.csr_matrix(np.eye(4))) == 4
    assert batch_rows(_M, {"counts": (None, None, None, (12, 5))}) == 12
