"""A fixed slice of tools/fuzz_parity.py inside the suite: random geometries (sizes, k, h, layouts, N content, skew,
scratch caps, spaced seeds, counting filters), direct vs partitioned vs auto, a third of the cases with the stateful
leg (second insert, lazy clear, fresh insert, query on the same filter)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fixed_fuzz_slice(seed):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.zeros(1, device="cuda")
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(seed)
    bad = []
    for it in range(25):
        ok, desc = fz.one_case(rng, it)
        if not ok:
            bad.append(desc)
    assert not bad, bad
