"""The producer / consumer form of the backward blend (render_bwd_pc, GSR_BWD_PC=1: round 5, opt-in) must write exactly the records the
barrier form writes: every gradient of a multi-view fwd + bwd bit for bit, all colour modes.  The launcher reads the switch once per
process, so each arm runs in its own interpreter (tools/r05_pc_check.py is the worker)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tools", "r05_pc_check.py")


def _grads(tmp_path, pc, V, P, S, frozen):
    out = str(tmp_path / f"g{pc}.npz")
    env = dict(os.environ, GSR_BWD_PC=str(pc), FROZEN=str(int(frozen)))
    subprocess.run([sys.executable, WORKER, out, str(V), str(P), str(S)], check=True, env=env, timeout=300,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize("V,P,S,frozen", [(2, 20000, 400, False), (3, 5000, 200, True), (1, 30000, 304, False)])
def test_producer_consumer_backward_is_bit_identical(tmp_path, V, P, S, frozen):
    a = _grads(tmp_path, 0, V, P, S, frozen)
    b = _grads(tmp_path, 1, V, P, S, frozen)
    assert set(a.files) == set(b.files) and len(a.files) >= 5
    for k in a.files:
        assert np.array_equal(a[k], b[k]), f"{k}: producer / consumer backward differs from the barrier form"
