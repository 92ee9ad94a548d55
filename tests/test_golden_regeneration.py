"""CPU, only where the reference is mounted: committed fixtures are what `oracle/gen_golden.py` produces from the
unmodified reference TODAY -- so that "the oracle is pinned to reference outputs" stays checkable by anyone with the
reference (one fixture had been seeded with a per-process str hash and could not be regenerated).
Regenerates four small fixtures into a scratch directory and compares every array bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference not mounted")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("what,files", [
    ("recurrent", ["recurrent_nets.npz"]),
    ("sample_stack", ["sample_stack.npz"]),
    ("sample_random", ["sample_random.npz"]),
    ("ppo_sched", ["ppo_sched.npz"]),
    ("depth", ["sac_depth3.npz", "sac_depth1.npz", "td3_depth4.npz", "td3_ddpg_depth1.npz", "dsac_depth3.npz", "redq_depth1.npz",
               "sac_bounded.npz", "sac_bounded_depth3.npz", "npg_npg_relu3.npz", "npg_trpo_tanh1.npz", "sac_tanh.npz", "td3_tanh3.npz"]),
])
def test_fixture_regenerates_bit_for_bit(tmp_path, what, files):
    env = dict(os.environ, TS_GOLDEN_OUT=str(tmp_path), PYTHONHASHSEED="random")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), what], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for f in files:
        new, old = np.load(os.path.join(tmp_path, f)), np.load(os.path.join(GOLDEN, f))
        assert sorted(new.files) == sorted(old.files), f
        same = lambda a, b: np.array_equal(a, b, equal_nan=a.dtype.kind == "f")      # noqa: E731 (string arrays: no isnan)
        bad = [k for k in old.files if not same(new[k], old[k])]
        assert not bad, (f, bad[:5])
