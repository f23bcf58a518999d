"""Opt-in kernel variants stay byte-identical to the default: they are selected by an environment variable that the library reads
once, so the parity cases run in ONE child interpreter with the variable set (H2W_EXPAND_VARIANT=3: expand_kernel_h)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hinted_expansion_kernel_parity():
    env = dict(os.environ, H2W_EXPAND_VARIANT="3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_batch.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "small_shapes or other_lookup_bits or valid_fri or sharding or config1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "deselected" in r.stdout
