"""RCCL on the box: a process group of ONE rank (backend nccl) carries real frame records through the collective branch of the
sequence-level exchange, in a fresh child process (tests/rccl_one_rank.py) that also holds liborbfe_hip.so -- the 1-GPU rehearsal of
BASELINE config 4's gather (example/Stereo/KittiStereo.cc:28-37 sharded over ranks).  The N > 1 ordering logic is covered on the CPU by
tests/test_dist_gloo.py (world size 2, gloo)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_gathers_real_records_through_the_collective_branch():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_one_rank.py"), "40", "16"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1].split() == ["RCCL_ONE_RANK_OK", "40", "3"]
