"""tests/util.run_ranks, the harness of the two-process GPU tests, on CPU children: output comes back from files, a rank that exits
non-zero fails at once (its peer is killed, both tails are in the message), a hang is killed at the deadline and retried once."""
import sys
import warnings

import pytest

from tests import util


def test_both_ranks_output_is_returned(tmp_path):
    codes, outs, _ = util.run_ranks([sys.executable, "-c", "import os; print('rank', os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_PORT'])"],
                                    2, str(tmp_path))
    assert codes == [0, 0] and outs[0].startswith("rank 0 2 ") and outs[1].startswith("rank 1 2 ")
    assert outs[0].split()[-1] == outs[1].split()[-1], "one rendezvous port for both"


def test_a_crashed_rank_fails_at_once_and_its_peer_is_killed(tmp_path):
    with pytest.raises(pytest.fail.Exception) as e:
        util.run_ranks([sys.executable, "-c", "import os, sys, time; sys.stderr.write('boom'); sys.exit(3) if os.environ['RANK'] == '1' else time.sleep(60)"],
                       2, str(tmp_path), deadline_s=20)
    assert "rank 1: rc 3" in str(e.value) and "boom" in str(e.value) and "killed at the deadline" in str(e.value)


def test_a_hang_is_killed_at_the_deadline_and_retried_once(tmp_path, monkeypatch):
    monkeypatch.setattr(util, "_RETRIES_LEFT", [1])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with pytest.raises(pytest.fail.Exception) as e:
            util.run_ranks([sys.executable, "-c", "import time; time.sleep(60)"], 2, str(tmp_path), deadline_s=2)
    assert len(w) == 1 and "attempt 0" in str(w[0].message) and "attempt 1" in str(e.value)
    assert util._RETRIES_LEFT == [0]
