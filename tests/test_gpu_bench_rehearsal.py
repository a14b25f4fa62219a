"""bench.py's multi-rank flow rehearsed on ONE GPU (--same-device: every rank uses GPU 0, the border messages move by host
staging over the control plane instead of RCCL, which cannot form a communicator between ranks on one device).  What it
pins: the N > 1 line carries the per-rank pair gate (every rank against the oracle on its tile + ring) and the per-rank
diagnostics, and the run fails loudly on a mismatch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_rank_rehearsal_line_has_pair_gate_and_per_rank_block():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--sectors", "16", "--workload", "config5", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=550, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rehearsal_same_device"] and d["scaling"] == "weak"
    p = d["parity_in_run"]
    assert p["ok"] and p["all_ranks_ok"] and p["pairs_equal"] and p["visible_equal"] and p["matrices_equal"] and p["border_lost"] == 0
    assert p["pairs"] > 10
    pr = d["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["border_lost"] == 0 for x in pr)
    assert d["value"] > 0 and d["config"]["entities_total"] == 2 * 8 * 16 * 32
    # result assembly: the ranks' visible lists at their offsets are the concatenation of the ranks' oracle lists
    va = d["visible_assembly"]
    assert va["totals_agree"] and va["every_slot_filled"] and va["equals_oracle_concatenation"]
    assert va["visible_total"] == sum(x["visible"] for x in pr) and va["offsets"] == [0, pr[0]["visible"]]
