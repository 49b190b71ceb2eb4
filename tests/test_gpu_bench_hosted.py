"""bench.py's OWN multi-rank control flow on a one-GPU box: `bench.py --gpus N --hosted` runs N processes that share
GPU 0, with the collectives of np_hip_search_batch_sharded over the hosted (gloo) transport.  Everything the driver's 8-GPU run
executes in this script except `init_process_group("nccl")` and ncclCommInitRank / ncclAllGather themselves runs here: the
torchrun rendezvous, shard / replica-group arithmetic, one communicator per stream, the round-robin step order, status polling,
max-over-ranks timing, rank 0's JSON line -- and the merged result is compared with the CPU oracle on the whole corpus.
Needs a real MI355X."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nproc, extra, launcher=True):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    pre = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] if launcher else [sys.executable]
    cmd = pre + [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--hosted", "--docs", "120000",
           "--doc-len", "120", "--centroids", "8192", "--steps", "6", "--warmup", "2", "--cpu-queries", "0", "--parity-queries", "16",
           "--workspace-gib", "1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


def test_two_ranks_one_replica_group():
    d = _run(2, [])
    assert d["n_gpus"] == 2 and d["hosted"] is True and "NOT a scaling measurement" in d["hosted_note"]
    assert d["config"]["shards"] == 2 and d["config"]["replicas"] == 1 and d["config"]["docs_per_gpu"] == 60000
    assert "np_hip_search_batch_sharded" in d["config"]["parallelism"] and "hosted gloo" in d["config"]["parallelism"]
    assert d["value"] > 0 and abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    pv = d["parity_vs_oracle"]
    assert pv["queries"] == 16 and pv["topk_ids_identical"] == 16 and pv["max_rel_score_err"] < 2e-5, pv
    assert "2 ranks" in pv["through"]


def test_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run used to run one rank and label the line
    n_gpus: 1.  It now re-executes itself under the launcher: two ranks, one line, n_gpus 2, and the line names the transport the
    communicator really has (np_hip_comm_info)."""
    d = _run(2, [], launcher=False)
    assert d["n_gpus"] == 2 and d["config"]["shards"] == 2 and d["config"]["docs_per_gpu"] == 60000
    assert d["transport"] == "hosted" and d["rccl_ranks_seen"] == 0
    pv = d["parity_vs_oracle"]
    assert pv["topk_ids_identical"] == pv["queries"] == 16, pv


def test_four_ranks_two_replica_groups_of_two_shards():
    d = _run(4, ["--shards", "2", "--streams", "2"])
    assert d["n_gpus"] == 4 and d["config"]["shards"] == 2 and d["config"]["replicas"] == 2
    assert "x2 replica groups" in d["config"]["parallelism"]
    # every replica group answered `steps` batches of its own
    assert abs(d["value"] - 2 * 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    pv = d["parity_vs_oracle"]
    assert pv["topk_ids_identical"] == pv["queries"] == 16, pv
