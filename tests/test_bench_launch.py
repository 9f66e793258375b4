"""bench.py's N > 1 start-up path, exercised without a GPU (round-4 review: `python3 bench.py --gpus N` used to SystemExit).

V2E_AMD_BENCH_STUB=1 swaps the engine for tests/bench_stub.py and the backend for gloo; everything else -- the self-launch
under torch.distributed.run, the process group, run_steps, the event-stream all-gather, the MAX / SUM reductions over ranks
and the single JSON line from rank 0 -- is bench.py's own code.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["V2E_AMD_BENCH_STUB"] = "1"
    env["OMP_NUM_THREADS"] = "1"
    return env


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must hold the JSON line and nothing else: %r" % lines
    return json.loads(lines[0])


def _check_line(d, n, steps, warmup):
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["steps"] == steps and d["warmup"] == warmup
    assert d["data"] == "stub" and d["metric"].startswith("STUB"), "a stub run must never look like a measurement"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0
    if n > 1:
        assert d["config"]["event_stream_allgather"] is True
        assert d["compute_only"]["value"] > 0 and d["with_allgather"]["bytes_gathered_per_rank_per_step"] > 0
        # the SuperSloMo stage of one clip sharded over the ranks by source pairs: every rank went through the leg's collectives in
        # the same order (no hang), and rank 0 assembled the unsharded clip (the stub interpolator's frames, compared in the leg)
        sh = d["slomo_sharded"]
        assert "error" not in sh, sh
        assert sh["ranks"] == n and sh["scaling"] == "strong" and sh["frames_gathered_in_order"] is True and sh["value"] > 0
        assert sh["max_abs_diff_vs_unsharded"] == 0 and sh["bytes_received_by_owner_per_clip"] == 10 * 3 * 6 * 8 * (n - 1) // n
    # the stub's event counts are a known function of (rank, step): SUM over ranks of the median block's steps
    from tests.bench_stub import stub_counts
    F = d["config"]["frames_per_step"]
    per_frame = d["events_per_frame"]
    assert 3 <= per_frame <= 3 + n + 4 + 255 / F, per_frame  # 3 + rank + (step + f) % 5 (+ first pixel of the step's frames / F)
    assert stub_counts(0, 0, F, 0).sum() > 0


@pytest.mark.parametrize("n", [2, 4])
def test_bare_bench_py_self_launches_ranks(n):
    """`python bench.py --gpus N` exactly as BENCH_r04.json's cmd was formed, with N > 1 and no launcher around it."""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "3", "--warmup", "1", "--blocks", "2"],
                       cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    _check_line(_one_json_line(p.stdout.decode()), n, 3, 1)


def test_bench_py_under_torch_distributed_run():
    """The driver's documented N > 1 command line (python -m torch.distributed.run ... bench.py --gpus N ...)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--blocks", "1"], cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = [json.loads(ln) for ln in p.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(d) == 1  # rank 0 only
    _check_line(d[0], 2, 2, 1)


def test_bare_bench_py_single_rank_stub():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--blocks", "1"],
                       cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = _one_json_line(p.stdout.decode())
    _check_line(d, 1, 2, 1)
    assert d["collective_backend"] is None


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--blocks", "1"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr
