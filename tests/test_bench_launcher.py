"""CPU: bench.py's own multi-GPU launcher, end to end over gloo (world size 2): plain
`python bench.py --gpus 2 --dist-dry-run` must start two ranks itself, rendezvous, run the
statistics collective and print one JSON line with n_gpus == ranks_seen == 2."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_gloo():
    n, steps = 100003, 2
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", str(steps), "--reads", str(n), "--dist-dry-run"])
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["dry_run"] is True
    assert len(d["per_rank_reads_per_s"]) == 2
    # the slowest rank (0.75 s) sets the whole-job rate; every rank mapped n reads per step
    assert abs(d["value"] - 2 * n * steps / 0.75) < 1e-3
    assert d["per_rank_reads_per_s"] == [round(n * steps / 0.5, 1), round(n * steps / 0.75, 1)]
    total = 2 * n
    assert d["mapping"]["total"] == total
    assert d["mapping"]["unique"] == sum(1 for i in range(total) if i % 3 == 0)
    assert d["mapping"]["edits"] == total * (total - 1) // 2


def test_single_rank_needs_no_launcher():
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--reads", "1000", "--dist-dry-run"])
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_under_torchrun_and_world_size_mismatch_is_an_error():
    def base():
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", _free_port(), "bench.py"]
    ok = _run(base() + ["--gpus", "2", "--reads", "5000", "--dist-dry-run"])
    assert ok.returncode == 0, ok.stderr
    assert _line(ok.stdout)["ranks_seen"] == 2
    bad = _run(base() + ["--gpus", "4", "--reads", "5000", "--dist-dry-run"])
    assert bad.returncode != 0
    assert "WORLD_SIZE=2" in bad.stderr


def test_failed_rank_fails_the_launch():
    # a worker that dies (here: an impossible read count) must surface as a non-zero exit of the parent
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--reads", "-5", "--dist-dry-run"])
    assert r.returncode != 0
