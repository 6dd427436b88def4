"""CPU (not gpu): first contact with the N > 1 launch path of bench.py without hardware — `python bench.py --gpus 2` with no
launcher in the environment must start its own ranks through torch.distributed.run as CHILD processes, never touch the GPU in
the parent, pass exactly ONE JSON line (rank 0's) through on stdout, and hand a failing rank's exit code back.
The training step is replaced by bench.py's `--mock-step` body (sleep + a small gloo all-reduce); everything around it — port
choice, rendezvous on 127.0.0.1, env plumbing, stdout filtering, rc — is the production code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, gpus=2):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    # the parent must not make a GPU call: poison torch.cuda's lazy initialisation in THIS interpreter only (sitecustomize-style
    # hook through PYTHONSTARTUP is not honoured by `python script.py`, so wrap the call)
    code = (
        "import sys, os, runpy, torch\n"
        "def _boom(*a, **k): raise RuntimeError('parent process touched the GPU')\n"
        "torch.cuda._lazy_init = _boom; torch.cuda.set_device = _boom; torch.cuda.init = _boom\n"
        f"sys.argv = ['bench.py', '--gpus', '{gpus}', '--steps', '5', '--warmup', '1', '--mock-step']\n"
        f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)


def test_bench_self_launch_two_ranks_one_json_line():
    r = _run()
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, ("stdout must carry exactly one line", r.stdout)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_reported"] == 2 and rec["steps"] == 5 and rec["allreduce_ok"] is True
    assert rec["value"] > 0 and rec["scaling"] == "weak"
    assert "launching 2 ranks via torch.distributed.run" in r.stderr


def test_bench_self_launch_passes_a_failing_rank_through():
    r = _run({"VACNIC_BENCH_MOCK_FAIL": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no result line when a rank failed"


def test_bench_under_an_external_launcher_does_not_relaunch():
    """the driver's form: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` (WORLD_SIZE set by the launcher)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--mock-step"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(recs) == 1 and recs[0]["world_size_reported"] == 2
    assert "launching" not in r.stderr
