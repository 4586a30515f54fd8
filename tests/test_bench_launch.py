"""CPU: `python bench.py --gpus N` started bare must spawn N ranks itself (before any GPU call), rendezvous, take the MAX
over ranks and let rank 0 print ONE line with n_gpus = N; a WORLD_SIZE that disagrees with --gpus is an error.  Uses
bench.py's CPU rehearsal mode (gloo, no GPU work), so this is the spawn / reporting path and nothing else."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_bare_invocation_spawns_one_rank_per_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "cpu", "--steps", "3",
                        "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["ranks"] == {"world_size": 2, "backend": "gloo", "ranks_in_group": 2, "launched_by": "bench.py spawn"}
    assert "REHEARSAL" in line["data"]


def test_world_size_disagreeing_with_gpus_is_an_error():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse", "cpu"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_launcher_reports_a_failing_rank():
    from underwater_image_restoration_amd.shard import launch_local_ranks
    script = os.path.join(ROOT, "tests", "_failing_rank.py")
    with open(script, "w") as f:
        f.write("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\nprint('rank0 ok', flush=True)\ntime.sleep(0.5)\n")
    try:
        assert launch_local_ranks(script, [], 2) == 7
    finally:
        os.remove(script)
