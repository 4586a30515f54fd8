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


def _write(name, body):
    script = os.path.join(ROOT, "tests", name)
    with open(script, "w") as f:
        f.write(body)
    return script


def test_launcher_does_not_hang_when_a_peer_dies_under_a_blocked_rank0():
    """Rank 1 exits 7 while rank 0 blocks for ever (as it would inside a rendezvous or an RCCL barrier): the launcher must
    terminate rank 0 by PID and return 7 within seconds, not wait for rank 0's stdout to close."""
    import time
    from underwater_image_restoration_amd.shard import launch_local_ranks
    script = _write("_blocked_rank0.py", "import os, sys, time\nif os.environ['RANK'] == '1':\n    time.sleep(0.3)\n    sys.exit(7)\n"
                    "print('rank0 waiting', flush=True)\ntime.sleep(3600)\n")
    try:
        t0 = time.monotonic()
        assert launch_local_ranks(script, [], 2) == 7
        assert time.monotonic() - t0 < 20
    finally:
        os.remove(script)


def test_launcher_deadline_kills_every_rank():
    import time
    from underwater_image_restoration_amd.shard import launch_local_ranks
    script = _write("_all_blocked.py", "import signal, time\nsignal.signal(signal.SIGTERM, signal.SIG_IGN)\ntime.sleep(3600)\n")
    try:
        t0 = time.monotonic()
        assert launch_local_ranks(script, [], 2, deadline_s=1.0) == 124      # SIGTERM ignored -> SIGKILL after 5 s
        assert time.monotonic() - t0 < 30
    finally:
        os.remove(script)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_strong_scaling_leg_on_two_ranks_sharing_the_gpu():
    """configs[4]'s strong-scaling leg on more than one rank: `bench.py --gpus 2 --rehearse gpu --workload dataset4096`
    started bare spawns two ranks (both on cuda:0, rendezvous over gloo), splits the image set with shard.shard_range,
    all-reduces the metrics (train_student.py:646 counterpart) and lets rank 0 print one line."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "gpu", "--workload",
                        "dataset4096", "--images", "8", "--dataset-size", "64", "--warmup", "1", "--gather"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["ranks"]["ranks_in_group"] == 2 and line["ranks"]["backend"] == "gloo"
    assert line["config"]["images_per_rank"] == [4, 4] and line["config"]["images_total"] == 8
    assert line["config"]["gathered_on_rank0"] == [8, 3, 64, 64]
    m = line["dataset_metrics"]
    assert m["images"] == 8 and 0.0 < m["mean_psnr_vs_input_db"] <= 100.0 and -1.0 <= m["mean_ssim_vs_input"] <= 1.0
    assert line["value"] > 0
