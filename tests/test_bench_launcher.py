"""bench.py --gpus N started as a plain `python bench.py --gpus N` (no torchrun around it) must launch one rank per
GPU itself: as a child process, before anything in the parent has touched the GPU, with the driver's own command
line.  CPU only: the command is built and shown, not run."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_launcher_command_matches_the_drivers():
    sys.path.insert(0, ROOT)
    import bench

    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "3", "--warmup", "1"], port=29777)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                   "--master-addr", "127.0.0.1", "--master-port", "29777", os.path.join(ROOT, "bench.py"),
                   "--gpus", "8", "--steps", "3", "--warmup", "1"]


def test_plain_invocation_with_gpus_above_one_spawns_the_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BTLBF_BENCH_LAUNCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    # under a launcher (WORLD_SIZE set) nothing is spawned: the rank count wins over --gpus
    assert "import torch" not in open(os.path.join(ROOT, "bench.py")).read().split("def main()")[0]
