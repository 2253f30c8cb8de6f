"""bench.py as its own launcher: `python bench.py --gpus N` must start N fresh rank processes BEFORE the parent has
made any GPU call (it may not even import torch or load libmcx), hand them the torchrun-style environment and relay
rank 0's JSON line. Runs on CPU: --rehearse-cpu replaces the GPU work by one gloo all-reduce."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_parent_spawns_ranks_without_touching_torch_or_libmcx():
    code = r"""
import json, sys
sys.path.insert(0, %r)
import bench

seen = []
class FakeProc:
    def __init__(self, cmd, env=None, stdout=None, text=None):
        # what the parent process looks like at the moment it starts a rank
        assert "torch" not in sys.modules and "torch.cuda" not in sys.modules, "parent imported torch before spawning"
        assert "wgpu_montecarlo" not in sys.modules and "wgpu_montecarlo.runtime" not in sys.modules
        seen.append(dict(cmd=cmd, rank=env["RANK"], local=env["LOCAL_RANK"], world=env["WORLD_SIZE"],
                         addr=env["MASTER_ADDR"], port=env["MASTER_PORT"]))
        self.returncode = 0
        import io
        self.stdout = io.StringIO('{"ok": true}\n') if stdout is not None else None
    def poll(self): return 0
    def wait(self, timeout=None): return 0
    def kill(self): pass

argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--no-cold"]
rc = bench.spawn_ranks(bench.parse_args(argv), argv, popen=FakeProc)
assert rc == 0
assert [s["rank"] for s in seen] == ["0", "1", "2", "3"] and [s["local"] for s in seen] == ["0", "1", "2", "3"]
assert all(s["world"] == "4" and s["addr"] == "127.0.0.1" and s["port"] == seen[0]["port"] for s in seen)
assert all(s["cmd"][0] == sys.executable and s["cmd"][1].endswith("bench.py") and s["cmd"][2:] == argv for s in seen)
assert "torch" not in sys.modules
print("LAUNCHER-OK")
""" % str(ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "LAUNCHER-OK" in res.stdout, res.stdout + res.stderr


def test_two_rank_rehearsal_end_to_end():
    """The real thing minus the GPU: parent -> 2 children -> gloo rendezvous on 127.0.0.1 -> all-reduce -> one JSON line."""
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--rehearse-cpu"], capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["rehearsal"] and line["n_gpus"] == 2 and line["ranks_sum"] == line["expected"] == 3.0


def test_a_failing_rank_takes_the_job_down_quickly():
    """--gpus 2 with WORLD_SIZE mismatch inside the children is impossible to provoke from outside; instead start the
    launcher with an invalid backend: both ranks fail at init_process_group and the parent returns non-zero."""
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--no-cold", "--backend", "no-such-backend",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert res.returncode != 0


def test_driver_style_launch_is_still_accepted():
    """Under `python -m torch.distributed.run` WORLD_SIZE is already set: bench.py is then one of the ranks."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2",
                          "--rehearse-cpu"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_sum"] == 3.0
