"""`python bench.py --gpus N` without a launcher (the driver's command shape) must start the N ranks itself, before
any GPU call, relay rank 0's line and return the worst child code (reference layout: one process per GPU under DDP,
reflect_sampling_nerf_pipeline.py:72-77).  CPU only: the `selftest` workload runs the plumbing over gloo."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(extra_env=None, *argv, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH, *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus2_self_launches_two_gloo_ranks_and_prints_one_line():
    res = _run(None, "--gpus", "2", "--workload", "selftest")
    assert res.returncode == 0, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout  # exactly ONE JSON line on stdout (gloo's chatter goes to stderr)
    rec = json.loads(lines[0])
    assert rec == {"selftest": True, "n_gpus": 2, "rank_sum": 3.0}  # both ranks joined the all-reduce


def test_a_killed_rank_gives_a_nonzero_exit_code():
    # rank 1 dies by SIGKILL before the collective: rank 0's watchdog ends it with rc 3, the launcher reports the
    # worst code (128 + 9) and prints no result line
    res = _run({"RSN_BENCH_SELFTEST_FAIL_RANK": "1", "RSN_BENCH_WATCHDOG_S": "6"}, "--gpus", "2", "--workload", "selftest")
    assert res.returncode == 137, (res.returncode, res.stderr)
    assert not [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert "child exit codes" in res.stderr


def test_too_few_gpus_is_refused_before_any_child_starts():
    res = _run({"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert res.returncode == 2
    assert "GPU(s) visible" in res.stderr and not res.stdout.strip()


def test_self_launch_environment_and_worst_code(monkeypatch):
    """The spawn itself, mocked: N children of bench.py with the torchrun variables, same argv, rank 0 piped; the
    return value is the worst child code, a signal reads as 128 + k, survivors of a failed run are ended by PID."""
    sys.path.insert(0, REPO)
    import bench

    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, stderr=None):
            self.cmd, self.env, self.rank = cmd, env, int(env["RANK"])
            self.stdout = iter([b'{"ok": 1}\n']) if stdout == subprocess.PIPE else None
            self.terminated = False
            started.append(self)

        def poll(self):
            if self.rank == 0:
                return 4  # the exception path of rank 0
            return -15 if self.terminated else None  # rank 1 hangs in the collective until terminated

        def terminate(self):
            self.terminated = True

        def kill(self):
            self.terminated = True

        def wait(self):
            return self.poll()

    monkeypatch.delenv("MASTER_PORT", raising=False)
    rc = bench.self_launch(2, ["--gpus", "2", "--steps", "3"], popen=FakeProc, grace=0.3, poll=0.05)
    assert rc in (5, 143)  # worst of [4 (rank 0), the survivor ended by the launcher]
    assert rc != 0 and len(started) == 2
    for r, p in enumerate(started):
        assert p.cmd[0] == sys.executable and p.cmd[1] == os.path.abspath(bench.__file__)
        assert p.cmd[2:] == ["--gpus", "2", "--steps", "3"]
        assert p.env["RANK"] == str(r) and p.env["LOCAL_RANK"] == str(r) and p.env["WORLD_SIZE"] == "2"
        assert p.env["MASTER_ADDR"] == "127.0.0.1" and int(p.env["MASTER_PORT"]) > 0
        assert p.env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert started[0].env["MASTER_PORT"] == started[1].env["MASTER_PORT"]
    assert started[1].terminated  # the hung survivor was ended


def test_main_takes_the_self_launch_branch_without_world_size(monkeypatch):
    sys.path.insert(0, REPO)
    import bench

    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("RSN_BENCH_SHARE_GPU", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2"])
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: calls.append((n, argv)) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [(2, ["--gpus", "2", "--steps", "5", "--warmup", "2"])]
    # with WORLD_SIZE set (under torch.distributed.run) it is a rank and must NOT launch again
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--workload", "selftest"])
    monkeypatch.setattr(bench, "run_selftest", lambda rank, world: calls.append(("rank", rank, world)))
    bench.main()
    assert calls == [("rank", 0, 2)]


def _load_bench():
    sys.path.insert(0, REPO)
    import bench

    return bench


def test_json_line_reports_rccl_ranks_for_the_nccl_backend():
    """The driver checks `config.rccl_ranks == N` on the N > 1 lines: it must be the process group's size whenever the
    backend is RCCL ("nccl" on ROCm), and 0 for the gloo rehearsal / a run without a collective."""
    bench = _load_bench()

    class FakeDist:
        def __init__(self, n):
            self.n = n

        def get_world_size(self):
            return self.n

    for n in (2, 4, 8):
        cfg = bench.collective_config(FakeDist(n), "nccl", n)
        assert cfg["rccl_ranks"] == n and cfg["parallelism"] == "dp%d" % n and cfg["backend"] == "nccl"
        assert "all-reduce" in cfg["collective"]
    assert bench.collective_config(FakeDist(2), "gloo", 2)["rccl_ranks"] == 0
    assert bench.collective_config(None, None, 1) == {"parallelism": "dp1", "collective": None, "rccl_ranks": 0, "backend": None}


def test_launcher_counts_gpus_without_initialising_hip(monkeypatch):
    """The self-launching parent must not touch HIP before it forks the ranks: the GPU count comes from the
    visible-devices environment or the KFD topology in sysfs, never from torch.cuda / hipGetDeviceCount."""
    bench = _load_bench()
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count() == 3
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    n = bench.visible_gpu_count()  # this container has no /dev/kfd: None (the ranks then report a shortfall themselves)
    assert n is None or n >= 0
    import inspect

    src = inspect.getsource(bench.main)
    assert "torch.cuda.device_count" not in src.split("self_launch(args.gpus")[0]
