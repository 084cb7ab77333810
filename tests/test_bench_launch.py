"""bench.py's launch forms (CPU): called plainly with --gpus N > 1 it must turn itself into the torch.distributed.run
command line the driver would have used - as a child process, before anything touches the GPU - instead of exiting with
a usage message; under torch.distributed.run it must not hop again."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_8_builds_the_drivers_command(bench):
    cmd = bench.self_launch_command(["--gpus", "8", "--steps", "5", "--warmup", "2"], 8, port=29511)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]      # the caller's arguments, unchanged
    # a free port is picked when none is given
    auto = bench.self_launch_command([], 2)
    assert 1024 <= int(auto[auto.index("--master-port") + 1]) <= 65535


@pytest.mark.parametrize("argv", [["--gpus", "8"], ["--gpus", "2", "--steps", "1"], ["--gpus", "1", "--force-collective"]])
def test_plain_call_hops_before_any_gpu_use(bench, monkeypatch, argv):
    """No WORLD_SIZE in the environment: main() hands over to self_launch and exits with the child's code; it neither
    selects a device nor creates a process group first."""
    import torch
    seen = {}

    def fake_launch(args):
        seen["gpus"] = args.gpus
        return 37

    def forbidden(*a, **k):
        raise AssertionError("the GPU / process group was touched before the hop")

    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NERF_BENCH_SELF_LAUNCHED"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(bench, "self_launch", fake_launch)
    monkeypatch.setattr(torch.cuda, "set_device", forbidden)
    monkeypatch.setattr(torch.distributed, "init_process_group", forbidden)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 37 and seen["gpus"] == int(argv[1])


def test_no_second_hop_under_the_launcher(bench, monkeypatch):
    """With WORLD_SIZE set (torch.distributed.run started us) a mismatch is an error, never another launch."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(bench, "self_launch", lambda args: pytest.fail("hopped again"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_self_launch_relays_output_and_exit_code(bench, monkeypatch, capfd):
    """The hop itself, with a stand-in command: stdout is relayed line by line and the child's return code comes back."""
    import argparse
    monkeypatch.setattr(bench, "self_launch_command",
                        lambda argv, n, port=None: [sys.executable, "-c", "import sys; print('{\"metric\": 1}'); sys.exit(5)"])
    rc = bench.self_launch(argparse.Namespace(gpus=2))
    out = capfd.readouterr().out
    assert rc == 5 and '{"metric": 1}' in out


@pytest.mark.parametrize("world", ["8", "1"])
def test_launcher_form_sets_the_ipc_mode_before_any_gpu_call(bench, monkeypatch, world):
    """Under `python -m torch.distributed.run ... bench.py --gpus 8` (WORLD_SIZE / RANK / LOCAL_RANK in the environment,
    no self-launch hop) main() must have HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment when torch.cuda.set_device - the
    process's first GPU call, where HSA reads it - runs; a value chosen by the user is kept."""
    import torch

    class Reached(Exception):
        pass

    seen = {}

    def first_gpu_call(*a, **k):
        seen["value"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
        raise Reached()

    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setenv("WORLD_SIZE", world)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(bench, "self_launch", lambda args: pytest.fail("hopped under the launcher"))
    monkeypatch.setattr(torch.cuda, "set_device", first_gpu_call)
    monkeypatch.setattr(torch.distributed, "init_process_group",
                        lambda *a, **k: pytest.fail("process group before the device was selected"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", world])
    with pytest.raises(Reached):
        bench.main()
    assert seen["value"] == "0"
    # the user's own choice survives
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    with pytest.raises(Reached):
        bench.main()
    assert seen["value"] == "1"


def test_self_launch_child_environment_carries_it_too(bench, monkeypatch, capfd):
    import argparse
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setattr(bench, "self_launch_command", lambda argv, n, port=None: [
        sys.executable, "-c", "import os; print(os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), os.environ.get('NERF_BENCH_SELF_LAUNCHED'))"])
    assert bench.self_launch(argparse.Namespace(gpus=2)) == 0
    assert capfd.readouterr().out.split() == ["0", "1"]


def test_package_import_sets_it_for_library_users(monkeypatch):
    """sharded.ensure_ipc_env: what `import nerf_projects_amd` leaves in the environment (never overriding)."""
    sys.path.insert(0, ROOT)
    import nerf_projects_amd as N
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    assert N.ensure_ipc_env() == "0" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    assert N.ensure_ipc_env() == "1"


def test_whole_host_cpu_baseline_runs_worker_processes(bench):
    """cpu_baseline's second leg: P worker processes x T threads on disjoint ray slices, started
    together, GPU hidden from them; the record names P, T and the aggregate rate."""
    import numpy as np
    sys.path.insert(0, ROOT)
    g = np.load(os.path.join(ROOT, "tests", "golden", "render_rays_lego.npz"))
    out = bench.cpu_baseline_all_cores(g["rays"][:256], 8, 8, True, threads=1, ncpu=2, single_pool_rate=2e4, target_s=1.0)
    assert out["processes"] == 2 and out["threads_per_process"] == 1 and out["cores_all"] == 2, out
    assert out["value_all_cores"] and out["value_all_cores"] > 0, out
    # one process already covering the host: nothing to add
    one = bench.cpu_baseline_all_cores(g["rays"][:64], 8, 8, True, threads=8, ncpu=8, single_pool_rate=5.0, target_s=1.0)
    assert one["processes"] == 1 and one["value_all_cores"] == 5.0
