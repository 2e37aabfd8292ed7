"""CPU: `python bench.py --gpus N` starts N ranks as a child process before anything touches the GPU (BASELINE configs[3]: the
reference starts one process per GPU itself, /root/reference/train.py:296-313), and a rank whose WORLD_SIZE disagrees with
--gpus refuses instead of printing a record for the wrong N."""
import argparse
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Done:
    def __init__(self, rc):
        self.returncode = rc


def test_launcher_builds_one_rank_per_gpu(monkeypatch):
    bench = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    seen = {}

    def runner(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        return _Done(7)

    rc = bench.launch_ranks(argparse.Namespace(gpus=4), argv=["--gpus", "4", "--steps", "5", "--warmup", "2"], runner=runner)
    assert rc == 7                                   # the child's status is the launcher's status
    cmd = seen["cmd"]
    assert cmd[:4] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_single_gpu_and_ranks_do_not_launch(monkeypatch):
    bench = _bench()

    def runner(cmd, env):
        raise AssertionError("must not start a child")

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.launch_ranks(argparse.Namespace(gpus=1), argv=[], runner=runner) is None
    monkeypatch.setenv("WORLD_SIZE", "8")            # a rank started by torch.distributed.run
    assert bench.launch_ranks(argparse.Namespace(gpus=8), argv=[], runner=runner) is None
    monkeypatch.setenv("WORLD_SIZE", "2")            # --gpus and the launcher disagree: refuse loudly
    assert bench.launch_ranks(argparse.Namespace(gpus=8), argv=[], runner=runner) == 2
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert bench.launch_ranks(argparse.Namespace(gpus=1), argv=[], runner=runner) is None


def test_mismatch_exits_nonzero_before_any_gpu_call():
    """the real entry point, no GPU in this container: the refusal must come before torch.cuda.set_device"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2, p.stderr[-2000:]
    assert p.stdout.strip() == ""
    assert "WORLD_SIZE=2" in p.stderr


def test_failed_rank_ends_the_launcher_nonzero(tmp_path, monkeypatch):
    """A rank that dies must end the launch with a non-zero status instead of leaving the launcher (and the surviving ranks, blocked in a
    collective) hanging: the real `torch.distributed.run` child, started through launch_ranks' own command line with the script swapped
    for one whose rank 1 exits with status 5 while rank 0 would sleep for ten minutes."""
    import time
    bench = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(5)\n"
                      "time.sleep(600)\n")

    def runner(cmd, env):
        i = cmd.index(os.path.join(ROOT, "bench.py"))
        return subprocess.run(cmd[:i] + [str(script)], env=env, capture_output=True, timeout=240)

    t0 = time.time()
    rc = bench.launch_ranks(argparse.Namespace(gpus=2), argv=["--gpus", "2"], runner=runner)
    assert rc not in (0, None)
    assert time.time() - t0 < 200                    # (the agent tears the other rank down; it does not wait for the sleep)


def test_work_skipping_switches_are_refused():
    """with a work-skipping dev switch in the environment bench.py prints no `value` and exits non-zero - before any GPU
    call (this container has none) - and the package itself refuses to import unless MMD_DEV=1 opts in explicitly."""
    import json
    env = {k: v for k, v in os.environ.items() if not k.startswith("MMD_")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], env=dict(env, MMD_DEV="1", MMD_DEV_SKIP_CALLS="mmd_se_fc_bwd"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, p.stderr[-2000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["invalid"] is True and "value" not in rec and rec["switches"] == ["MMD_DEV_SKIP_CALLS"]
    assert "SKIPPED" in p.stderr                     # the import-time warning
    # without the explicit opt-in the switch is an error, not a silent no-op and not a silent skip
    p = subprocess.run([sys.executable, "-c", "import mm_distillnet_amd._lib"], env=dict(env, MMD_DEV_SKIP_WG="1", PYTHONPATH=ROOT),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "MMD_DEV=1" in p.stderr
    # train.py refuses too (checked on the function: starting train.py needs a GPU)
    from mm_distillnet_amd import _lib
    assert _lib.work_skipping_switches({"MMD_DEV_NO_BWD": "1", "MMD_NO_PACK": "1"}) == ["MMD_DEV_NO_BWD"]
    assert _lib.work_skipping_switches({"MMD_NO_PACK": "1"}) == []


def test_alt_leg_only_for_the_plain_full_record_command():
    """bench.py's second leg (the same command with MMD_MFMA_F32=1 in a child process, `alt_mfma_f32` in the record) must start only where
    starting a process is safe and the record is the full one: fp32, one process, CPU-baseline leg on, no profiler preloaded (rocprofv3's
    library initialises the GPU before the program starts - a process in that state must not start another program), not already on v_mfma_f32."""
    import argparse
    import bench
    def args(**kw):
        d = dict(precision="fp32", no_alt=False, no_cpu_baseline=False, gpus=1)
        d.update(kw)
        return argparse.Namespace(**d)
    guard = "/usr/local/graft/lib/libasan.so.libclang_rt.asan.graft-execguard.so"
    assert bench.alt_leg_wanted(args(), {"LD_PRELOAD": guard})
    assert bench.alt_leg_wanted(args(), {})
    assert not bench.alt_leg_wanted(args(no_alt=True), {})
    assert not bench.alt_leg_wanted(args(no_cpu_baseline=True), {})
    assert not bench.alt_leg_wanted(args(precision="bf16"), {})
    assert not bench.alt_leg_wanted(args(gpus=4), {})
    assert not bench.alt_leg_wanted(args(), {"WORLD_SIZE": "2"})
    assert not bench.alt_leg_wanted(args(), {"MMD_MFMA_F32": "1"})
    assert not bench.alt_leg_wanted(args(), {"LD_PRELOAD": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"})
    assert not bench.alt_leg_wanted(args(), {"ROCPROF_OUTPUT_PATH": "/tmp/x"})
    assert not bench.alt_leg_wanted(args(), {}, skipping=["MMD_DEV_SKIP_WG"])
