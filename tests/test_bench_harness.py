"""bench.py plumbing that must hold without a GPU: the self-launch for --gpus N > 1 starts torch.distributed.run as a
child (and does nothing else), and committed PMC numbers are only reported for the build they were measured on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_spawn_ranks_builds_a_torchrun_child(monkeypatch):
    seen = {}

    def fake_call(cmd):
        seen["cmd"] = cmd
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.spawn_ranks(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_main_self_launches_before_touching_the_gpu(monkeypatch):
    """world size 1 in the environment and --gpus 2 on the command line: spawn and exit with the child's code;
    torch.cuda.is_available() (which initialises the GPU) must not have been called."""
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: 0)

    def boom():
        raise AssertionError("GPU touched before the spawn")

    monkeypatch.setattr(bench.torch.cuda, "is_available", boom)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    else:
        raise AssertionError("main() should exit with the child's code")


def test_profile_counters_only_for_the_measured_build(monkeypatch, tmp_path):
    lib = bench._native.load().bdg_version().decode()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    kh = bench.kernels_hash(lib)
    assert kh != lib and kh in lib                       # "... kernels <hash> host <hash>": counters follow the device code only
    json.dump({"k_scan_reads": 123.0, "_meta": {"lib": "some other build", "kernels": "0123456789ab", "tag": "old"}}, open(prof / "traffic.json", "w"))
    json.dump({"k_scan_reads": 9.0, "_meta": {"lib": "a build with another host side", "kernels": kh, "tag": "new"}}, open(prof / "valu.json", "w"))
    pc = bench.profile_counters("k_scan_reads")
    assert pc["traffic"] is None and pc["valu"] == 9.0
    json.dump({"k_scan_reads": 123.0, "_meta": {"lib": lib, "kernels": kh, "tag": "new"}}, open(prof / "traffic.json", "w"))
    pc = bench.profile_counters("k_scan_reads")
    assert pc["traffic"] == 123.0 and "new" in pc["source"]


def test_clock_ramp_runs_whole_groups_of_steps_for_the_stated_time(monkeypatch):
    """bench.clock_ramp: untimed steps for a fixed wall time before the warm-up, synchronising between groups."""
    import time
    import bench
    calls = {"steps": 0, "syncs": 0}
    monkeypatch.setattr(bench.torch.cuda, "synchronize", lambda dev=None: calls.__setitem__("syncs", calls["syncs"] + 1))

    def step():
        calls["steps"] += 1
        time.sleep(0.001)
    t0 = time.perf_counter()
    k = bench.clock_ramp(step, None, seconds=0.05)
    dt = time.perf_counter() - t0
    assert k == calls["steps"] and k % 8 == 0 and k >= 8 and calls["syncs"] == k // 8
    assert 0.05 <= dt < 0.5


def test_multi_rank_rehearsal_over_gloo():
    """`python bench.py --gpus 2 --config 5 --rehearse` as the driver's SCALE leg starts it, minus the GPUs: the parent
    launches torch.distributed.run as a child, two ranks rendezvous on 127.0.0.1 over gloo, take part `rank` of 2 of the
    edge list (the library cuts the shares), time a stub step with the max-over-ranks clock, and rank 0 prints ONE line whose totals are sums
    over the ranks.  Same for the headline config (reads per GPU, weak scaling)."""
    import subprocess
    from badger_amd import dist as bdist
    env = dict(os.environ, BADGER_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "5", "--steps", "3", "--warmup", "1",
                        "--rows", "500000", "--rehearse"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["rehearsal"] is True and d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong"
    assert d["config"]["units_all_ranks"] == 500000 and d["config"]["part"] == 0 and d["config"]["nparts"] == 2 and d["config"]["threshold"] == 2
    assert d["ms_per_step"] >= 3.9                      # max over ranks: rank 1's stub sleeps 4 ms a step
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--reads", "1000", "--rehearse"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["units_all_ranks"] == 2000
