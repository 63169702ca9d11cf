"""bench.py's own launcher (no GPU needed): `python bench.py --gpus N` without WORLD_SIZE must re-execute itself under
torch.distributed.run with the same arguments; with WORLD_SIZE set it must not."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_builds_the_drivers_command(monkeypatch):
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.self_launch(8) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_pmc_summary_prefers_the_running_code_and_says_so():
    import bench
    w = "examples/npt-flange resdiv 1600: octree prune + marching cubes on device (res 0.0541987, 12 levels)"
    any_ = bench.pmc_summary(w, None)
    assert any_ and any_["code_match"] is True and any_["valu_lane_instr_per_eval"] > 0
    other = bench.pmc_summary(w, "not-a-code-key")
    assert other and other["code_match"] is False and other["source"] == any_["source"]
    assert "DIFFERS" in bench.counters_note(other["source"], other["code_match"])
    assert bench.pmc_summary("examples/no-such-scene resdiv 7", None) == {}
    va = bench.valu_roofline(5e11, w, "not-a-code-key")
    assert va["code_match"] is False and 0 < va["frac"] < 1
