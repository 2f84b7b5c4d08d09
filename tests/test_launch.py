"""`bench.py --gpus N` must run N ranks or refuse: the rank / port / environment arithmetic of
rtlamr_amd.launch, the spawner (with a trivial child instead of bench.py), and bench.py's own refusals -- all on CPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rtlamr_amd import launch  # noqa: E402


def test_single_rank_needs_no_launcher():
    assert launch.rank_plan(1, {}, 1) == ("rank", [])
    assert launch.rank_plan(1, {}, 8) == ("rank", [])
    with pytest.raises(launch.LaunchError, match="1 rank requested, 0 devices"):
        launch.rank_plan(1, {}, 0)


def test_spawn_plan_gives_every_rank_its_device_and_one_rendezvous():
    mode, envs = launch.rank_plan(8, {"HOME": "/root"}, 8, port=29999)
    assert mode == "spawn" and len(envs) == 8
    for r, e in enumerate(envs):
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["LOCAL_WORLD_SIZE"]) == (str(r), str(r), "8", "8")
        assert (e["MASTER_ADDR"], e["MASTER_PORT"]) == ("127.0.0.1", "29999")
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # dmabuf IPC: RCCL across processes needs it on this driver
    # fewer ranks than devices is fine (the driver runs N = 1, 2, 4, 8 on an 8-GPU node)
    mode, envs = launch.rank_plan(2, {}, 8)
    assert mode == "spawn" and [e["LOCAL_RANK"] for e in envs] == ["0", "1"]
    assert 1024 < int(envs[0]["MASTER_PORT"]) < 65536 and len({e["MASTER_PORT"] for e in envs}) == 1


def test_refusals_have_one_line_reasons():
    with pytest.raises(launch.LaunchError, match="2 ranks requested, 1 device$"):
        launch.rank_plan(2, {}, 1)
    with pytest.raises(launch.LaunchError, match="8 ranks requested, 4 devices"):
        launch.rank_plan(8, {}, 4)
    with pytest.raises(launch.LaunchError, match="--gpus 8 but the launcher started WORLD_SIZE=1"):
        launch.rank_plan(8, {"WORLD_SIZE": "1", "RANK": "0"}, 8)
    with pytest.raises(launch.LaunchError, match="--gpus 1 but the launcher started WORLD_SIZE=8"):
        launch.rank_plan(1, {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}, 8)
    with pytest.raises(launch.LaunchError, match="LOCAL_RANK=5 but 4 device"):
        launch.rank_plan(8, {"WORLD_SIZE": "8", "RANK": "5", "LOCAL_RANK": "5"}, 4)
    with pytest.raises(launch.LaunchError):
        launch.rank_plan(0, {}, 8)


def test_a_launchers_rank_is_taken_as_is():
    # torchrun's environment with a matching --gpus: the process is a rank, nothing is spawned
    env = {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500"}
    assert launch.rank_plan(4, env, 8) == ("rank", [])


def test_spawner_starts_every_rank_and_reports_the_first_failure(tmp_path):
    child = ("import os, sys, json; r = os.environ['RANK'];"
             f"open(os.path.join({str(tmp_path)!r}, 'rank' + r + '.json'), 'w').write(json.dumps(dict("
             "rank=r, local=os.environ['LOCAL_RANK'], world=os.environ['WORLD_SIZE'], port=os.environ['MASTER_PORT'])));"
             "print('only rank 0 prints to stdout' if r == '0' else 'rank ' + r);"
             "sys.exit(0)")
    mode, envs = launch.rank_plan(3, {}, 4)
    out = subprocess.run([sys.executable, "-c",
                          "import sys; sys.path.insert(0, %r); from rtlamr_amd import launch; import json;"
                          "sys.exit(launch.spawn_ranks([sys.executable, '-c', %r], json.loads(%r)))" % (ROOT, child, json.dumps(envs))],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "only rank 0 prints to stdout"            # the other ranks' stdout went to stderr
    assert "rank 1" in out.stderr and "rank 2" in out.stderr
    seen = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [s["local"] for s in seen] == ["0", "1", "2"] and {s["world"] for s in seen} == {"3"} and len({s["port"] for s in seen}) == 1
    # one rank fails: the job's exit code is non-zero and the rank that would wait forever is terminated
    bad = "import os, sys, time; sys.exit(3) if os.environ['RANK'] == '1' else time.sleep(60)"
    rc = launch.spawn_ranks([sys.executable, "-c", bad], envs)
    assert rc == 3          # the failing rank's own code: not 143 = the SIGTERM this function sent the others
    # a rank killed by a signal nobody here sent: 128 + signal
    killed = "import os, signal, time; os.kill(os.getpid(), signal.SIGKILL) if os.environ['RANK'] == '2' else time.sleep(60)"
    assert launch.spawn_ranks([sys.executable, "-c", killed], envs) == 128 + 9
    # two ranks fail with different codes: the first one seen failing decides
    two = "import os, sys, time; r = os.environ['RANK']; time.sleep(0 if r == '0' else 1.5); sys.exit(6 if r == '0' else 7 if r == '1' else 0)"
    assert launch.spawn_ranks([sys.executable, "-c", two], envs) == 6


def test_bench_refuses_instead_of_running_one_rank():
    # no device in the CPU container: every request for ranks is refused with the reason on stderr, nothing is printed
    # on stdout (no JSON line that could be mistaken for a measurement)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    if "0 devices" not in r.stderr and "1 device" not in r.stderr:
        pytest.skip("this host has two or more gfx950 devices")
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "2 ranks requested" in r.stderr
