"""bench.py's watchdog (the guard around the N > 1 exchange probe): a deadline that passes makes rank 0 print the record it
already holds -- with the reason -- and end the process with status 0; a rank without a record ends with a failure status;
a disarmed or re-armed watchdog does nothing.  Runs without a GPU (the class only needs the standard library)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

PROLOGUE = "import sys, time, json; sys.path.insert(0, %r); import bench\n" % ROOT


def _run(body, timeout=30):
    return subprocess.run([sys.executable, "-c", PROLOGUE + body], capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_deadline_prints_the_fallback_record_and_exits_cleanly():
    r = _run("w = bench.Watchdog(0)\n"
             "w.fallback = lambda reason: {'value': 1.0, 'config': {'watchdog': reason}}\n"
             "w.arm('exchange candidate X', 1.0)\n"
             "time.sleep(20)\nprint('NOT REACHED')\n")
    assert r.returncode == 0 and "NOT REACHED" not in r.stdout
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["value"] == 1.0 and "exchange candidate X" in rec["config"]["watchdog"]
    assert "did not finish in time" in r.stderr


def test_rank_without_a_record_fails_and_other_ranks_leave_quietly():
    r0 = _run("w = bench.Watchdog(0)\nw.arm('first exchange', 1.0)\ntime.sleep(20)\n")
    assert r0.returncode == 3 and "{" not in r0.stdout
    r1 = _run("w = bench.Watchdog(5)\nw.fallback = lambda reason: {'never': 'printed by rank 5'}\nw.arm('first exchange', 1.0)\ntime.sleep(20)\n")
    assert r1.returncode == 0 and "never" not in r1.stdout
    # after the record is out (teardown) a missed deadline must not turn the run into a failure
    r2 = _run("w = bench.Watchdog(0)\nw.record_printed = True\nw.arm('teardown', 1.0)\ntime.sleep(20)\n")
    assert r2.returncode == 0


def test_disarm_and_rearm():
    r = _run("w = bench.Watchdog(0)\nw.fallback = lambda reason: {'x': 1}\n"
             "w.arm('a', 1.0)\nw.disarm()\ntime.sleep(2.5)\n"
             "w.arm('b', 60.0)\ntime.sleep(1.0)\nw.disarm()\nprint('reached')\n")
    assert r.returncode == 0 and r.stdout.strip() == "reached"


def test_self_launch_builds_the_driver_command_and_passes_the_status_through(tmp_path):
    """bench.self_launch (what `python bench.py --gpus N` does when no launcher set WORLD_SIZE): the command is the one the
    driver would have used -- torch.distributed.run, one node, N processes, 127.0.0.1 rendezvous, the same arguments -- and
    the launcher's exit status is returned.  No GPU: subprocess.call is replaced by a recorder."""
    r = _run("import subprocess\n"
             "seen = {}\n"
             "def fake(cmd, env=None):\n"
             "    seen['cmd'] = cmd; seen['legacy'] = env.get('HSA_ENABLE_IPC_MODE_LEGACY'); return 7\n"
             "subprocess.call = fake\n"
             "sys.argv = ['bench.py', '--gpus', '4', '--steps', '9', '--warmup', '3']\n"
             "rc = bench.self_launch(4)\n"
             "print(json.dumps({'rc': rc, 'cmd': seen['cmd'], 'legacy': seen['legacy']}))\n")
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["cmd"]
    assert d["rc"] == 7 and d["legacy"] == "0"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "9", "--warmup", "3"] and cmd[-7].endswith("bench.py")


def test_a_launcher_in_the_environment_is_respected():
    """With WORLD_SIZE set (the driver's torchrun command) bench.py must NOT start ranks of its own; a mismatch between
    --gpus and WORLD_SIZE is still an error."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       cwd=ROOT, env=dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr and "no launcher in the environment" not in r.stderr
