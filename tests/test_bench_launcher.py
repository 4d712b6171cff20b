"""bench.py --gpus N really starts N ranks (the reference's fan-out of independent runs:
examples/nnet_barimages/SGEcluster/submit_multiM.sh:14-30, qsub_command.sh:8).

Runs here without a GPU: `--dry-run --backend gloo` makes every rank join the process group and count the
ranks with an all-reduce; nothing is timed and no device is touched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_gpus_2_starts_two_ranks():
    r = run(["--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = last_json(r.stdout)
    assert rec["n_gpus"] == 2 and rec["config"]["rccl_ranks"] == 2
    assert rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    # one line only: rank 1 prints nothing
    assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1


def test_single_rank_default():
    r = run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = last_json(r.stdout)
    assert rec["n_gpus"] == 1 and rec["config"]["rccl_ranks"] == 1


def test_world_size_must_match_gpus():
    # a rank of a 2-rank job that was told --gpus 4 refuses to run
    r = run(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
    # and a lone process with WORLD_SIZE unset and --gpus 1 must not spawn anything
    r = run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0


def test_env_switches_are_recorded():
    r = run(["--dry-run"], env={"VA_SOMETHING": "1", "VARANNEAL_AMD_LIB_X": "0", "UNRELATED": "x"})
    rec = last_json(r.stdout)
    assert rec["config"]["env"] == {"VARANNEAL_AMD_LIB_X": "0", "VA_SOMETHING": "1"}
