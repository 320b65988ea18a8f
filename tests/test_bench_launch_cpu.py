"""bench.py --gpus N without the torchrun environment launches the N ranks itself (a parent that never touches a GPU)
and relays rank 0's line: exercised here with --dry-launch over gloo (no GPU in this container)."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT


@pytest.mark.timeout(300)
def test_bench_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True,
                       text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_launch"] is True and d["backend"] == "gloo"


def test_bench_single_rank_dry_and_args():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.parse_args([])
    assert a.gpus == 1 and a.steps % 9 == 0 and a.mode == "render"
    assert len(bench.GAZES) == 9 and bench.GAZES[4] == (0.5, 0.5) and (0.25, 0.75) in bench.GAZES
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-launch"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1
