"""GPU suite: a short, seeded run of each randomised sweep under scripts/ (fuzz_*.py: entry points against fp64 evaluations on
the device; the partition and trainer sweeps against the global / single-process results).  The long runs are development
tools; a few seconds of each here keeps them working and catches what a fixed case list does not."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,extra", [
    ("fuzz_aggregation.py", []),
    ("fuzz_sgemm.py", []),
    ("fuzz_gat_layer.py", []),
    ("fuzz_layers.py", []),
    ("fuzz_rows_and_order.py", []),
    ("fuzz_partition.py", ["--world", "3"]),
    ("fuzz_partition.py", ["--world", "2", "--transport", "fake-rccl"]),
    ("fuzz_trainer.py", []),
])
def test_short_seeded_sweep(script, extra):
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / script), "--seconds", "6", "--seed", "1", *extra],
                       capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert '"failures": 0' in r.stdout or '"failures":0' in r.stdout, r.stdout[-2000:]
