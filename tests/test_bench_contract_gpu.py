"""bench.py's contract on a small grid: the single-GPU line (keys, roofline, parity, cpu_baseline) and the N = 2
launch the driver uses (`python -m torch.distributed.run ... bench.py --gpus 2`) rehearsed on ONE GPU with the gloo
transport -- two ranks share the card, everything except the RCCL wire is the code the 8-GPU run executes, and every
rank checks its band against the CPU oracle (`parity` in the JSON line, exit code 4 on failure)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "512", "--ny", "384", "--nz", "8", "--steps", "5",
                        "--warmup", "2", "--cpu-budget", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "secondary"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["kernel_ms"]["k_wind"] > 0 and rf["kernel_ms"]["k_thc"] > 0 and rf["kernel_ms"]["k_scan"] > 0 and rf["launches_per_call"] == 3
    assert d["parity"]["ok"] and max(d["parity"]["max_rel_err"].values()) < 1e-6
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    # the line says which runtime ran, which stream, which BASELINE configuration (none for this grid), and which state
    # of the contrast kernel's plan was timed -- with the replanned figures beside it
    c = d["config"]
    assert len(c["hip_runtime"]) == 1 and "torch" in c["stream"] and c["baseline_config_index"] is None
    assert "not a BASELINE.json configuration" in c["workload"] and c["plan_cache"] == "stored"
    rp = rf["replan"]
    assert rp["ms_per_step"] > 0 and rp["k_thc"] > 0 and rf["plan_cache"] == "stored"


def test_bench_refuses_the_other_import_order():
    """The library loaded BEFORE torch: two HIP runtimes in the process (torch loads its bundled copy by path), whose
    streams and synchronisation know nothing of each other.  bench.py must say so and stop -- not print a time that
    torch.cuda.synchronize() did not actually close."""
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "512", "--ny", "384", "--nz", "8", "--steps", "5",
            "--warmup", "2", "--no-cpu-baseline"]
    two = subprocess.run(args, capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, SEABREEZE_BENCH_LIBRARY_FIRST="1"))
    assert two.returncode != 0 and not [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert "HIP runtime" in two.stderr and "import torch before" in two.stderr, two.stderr[-1500:]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_bench_two_ranks_rehearsal(dtype):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "gloo",
                        "--nx", "512", "--ny", "384", "--nz", "8", "--steps", "4", "--warmup", "2", "--dtype", dtype],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "latband2" and d["config"]["comm"] == "torch-gloo"
    assert d["parity"]["ok"], d["parity"]
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
