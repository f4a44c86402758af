"""bench.py's one-line JSON contract (task statement + ④): run a short bench on the GPU and check the fields the driver and
the judge read.  Also the item-complete distribution (the shape FOCFDataLoader produces) and the eager-only mode."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "2", "--age", "8",
                          "--no-cpu-baseline", *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line
    return json.loads(lines[0])


SMALL = ("--users", "200001", "--items", "50001", "--nfcf-users", "200001", "--nfcf-items", "50001")   # the other_workloads tables


@pytest.mark.parametrize("args", [SMALL, ("--item-dist", "grouped"), ("--no-graph", "--no-workloads"), ("--launch", "graph", "--no-workloads"),
                                  ("--launch", "library", "--no-workloads"), ("--launch", "library", "--item-dist", "grouped")],
                         ids=["default", "grouped", "eager", "graph", "library", "library_grouped"])
def test_bench_json_contract(args):
    d = _bench(*args)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "interactions/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 12 * 8192 / (d["ms_per_step"] * 12e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert d["config"]["final_loss"] == d["config"]["final_loss"]      # not NaN
    if "--item-dist" not in args:
        # the item-complete shape FOCF's real loader produces is reported next to the figure of record,
        # under the unique-row bytes definition it names
        shapes = {x["item_distribution"]: x for x in d["other_batch_shapes"]}
        assert set(shapes) == {"grouped", "zipf"}
        g = shapes["grouped"]
        assert "unique rows" in g["bytes_definition"] and g["us_per_step"] > 0 and "fr_focf_step_runs" in g["step"]
        assert 60 < g["distinct_item_rows_per_batch"] < 100 and g["bytes_per_interaction"] < 3096
        assert abs(g["frac_of_hbm_peak"] - g["achieved_GBps"] / 8000.0) < 1e-3 and g["hipGraph_us_per_step"] > 0
    else:
        assert "other_batch_shapes" not in d
    # --launch auto (the default): 12 timed steps are one hipGraph replay (the library's step loop from 64 steps on), and the
    # line says so
    if "--no-graph" in args:
        mode = "eager"
    elif "--launch" in args:
        mode = {"graph": "hipGraph", "library": "library step loop"}[args[args.index("--launch") + 1]]
    else:
        mode = "hipGraph"
        assert "auto" in d["config"]["launch_note"]
    assert d["config"]["launch"].startswith(mode), d["config"]["launch"]
    if args is SMALL:
        # BASELINE.json configs[2..4] ride in the default line (here on small tables): ms per step, rate, roofline of each
        modes = d["config"]["launch_modes_timed"]
        assert {"library_loop_ms_per_step", "hipGraph_ms_per_step", "eager_ms_per_step"} <= set(modes)
        assert abs(modes["hipGraph_ms_per_step"] - d["ms_per_step"]) < 1e-4
        w = d["other_workloads"]
        assert set(w) == {"pfcn10m", "nfcf100m", "fairgo10m"}
        for name, x in w.items():
            assert "skipped_reason" not in x, (name, x)
            assert x["steps"] >= 5 and x["ms_per_step"] > 0 and x["interactions_per_s"] > 0, (name, x)
            assert x["roofline"]["bound"] in ("hbm", "mfma", "valu (exact replay)") and 0.0 < x["roofline"]["frac"] < 1.0, (name, x)
            assert abs(x["interactions_per_s"] - 8192 / (x["ms_per_step"] * 1e-3)) <= 2e-3 * x["interactions_per_s"], (name, x)
    else:
        assert "other_workloads" not in d


def test_bench_nfcf_workload_contract():
    """`--workload nfcf100m` (BASELINE.json configs[4]) at a reduced table size: same JSON contract, NFCF finetune step."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "nfcf100m", "--nfcf-users", "200001",
                          "--nfcf-items", "50001", "--steps", "6", "--warmup", "2"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["unit"] == "interactions/s" and d["value"] > 0
    assert "NFCF finetune" in d["config"]["workload"] and d["roofline"]["algorithmic_bytes_per_launch"] == 7192 * 8192
    assert d["config"]["final_loss"] == d["config"]["final_loss"]
