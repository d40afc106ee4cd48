"""CPU: bench.py's workload constants are the ones BASELINE.json / SURVEY.md section 8(d) fix."""
import importlib.util
import json
import os

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workload_is_config_2():
    b = _load_bench()
    assert (b.FFT, b.HOP, b.N_SAMPLES) == (1024, 256, 28_800_000)   # 10 min mono 48 kHz
    assert b.FRAMES == 112_497                                       # floor((n - N)/hop) + 1
    assert b.BYTES_PER_FRAME == 3072                                 # hop*4 + N/2*4
    assert b.HBM_PEAK_GBPS == 8000.0
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "1024-pt, hop 256" in base["metric"] and base["published"] == {}


def test_traffic_file_shape():
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return
    t = json.load(open(p))["power"]
    assert t["algorithmic_bytes_per_launch"] == 112_497 * 3072
    # PMC traffic may exceed the algorithmic bytes (halo re-reads) but not wildly
    assert 1.0 <= t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"] < 1.25
