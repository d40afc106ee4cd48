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
    u8 = json.load(open(p)).get("db_u8")      # bench.py --out-kind db_u8 reads its roofline.traffic here (it was null)
    if u8:
        assert u8["algorithmic_bytes_per_launch"] == 112_497 * 1536
        assert 1.0 <= u8["hbm_bytes_per_launch"] / u8["algorithmic_bytes_per_launch"] < 1.25


def test_config5_constants():
    b = _load_bench()
    assert (b.UTT_SAMPLES, b.UTT_FRAMES) == (160_000, 622)            # 10 s at 16 kHz, N=1024, hop 256
    assert 1_000_000 * b.UTT_FRAMES == 622_000_000                     # 6.22e8 frames


def test_gpus_without_launcher_spawns_fresh_ranks(tmp_path, monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE must start N child processes (one per GPU, rendezvous on
    127.0.0.1) from a parent that never touched the GPU, instead of asserting.  The children are
    replaced here by a stub interpreter that records what it was given."""
    import subprocess
    import sys
    b = _load_bench()
    seen = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
            self.returncode = 0
            self._rank0 = env["RANK"] == "0"

        def communicate(self):
            return ('{"rank0": true}\n' if self._rank0 else ""), None

        def wait(self):
            return 0

    monkeypatch.setattr(b.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    ns = type("A", (), {"gpus": 4})()
    rc = b.spawn_ranks(ns, ["--gpus", "4", "--steps", "3"])
    assert rc == 0 and len(seen) == 4
    assert [e["RANK"] for _, e in seen] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for _, e in seen]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for _, e in seen)
    assert len({e["MASTER_PORT"] for _, e in seen}) == 1
    assert all(cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
               for cmd, _ in seen)
    # and the parent path is taken before torch is imported: main() with --gpus 2 and no WORLD_SIZE exits via spawn_ranks
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('sys.exit(spawn_ranks(args, sys.argv[1:]))') < src.index("    import torch\n    import torch.distributed as dist")


def test_timed_region_excludes_the_closing_collective():
    """the closing barrier (an RCCL all-reduce at N > 1) must not sit inside the interval `value` is
    computed from.  A barrier that takes 50 ms must leave the measured time of a 10 ms body at ~10 ms, the order of
    calls must be barrier, body, synchronize, (stamp), barrier, max-reduce, and one rank keeps its meaning."""
    import time
    b = _load_bench()
    calls = []

    class FakeCuda:
        @staticmethod
        def synchronize():
            calls.append("sync")

    fake_torch = type("T", (), {"cuda": FakeCuda})

    def barrier():
        calls.append("barrier")
        time.sleep(0.05)

    def body():
        calls.append("body")
        time.sleep(0.01)

    def max_over_ranks(dt):
        calls.append("max")
        return dt

    dt = b.timed_region(body, barrier, max_over_ranks, fake_torch)
    assert calls == ["barrier", "body", "sync", "barrier", "max"]
    assert 0.009 <= dt < 0.04, dt   # the 50 ms barriers on either side are not billed
    src = open(os.path.join(ROOT, "bench.py")).read()
    fn = src[src.index("def timed_region"):src.index("def spawn_ranks")]
    assert fn.index("t0 = time.perf_counter()") < fn.index("body()") < fn.index("torch.cuda.synchronize()") \
        < fn.index("dt = time.perf_counter() - t0") < fn.rindex("barrier()")
    # every timed interval in the file goes through it: no other perf_counter() pair brackets a barrier()
    assert src.count("timed_region(") >= 4


def test_stdout_line_is_compact_and_complete():
    """BENCH_r04.json: round 4's default line had grown to 37 KB and the driver recorded `"parsed": null`.  The stdout line
    is now a summary (compact_line) and the full object goes to a detail file: on a real full line of round 4 the summary
    stays under MAX_LINE_BYTES, carries every key of the contract, `roofline` (with traffic) and `cpu_baseline`, and one
    figure set per BASELINE.json config (configs[2] batched, configs[3] streaming, configs[4] config5)."""
    b = _load_bench()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_n1_driver_cmd.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 30000
    c = b.compact_line(full, "gpurun_out/bench_detail_config2_n1.json")
    text = json.dumps(c)
    assert len(text) <= b.MAX_LINE_BYTES <= 8000 and "\n" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["value"] == full["value"] and c["config"]["workload"] == full["config"]["workload"]   # the headline keeps every digit
    r = c["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] > 3.4e8
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and abs(r["achieved"] - full["roofline"]["achieved"]) < 1.0
    assert r["kernel"] == "sg::stft_kernel<10,2,0>" and r["avg_launch_ms"] > 0
    cb = c["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and len(cb["sample"]) <= 200
    assert c["batched"]["roofline"]["kernel"] == "sg::stft_kernel<11,2,0>" and c["batched"]["value"] > 1e8
    assert c["config5"]["value"] > 1e8 and c["config5"]["pcie_inclusive_value"] > 1e6
    s = c["streaming"]
    assert s["budget_ms"] > 21 and s["push_then_poll"]["host_wall"]["max"] > 0 and "one_launch" in s and "plain_c_host_wall" in s
    assert c["detail"].endswith(".json")
    # an object that would still overflow loses its optional parts, never the contract's
    fat = dict(full, config=dict(full["config"], sharding="x" * 7000))
    c2 = b.compact_line(fat)
    # whatever the contract's own objects hold, the line that goes out fits (free text is cut, as a last resort only
    # the contract's scalars remain) -- compact_line asserts it, and the contract keys survive every step of the fallback
    assert len(json.dumps(c2)) <= b.MAX_LINE_BYTES
    assert all(k in c2 for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"))
    assert c2["value"] == full["value"] and c2["roofline"]["frac"] > 0 and c2["cpu_baseline"]["kind"] == "port"
    fatter = dict(fat, roofline=dict(full["roofline"], note="n" * 9000), cpu_baseline=dict(full["cpu_baseline"], note="m" * 9000))
    c3 = b.compact_line(fatter)
    assert len(json.dumps(c3)) <= b.MAX_LINE_BYTES and c3["value"] == full["value"] and c3["roofline"]["bound"] == "hbm"
    # one optional object too many: only the smallest that makes the line fit is dropped
    pad = dict(full, host=dict(full.get("host", {}), cpu_quota_cores=16, torch_threads_used=16))
    pad["config"] = dict(full["config"], sharding="y" * (b.MAX_LINE_BYTES - len(json.dumps(c)) + 40))
    c4 = b.compact_line(pad)
    assert len(json.dumps(c4)) <= b.MAX_LINE_BYTES and "streaming" in c4 and "batched" in c4
    # a config-5 line keeps its PCIe-inclusive figure
    l5 = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_config5_n1.json")).read().strip().splitlines()[-1])
    c5 = b.compact_line(l5)
    assert c5["pcie_inclusive"]["value"] == float("%.5g" % l5["pcie_inclusive"]["value"]) and c5["roofline"]["frac"] > 0
