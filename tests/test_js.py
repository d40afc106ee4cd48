"""The reference-language (JavaScript) side: the JS oracle mirror against the golden vectors (CPU),
and the N-API shim + AnalyserNode-shaped class against that oracle (GPU)."""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT, golden

NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed on this machine")
DRIVER = os.path.join(ROOT, "tests", "js_check.js")


@needs_node
@pytest.mark.parametrize("name", ["noise_n1024", "sweep_n256_h128", "noise_n4096"])
def test_js_oracle_matches_golden(name):
    g = golden(name)
    with tempfile.TemporaryDirectory() as d:
        pin, pout = os.path.join(d, "pcm.f32"), os.path.join(d, "out.f64")
        g["pcm"].astype(np.float32).tofile(pin)
        subprocess.check_call([NODE, DRIVER, "oracle", pin, str(int(g["fft_size"])), str(int(g["hop"])), pout])
        p = np.fromfile(pout, np.float64).reshape(g["power"].shape)
    assert np.abs(p - g["power"]).max() <= 1e-11 * g["power"].max()


@needs_node
def test_napi_addon_builds_and_fails_loudly_without_gpu():
    import torch
    from spectrogram_amd.js import build_napi
    path = build_napi.build()
    assert path and os.path.exists(path)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([NODE, "-e", "const {Analyser}=require('%s'); new Analyser();" %
                        os.path.join(ROOT, "spectrogram_amd", "js", "analyser.js")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "hipGetDeviceCount" in r.stderr  # no JS/CPU fallback


@needs_node
@pytest.mark.gpu
def test_js_analyser_on_gpu():
    from spectrogram_amd.js import build_napi
    assert build_napi.build()
    rng = np.random.default_rng(3)
    x = (rng.uniform(-1, 1, 1024 + 40 * 256) * 0.7).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        pin = os.path.join(d, "pcm.f32")
        x.tofile(pin)
        out = subprocess.check_output([NODE, DRIVER, "gpu", pin, "1024", "256"], text=True)
    res = json.loads(out.strip().splitlines()[-1])
    assert res["batch_len_ok"] and res["batch_worst_rel"] <= 1e-4, res
    assert res["bin_count"] == 1024 and res["ret_undefined"]
    assert res["bytes_outside_tolerance_interval"] == 0 and res["byte_intervals_wider_than_one"] < 2e-2, res
    assert res["short_ok"] and res["long_tail_untouched"] and res["time_domain_ok"] and res["time_domain_f32_ok"], res
    assert res["type_error"] == "TypeError" and res["range_error"] == "RangeError" and res["connect_ok"]
    assert res["replan_bins"] == 256 and res["replan_polls"] and res["closed_throws"], res
    assert res["ring_offset"] == 1 and res["ring_ok"] and res["ring_pinned"], res
    assert res["row_length"] == 64 and res["downmix_cols_equal"] and res["lf_ring_len"] == 2 * 64, res
    assert res["history_kept"] and res["ragged_push"] == "RangeError", res
    assert res["fused_frame_equals_two_calls"] and res["fused_type_error"] == "TypeError", res
    assert res["vertex_cols_worst"] <= 0.5 + 1e-3 < res["centre_cols_worst"] and res["centre_option_worst"] <= 0.5 + 1e-3, res


@needs_node
@pytest.mark.parametrize("n,hop,window", [(256, 128, "hann"), (1024, 256, "hann"), (2048, 735, "blackman"), (8192, 4101, "hann")])
def test_c_twin_is_the_js_float32_fft_bit_for_bit(oracle, n, hop, window):
    """so_stft_f32js (the yardstick of the stated tolerance) restates oracle/stft_oracle.js on Float32Array: the two
    must agree in every bit, so that "the float32 radix-2 JS FFT" of north_star can be evaluated from C on the GPU box."""
    rng = np.random.default_rng(n + hop)
    x = rng.uniform(-1, 1, n + 24 * hop).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        pin, pout = os.path.join(d, "pcm.f32"), os.path.join(d, "out.f32")
        x.tofile(pin)
        subprocess.check_call([NODE, DRIVER, "oracle32", pin, str(n), str(hop), pout, window])
        js = np.fromfile(pout, np.float32).reshape(25, n // 2)
    c = oracle.stft_f32js(oracle.OracleConfig.make(n, hop, window=1 if window == "blackman" else 0), x)
    assert np.array_equal(js.view(np.uint32), c.view(np.uint32))


@needs_node
@pytest.mark.gpu
def test_js_draw_loop_example_draws_the_python_twins_picture():
    """examples/draw_loop.js: the reference's per-frame loop in its own language (one pushAndGetByteFrequencyData per animation
    frame, rows kept in the device-side history texture, PGM out).  The picture must be, byte for byte, what the Python twin of
    the same object produces from the same samples, and the last frame's row the float64 oracle's columns within the byte bounds."""
    from spectrogram_amd.js import build_napi
    import spectrogram_amd as S
    assert build_napi.build()
    rng = np.random.default_rng(11)
    t = np.arange(736 * 300) / 44100.0
    x = (0.4 * np.sin(2 * np.pi * (200 + 3000 * t) * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        pin, pout = os.path.join(d, "pcm.f32"), os.path.join(d, "out.pgm")
        x.tofile(pin)
        out = subprocess.check_output([NODE, os.path.join(ROOT, "examples", "draw_loop.js"), pin, pout], text=True)
        res = json.loads(out.strip().splitlines()[-1])
        raw = open(pout, "rb").read()
    assert res["frames"] == 300 and res["history_offset"] == 300 % 256
    head = b"P5\n256 256\n255\n"
    assert raw.startswith(head) and len(raw) == len(head) + 256 * 256
    img = np.frombuffer(raw[len(head):], np.uint8).reshape(256, 256)
    a = S.Analyser(fftSize=2048, logfreq_cols=256)
    a.smoothingTimeConstant = 0
    a.startHistory(256)
    row = np.zeros(256, np.uint8)
    for f in range(300):
        a.pushAndGetByteFrequencyData(x[f * 736:(f + 1) * 736], row)
    twin = a.readHistory()[0]
    a.close()
    assert np.array_equal(img, twin)
    assert int(row.sum()) == res["last_row_sum"] and img.max() > 100     # a chirp well above the noise floor


@needs_node
@pytest.mark.gpu
def test_js_argument_validation_against_the_real_addon():
    """tests/js_shim_args.js -- the driver `make -C oracle check-shim-asan` runs under the sanitizers against the host stub --
    against the real addon and the real device: the same ~1,200 checks of types, lengths, ragged pushes, fftSize changes, the
    history ring and use after close() must hold there too, and the newest pushed sample must really come back."""
    from spectrogram_amd.js import build_napi
    assert build_napi.build()
    env = dict(os.environ)
    env.pop("SPECTRO_NODE", None)
    out = subprocess.check_output([NODE, os.path.join(ROOT, "tests", "js_shim_args.js")], text=True, env=env, timeout=600)
    res = json.loads(out.strip().splitlines()[-1])
    assert res["ok"] and res["checks"] > 1000
