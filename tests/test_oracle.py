"""CPU: pins the float64 oracle against the committed golden vectors (numpy/scipy-made,
oracle/make_golden.py) and against analytic known-answer tests.  The reference holds no
vectors of its own for this path -- parity is unpinned and these are
the substitute."""
import numpy as np
import pytest

from conftest import golden, golden_names

O_REL = 1e-11  # float64 vs float64


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", golden_names("noise_") + golden_names("hop_") +
                         ["sweep_n256_h128", "two_tone_n1024"])
def test_power_matches_golden(oracle, name):
    g = golden(name)
    cfg = oracle.OracleConfig.make(int(g["fft_size"]), int(g["hop"]))
    P, _ = oracle.stft(cfg, g["pcm"])
    assert P.shape == g["power"].shape
    assert rel(P, g["power"]) < O_REL
    if "nyquist" in g.files:
        cfg.nyquist = 1
        Pn, _ = oracle.stft(cfg, g["pcm"])
        assert Pn.shape[1] == int(g["fft_size"]) // 2 + 1
        assert rel(Pn[:, -1], g["nyquist"]) < O_REL
        assert rel(Pn[:, :-1], g["power"]) < O_REL


def test_config1_shape(oracle):
    g = golden("sweep_n256_h128")  # BASELINE.json configs[0]
    assert g["power"].shape == (374, 128)
    assert oracle.num_frames(48000, 256, 128) == 374


def test_degenerate(oracle):
    g = golden("degenerate_n1024")
    cfg = oracle.OracleConfig.make(1024, 256)
    for key in ("zeros", "square", "denorm"):
        P, _ = oracle.stft(cfg, g[key])
        ref = g["power_" + key]
        assert np.abs(P - ref).max() <= O_REL * max(ref.max(), 1e-300)
    cfg.out_kind = oracle.oracle.OUT_DB
    L, _ = oracle.stft(cfg, g["zeros"])
    assert np.all(L == -200.0)  # log floor


def test_stereo_stride(oracle):
    g = golden("stereo_n2048")
    cfg = oracle.OracleConfig.make(2048, 512)
    x = g["pcm_interleaved"]
    Pl, _ = oracle.stft(cfg, x, sample_stride=2)
    Pr, _ = oracle.stft(cfg, x[1:], sample_stride=2)
    assert rel(Pl, g["power_l"]) < O_REL and rel(Pr, g["power_r"]) < O_REL


def test_analyser_bytes(oracle):
    g = golden("analyser_n2048")
    n, hop, x = int(g["fft_size"]), int(g["hop"]), g["pcm"]
    # batch path: Blackman, 1/N, dB -> byte
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=3)
    Y, b = oracle.stft(cfg, x)
    assert np.abs(Y - np.maximum(g["db"], -200.0)).max() < 1e-9
    assert np.abs(b.astype(int) - g["bytes_tau0"].astype(int)).max() <= 1  # floor() at a boundary
    assert (b != g["bytes_tau0"]).mean() < 1e-3
    # poll path with smoothing 0.75 (3D/visualizer.js:351)
    st = np.zeros(n // 2)
    for f in range(g["bytes_tau075"].shape[0]):
        _, by = oracle.analyser_poll(n, x[f * hop: f * hop + n], 0.75, -100.0, -30.0, st)
        assert np.abs(by.astype(int) - g["bytes_tau075"][f].astype(int)).max() <= 1
    # tau = 0 poll == batch byte row
    st[:] = 0
    _, by = oracle.analyser_poll(n, x[:n], 0.0, -100.0, -30.0, st)
    assert np.abs(by.astype(int) - g["bytes_tau0"][0].astype(int)).max() <= 1


def test_logfreq(oracle):
    g = golden("logfreq_1024_to_256")
    out = oracle.logfreq_resample(g["row"], int(g["cols"]))
    assert np.abs(out - g["out"]).max() < 1e-9


def test_time_domain_bytes(oracle):
    x = np.array([-2.0, -1.0, -0.5, 0.0, 0.5, 0.999, 1.0, 3.0], np.float32)
    assert list(oracle.time_domain_bytes(x)) == [0, 0, 64, 128, 192, 255, 255, 255]


# ---- analytic known-answer tests (no dependency on numpy.fft) ---------------------------

@pytest.mark.parametrize("n", [32, 256, 1024, 4096])
def test_kat_impulse(oracle, n):
    # unit impulse at n0 -> |X[k]| = w[n0] for every k
    n0 = n // 3
    x = np.zeros(n, np.float32)
    x[n0] = 1.0
    cfg = oracle.OracleConfig.make(n, n, out_kind=1, nyquist=1)
    mag, _ = oracle.stft(cfg, x)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n0 / n)
    assert np.abs(mag - w).max() < 1e-12


@pytest.mark.parametrize("n", [32, 1024])
def test_kat_dc_and_cosine(oracle, n):
    cfg = oracle.OracleConfig.make(n, n, out_kind=1, nyquist=1)
    mag, _ = oracle.stft(cfg, np.ones(n, np.float32))
    exp = np.zeros(n // 2 + 1)
    exp[0], exp[1] = n / 2, n / 4  # Hann DC: X[0]=N/2, X[1]=-N/4
    assert np.abs(mag[0] - exp).max() < 1e-9 * n
    k0 = n // 8
    x = np.cos(2 * np.pi * k0 * np.arange(n) / n).astype(np.float64)
    mag, _ = oracle.stft(cfg, x.astype(np.float32))
    exp = np.zeros(n // 2 + 1)
    exp[k0], exp[k0 - 1], exp[k0 + 1] = n / 4, n / 8, n / 8
    assert np.abs(mag[0] - exp).max() < 2e-7 * n  # input was rounded to float32


def test_kat_parseval_linearity(oracle):
    n = 512
    rng = np.random.default_rng(1)
    a = rng.standard_normal(n).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    cfg = oracle.OracleConfig.make(n, n, window=2, nyquist=1)  # rectangular
    P, _ = oracle.stft(cfg, a)
    two_sided = 2 * P[0].sum() - P[0, 0] - P[0, -1]
    assert abs(two_sided - n * (a.astype(np.float64) ** 2).sum()) < 1e-9 * two_sided
    za = oracle.fft_radix2(a.astype(np.float64))
    zb = oracle.fft_radix2(b.astype(np.float64))
    zab = oracle.fft_radix2(2.0 * a.astype(np.float64) - 3.0 * b.astype(np.float64))
    assert np.abs(zab - (2 * za - 3 * zb)).max() < 1e-10
    # time shift by s multiplies bin k by e^{-2 pi i k s / n}
    s = 5
    zs = oracle.fft_radix2(np.roll(a.astype(np.float64), s))
    assert np.abs(zs - za * np.exp(-2j * np.pi * np.arange(n) * s / n)).max() < 1e-10


def test_ragged_and_errors(oracle):
    cfg = oracle.OracleConfig.make(1024, 256)
    P, _ = oracle.stft(cfg, np.zeros(1023, np.float32))
    assert P.shape == (0, 512)
    assert oracle.num_frames(1024, 1024, 256) == 1 and oracle.num_frames(1279, 1024, 256) == 1
    assert oracle.num_frames(1280, 1024, 256) == 2
    with pytest.raises(ValueError):
        oracle.stft(oracle.OracleConfig.make(1000, 256), np.zeros(4096, np.float32))
    with pytest.raises(ValueError):
        oracle.fft_radix2(np.zeros(12))


def test_f32_baseline_twin_agrees(oracle):
    # the float32 CPU-baseline variant computes the same thing (it is only ever timed)
    g = golden("noise_n1024")
    cfg = oracle.OracleConfig.make(1024, 256)
    P = oracle.stft_f32(cfg, g["pcm"])
    assert np.abs(P - g["power"]).max() / g["power"].max() < 1e-5


def test_byte_intervals_of_the_power_tolerance(oracle):
    """The criterion the GPU byte tests use (conftest.byte_bounds): the power tolerance mapped through the monotone byte
    map.  Checked here on the CPU: the float64 oracle's byte is always inside its interval; an all-float32 evaluation
    (the radix-2 twin's power, float32 log2 / FMA like the kernel's epilogue) is inside too; the interval is a single
    byte for all but a fraction of a per cent; and a byte off by one where the interval is single is caught."""
    from conftest import assert_bytes_in_bounds, byte_bounds, power_error, REL
    rng = np.random.default_rng(11)
    n, hop = 2048, 735
    x = (rng.uniform(-1, 1, n + 40 * hop) * np.linspace(1e-4, 1.0, n + 40 * hop)).astype(np.float32)
    P, _ = oracle.stft(oracle.OracleConfig.make(n, hop, window=1, scale=1), x)
    _, ref_u8 = oracle.stft(oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=3), x)
    lo, hi = byte_bounds(P, floor_db=-200.0)
    wide = assert_bytes_in_bounds(ref_u8, lo, hi, "oracle bytes")
    assert wide < 1e-2
    p32 = oracle.stft_f32(oracle.OracleConfig.make(n, hop, window=1, scale=1), x)
    assert power_error(p32, P) <= REL
    with np.errstate(divide="ignore"):
        l2 = np.log2(p32.astype(np.float64)).astype(np.float32)
    a, b = np.float32(3.01029995663981195 * 255.0 / 70.0), np.float32(100.0 * 255.0 / 70.0)
    v = (l2.astype(np.float64) * a + b).astype(np.float32)          # one FMA, rounded to float32
    b32 = np.clip(np.floor(np.where(np.isnan(v), 0, v)), 0, 255)
    assert_bytes_in_bounds(b32, lo, hi, "float32 twin bytes", ref_u8)
    single = np.argwhere((hi == lo) & (lo < 255))[0]
    wrong = ref_u8.copy()
    wrong[tuple(single)] += 1
    with pytest.raises(AssertionError):
        assert_bytes_in_bounds(wrong, lo, hi, "off by one inside a single-valued interval")


def test_the_float32_radix2_twin_itself_crosses_the_old_floor(oracle):
    """Why the stated tolerance floors the denominator at 2e-5 of the frame's peak and not at 1e-5:
    the all-float32 radix-2 FFT -- the kind of reference north_star names -- itself exceeds 1e-4 at a 1e-5 floor on this
    frame (found by oracle/find_twin_crossing.py after 1.06e10 bins: int16 noise, Blackman, N = 256; bin 45 sits 50.7 dB
    below the frame's peak), and is comfortably inside at 2e-5.  The JS-semantics twin (double butterflies) is not
    exact either.  A statement the reference's own arithmetic cannot meet is not a usable criterion."""
    g = golden("twin_crosses_old_floor_n256")
    x = g["pcm_s16"].astype(np.float32) / np.float32(32768.0)
    n = int(g["fft_size"])
    cfg = oracle.OracleConfig.make(n, n, window=1)
    ref, _ = oracle.stft(cfg, x)
    mx = ref.max()
    e32 = np.abs(oracle.stft_f32(cfg, x).astype(np.float64) - ref)
    ejs = np.abs(oracle.stft_f32js(cfg, x).astype(np.float64) - ref)
    assert (e32 / np.maximum(ref, 1e-5 * mx)).max() > 1.0e-4          # 1.026e-4 at bin 45
    assert int(np.argmax(e32 / np.maximum(ref, 1e-5 * mx))) == int(g["bin"])
    assert (e32 / np.maximum(ref, 2e-5 * mx)).max() < 0.6e-4          # the stated floor: 0.51e-4
    assert 0.2e-4 < (ejs / np.maximum(ref, 1e-5 * mx)).max() < 1.0e-4


def test_the_js_semantics_twin_crosses_the_old_floor_too(oracle):
    """... and so does oracle/stft_oracle.js on Float32Array itself (the restatement below is bit-equal to it under node,
    tests/test_js.py): 1.0906e-4 at the DC bin of this frame, 50.0 dB below the frame's peak, found after 2.8e10 bins."""
    g = golden("twin_js_crosses_old_floor_n256")
    x = g["pcm_s16"].astype(np.float32) / np.float32(32768.0)
    n = int(g["fft_size"])
    cfg = oracle.OracleConfig.make(n, n, window=1)
    ref, _ = oracle.stft(cfg, x)
    ejs = np.abs(oracle.stft_f32js(cfg, x).astype(np.float64) - ref)
    assert (ejs / np.maximum(ref, 1e-5 * ref.max())).max() > 1.0e-4
    assert (ejs / np.maximum(ref, 2e-5 * ref.max())).max() < 0.6e-4
