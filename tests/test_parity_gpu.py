"""GPU: the HIP path, called through the C ABI, against the golden vectors and the oracle."""
import numpy as np
import pytest

from conftest import (REL, ByteBounds, assert_bytes_in_bounds, assert_power_close, byte_bounds, golden, golden_names,
                      logfreq_byte_bounds, power_error)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import spectrogram_amd as S
    return S


def supported(n):
    return 32 <= n <= 32768


def poll_power(oracle, last_n):
    """float64 |X|^2 of the frame an AnalyserNode poll transforms (Blackman, 1/N) -- input of the byte intervals."""
    n = len(last_n)
    return oracle.stft(oracle.OracleConfig.make(n, n, window=1, scale=1), last_n)[0][0]


@pytest.mark.parametrize("name", golden_names("noise_") + golden_names("hop_") +
                         ["sweep_n256_h128", "two_tone_n1024"])
def test_power_vs_golden_host_path(S, name):
    g = golden(name)
    n, hop = int(g["fft_size"]), int(g["hop"])
    if not supported(n):
        with pytest.raises(S.SpectroError):
            S.Engine(fft_size=n, hop=hop)
        return
    with S.Engine(fft_size=n, hop=hop) as e:
        P = e.stft_host(g["pcm"])
    assert_power_close(P, g["power"], name)
    if "nyquist" in g.files:
        with S.Engine(fft_size=n, hop=hop, nyquist=True) as e:
            Pn = e.stft_host(g["pcm"])
        assert Pn.shape[1] == n // 2 + 1
        assert_power_close(Pn, np.concatenate([g["power"], g["nyquist"][:, None]], axis=1), name + "+nyq")


@pytest.mark.parametrize("n,hop", [(256, 128), (256, 64), (512, 128), (1024, 256), (1024, 512),
                                   (1024, 128), (2048, 512), (4096, 1024), (8192, 2048), (16384, 4096), (32768, 8192),
                                   (1024, 300), (64, 16), (32, 8)])
def test_device_path_vs_oracle_multichannel(S, oracle, n, hop):
    """torch-owned device buffers, several channels, ragged last run, sliding-window variants."""
    import torch
    rng = np.random.default_rng(n * 131 + hop)
    n_ch = 3
    n_samples = n + hop * 77 + 5  # 78 frames: not a multiple of any run length
    x = rng.uniform(-1, 1, (n_ch, n_samples)).astype(np.float32)
    with S.Engine(fft_size=n, hop=hop) as e:
        out = e.stft(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        got = out.cpu().numpy()
    cfg = oracle.OracleConfig.make(n, hop)
    for c in range(n_ch):
        ref, _ = oracle.stft(cfg, x[c])
        assert_power_close(got[c], ref, "n=%d hop=%d ch=%d" % (n, hop, c))


@pytest.mark.parametrize("out,okind", [("mag", 1), ("db", 2), ("db_u8", 3)])
@pytest.mark.parametrize("n,hop", [(1024, 256), (2048, 735), (4096, 1024), (256, 128)])
def test_output_kinds(S, oracle, out, okind, n, hop):
    rng = np.random.default_rng(7)
    x = (rng.uniform(-1, 1, n + 20 * hop) * np.linspace(1e-4, 1.0, n + 20 * hop)).astype(np.float32)
    kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out)
    with S.Engine(**kw) as e:
        got = e.stft_host(x)
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=okind)
    ref, ref_u8 = oracle.stft(cfg, x)
    if out == "mag":
        assert_power_close(got.astype(np.float64) ** 2, ref ** 2, out)
    elif out == "db":
        # a relative power error of 1e-4 is 4.3e-4 dB; bins far below the frame peak are
        # compared through the power statement instead
        assert_power_close(10.0 ** (got.astype(np.float64) / 10), 10.0 ** (ref / 10), out)
    else:
        # bytes: inside the interval of bytes the stated power tolerance maps to (conftest.py: equal to the float64
        # oracle's byte wherever its pre-floor value is further than that tolerance from an integer)
        assert got.dtype == np.uint8
        P, _ = oracle.stft(oracle.OracleConfig.make(n, hop, window=1, scale=1), x)
        wide = assert_bytes_in_bounds(got, *byte_bounds(P, floor_db=-200.0), "db_u8", ref_u8)
        assert wide < 1e-2, "the tolerance interval is a single byte for all but %.2f %% of the bytes" % (100 * wide)


def test_degenerate_inputs(S, oracle):
    g = golden("degenerate_n1024")
    with S.Engine(fft_size=1024, hop=256) as e:
        for key in ("zeros", "square", "denorm"):
            assert_power_close(e.stft_host(g[key]), g["power_" + key], key)
    with S.Engine(fft_size=1024, hop=256, out="db") as e:
        assert np.all(e.stft_host(g["zeros"]) == -200.0)
    with S.Engine(fft_size=1024, hop=256, out="db_u8") as e:
        assert np.all(e.stft_host(g["zeros"]) == 0)
        assert e.stft_host(g["square"] * 1e3).max() == 255  # clamp


def test_stereo_interleaved(S):
    g = golden("stereo_n2048")
    x = g["pcm_interleaved"]
    with S.Engine(fft_size=2048, hop=512) as e:
        L = e.stft_host(x, sample_stride=2)
        R = e.stft_host(x[1:], sample_stride=2)
    assert_power_close(L, g["power_l"], "L")
    assert_power_close(R, g["power_r"], "R")


def test_empty_and_ragged(S):
    with S.Engine(fft_size=1024, hop=256) as e:
        assert e.stft_host(np.zeros(1023, np.float32)).shape == (0, 512)
        assert e.stft_host(np.zeros((2, 1024), np.float32)).shape == (2, 1, 512)
        assert e.stft_host(np.zeros((0, 4096), np.float32)).shape == (0, 13, 512)


def test_known_answers_on_gpu(S):
    n = 1024
    with S.Engine(fft_size=n, hop=n, out="mag", nyquist=True) as e:
        x = np.zeros(n, np.float32)
        x[n // 3] = 1.0
        w = 0.5 - 0.5 * np.cos(2 * np.pi * (n // 3) / n)
        assert np.abs(e.stft_host(x)[0] - w).max() < 1e-6          # impulse -> flat |X| = w[n0]
        mag = e.stft_host(np.ones(n, np.float32))[0]
        exp = np.zeros(n // 2 + 1)
        exp[0], exp[1] = n / 2, n / 4
        assert np.abs(mag - exp).max() < 1e-4 * n                  # Hann DC
        k0 = 100
        mag = e.stft_host(np.cos(2 * np.pi * k0 * np.arange(n) / n).astype(np.float32))[0]
        exp = np.zeros(n // 2 + 1)
        exp[k0], exp[k0 - 1], exp[k0 + 1] = n / 4, n / 8, n / 8
        assert np.abs(mag - exp).max() < 1e-4 * n


def test_full_size_properties(S):
    """BASELINE.json configs[1] at full size (10 min mono 48 kHz, N=1024, hop=256):
    size-independent properties instead of an oracle pass."""
    import torch
    n, hop, n_samples = 1024, 256, 28_800_000
    g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    x = torch.rand(n_samples, device="cuda", generator=g) * 2 - 1
    with S.Engine(fft_size=n, hop=hop, window="rect", nyquist=True) as e:
        P = e.stft(x[None, :])[0]
        assert P.shape == (112_497, 513)
        # Parseval per frame (rectangular window): sum_two_sided |X|^2 = N sum x^2
        fr = x.unfold(0, n, hop)
        lhs = 2 * P.double().sum(1) - P[:, 0].double() - P[:, -1].double()
        rhs = n * (fr.double() ** 2).sum(1)
        assert float(((lhs - rhs).abs() / rhs).max()) < 1e-5
        # linearity: STFT power of 2x is 4x, exactly (power-of-two scaling is exact in float)
        P2 = e.stft((2 * x)[None, :])[0]
        assert torch.equal(P2, 4 * P)
        # determinism / idempotence
        assert torch.equal(e.stft(x[None, :])[0], P)
    # time-shift invariance of the framing: dropping `hop` samples shifts frames by one
    with S.Engine(fft_size=n, hop=hop) as e:
        A = e.stft(x[None, :])[0]
        B = e.stft(x[None, hop:])[0]
        assert torch.equal(A[1:], B)
        # random frames against torch's own FFT in float64
        idx = torch.randint(0, A.shape[0], (64,), device="cuda")
        w = 0.5 - 0.5 * torch.cos(2 * torch.pi * torch.arange(n, device="cuda", dtype=torch.float64) / n)
        ref = torch.fft.rfft(x.unfold(0, n, hop)[idx].double() * w, dim=1).abs() ** 2
        err = power_error(A[idx].cpu().numpy(), ref[:, :512].cpu().numpy())
        assert err <= REL, err


def test_analyser_object(S, oracle):
    """The AnalyserNode mirror as the reference drives it (UI/player.js:10-11,
    3D/visualizer.js:346-368): fftSize 2048, tau 0 then 0.75, byte polls into a caller array."""
    g = golden("analyser_n2048")
    n, hop, x = int(g["fft_size"]), int(g["hop"]), g["pcm"]
    a = S.Analyser()
    a.fftSize = 2048
    a.smoothingTimeConstant = 0
    assert a.frequencyBinCount == 1024
    buf = np.zeros(a.frequencyBinCount, np.uint8)
    P, _ = oracle.stft(oracle.OracleConfig.make(n, hop, window=1, scale=1), x)   # float64 power of every polled frame
    a.push(x[:n])
    for f in range(g["bytes_tau0"].shape[0]):
        if f:
            a.push(x[n + (f - 1) * hop: n + f * hop])
        assert a.getByteFrequencyData(buf) is None
        assert_bytes_in_bounds(buf, *byte_bounds(P[f]), "tau 0, frame %d" % f, g["bytes_tau0"][f])
    # smoothing 0.75 from a fresh analyser
    a.fftSize = 2048  # re-plan: history and smoothing state restart
    a.smoothingTimeConstant = 0.75
    a.push(x[:n])
    bb = ByteBounds(n // 2)
    for f in range(g["bytes_tau075"].shape[0]):
        if f:
            a.push(x[n + (f - 1) * hop: n + f * hop])
        a.getByteFrequencyData(buf)
        assert_bytes_in_bounds(buf, *bb.step(P[f], 0.75), "tau 0.75, frame %d" % f, g["bytes_tau075"][f])
    # float dB poll and short / long caller arrays
    a.fftSize = 2048
    a.smoothingTimeConstant = 0
    a.push(x[-n:])
    fl = np.zeros(1024, np.float32)
    a.getFloatFrequencyData(fl)
    st = np.zeros(1024)
    db, _ = oracle.analyser_poll(n, x[-n:], 0.0, -100.0, -30.0, st)
    m = db > -120
    assert np.abs(fl[m] - db[m]).max() < 2e-2
    short = np.full(10, 7, np.uint8)
    a.getByteFrequencyData(short)
    a.getByteFrequencyData(buf)
    assert np.array_equal(short, buf[:10])
    long_ = np.full(2000, 7, np.uint8)
    a.getByteFrequencyData(long_)
    assert np.array_equal(long_[:1024], buf) and np.all(long_[1024:] == 7)  # excess untouched
    td = np.zeros(1024, np.uint8)
    a.getByteTimeDomainData(td)
    assert np.array_equal(td, oracle.time_domain_bytes(x[-n:][:1024]))
    with pytest.raises(TypeError):
        a.getByteFrequencyData(np.zeros(8, np.float32))
    a.close()


def test_poll_without_push_returns_same_block(S, oracle):
    """An AnalyserNode computes one frequency frame per block of audio however often it is read: float and byte
    polls in the same draw, or two byte polls, must not run the smoother twice.  With tau = 0.5 a
    second application would be visible in every bin."""
    n = 512
    rng = np.random.default_rng(77)
    x = rng.uniform(-1, 1, 3 * n).astype(np.float32)
    a = S.Analyser(fftSize=n)
    a.smoothingTimeConstant = 0.5
    st = np.zeros(n // 2)
    fl, fl2 = np.zeros(n // 2, np.float32), np.zeros(n // 2, np.float32)
    by, by2 = np.zeros(n // 2, np.uint8), np.zeros(n // 2, np.uint8)
    P, _ = oracle.stft(oracle.OracleConfig.make(n, n, window=1, scale=1), x)
    bb = ByteBounds(n // 2)
    for blk in range(3):
        a.push(x[blk * n:(blk + 1) * n])
        a.getFloatFrequencyData(fl)          # advances the smoother once for this block
        a.getByteFrequencyData(by)           # same block: re-emitted from the state
        a.getByteFrequencyData(by2)
        a.getFloatFrequencyData(fl2)
        db, ref_b = oracle.analyser_poll(n, x[blk * n:(blk + 1) * n], 0.5, -100.0, -30.0, st)
        m = db > -120
        assert np.abs(fl[m] - db[m]).max() < 2e-2, blk
        assert np.array_equal(fl, fl2) and np.array_equal(by, by2)
        assert_bytes_in_bounds(by, *bb.step(P[blk], 0.5), "block %d" % blk, ref_b)
    a.close()


def test_history_pinned_row(S):
    """Outside the sonogram modes the reference keeps yoffset = 0 (3D/visualizer.js:385-390): pinned, every
    poll lands in row 0 and the offset does not move; unpinned again, the ring resumes."""
    n = 256
    rng = np.random.default_rng(9)
    a = S.Analyser(fftSize=n)
    a.smoothingTimeConstant = 0
    a.startHistory(4)
    a.pinHistory(True)
    buf = np.zeros(n // 2, np.uint8)
    rows = []
    for i in range(3):
        a.push(rng.uniform(-1, 1, n).astype(np.float32))
        a.getByteFrequencyData(buf)
        rows.append(buf.copy())
        assert a.historyOffset == 0
        img = a.readHistory()[0]
        assert np.array_equal(img[0], buf) and not img[1:].any()
    a.pinHistory(False)
    for i in range(2):
        a.push(rng.uniform(-1, 1, n).astype(np.float32))
        a.getByteFrequencyData(buf)
        rows.append(buf.copy())
    assert a.historyOffset == 2
    img = a.readHistory()[0]          # oldest first: rows 2, 3 (zeros), then 0, 1
    assert not img[0].any() and not img[1].any()
    assert np.array_equal(img[2], rows[3]) and np.array_equal(img[3], rows[4])
    a.close()


def test_streaming_full_config4_512_channels(S, oracle):
    """BASELINE.json configs[3] at full size: 512 channels, N = 4096, 1024 new samples per hop, pushes from
    page-locked memory (SG_IN_PINNED), one fused launch per poll, byte rows + device history ring.  Oracle on
    sampled channels; every channel checked against its own duplicate-input twin."""
    import ctypes as C
    from spectrogram_amd import _lib as L
    n, hop, n_ch, hops = 4096, 1024, 512, 6
    m = n // 2
    rng = np.random.default_rng(0x5EED0004)
    base = rng.uniform(-1, 1, (n_ch // 2, n + hops * hop)).astype(np.float32)
    x = np.concatenate([base, base])                 # channel c + 256 carries channel c's samples
    lib = L.load()
    cfg = S.make_config(fft_size=n, hop=n, window="blackman", scale="inv_n", out="mag", smoothing=0.0)
    h = C.c_void_p()
    L.check(lib.sg_create(C.byref(cfg), C.byref(h)))
    L.check(lib.sg_stream_begin(h, n_ch))
    L.check(lib.sg_ring_begin(h, 4))
    pin = C.c_void_p()
    L.check(lib.sg_alloc_pinned(n_ch * n * 4, C.byref(pin)))
    stage = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_float)), shape=(n_ch * n,))
    out = np.zeros((n_ch, m), np.uint8)
    try:
        for hp in range(hops + 1):
            new = x[:, :n] if hp == 0 else x[:, n + (hp - 1) * hop: n + hp * hop]
            w = new.shape[1]
            stage[:n_ch * w].reshape(n_ch, w)[:] = new
            L.check(lib.sg_push(h, pin, w, w, L.IN_PINNED, None))
            L.check(lib.sg_poll_u8(h, out.ctypes.data, m, m, L.PTR_HOST, None))
            assert np.array_equal(out[:n_ch // 2], out[n_ch // 2:]), hp
            for c in (0, 131, 255, 511):
                st = np.zeros(m)
                seg = x[c, hp * hop: hp * hop + n]
                _, ref_b = oracle.analyser_poll(n, seg, 0.0, -100.0, -30.0, st)
                P, _ = oracle.stft(oracle.OracleConfig.make(n, n, window=1, scale=1), seg)
                assert_bytes_in_bounds(out[c], *byte_bounds(P[0]), "hop %d channel %d" % (hp, c), ref_b)
        img = np.zeros((n_ch, 4, m), np.uint8)
        L.check(lib.sg_ring_read(h, img.ctypes.data, 0, L.PTR_HOST, None))
        assert np.array_equal(img[:, 3], out)        # newest row last
    finally:
        lib.sg_free_pinned(pin)
        lib.sg_destroy(h)


def test_analyser_bank_streaming_config4(S, oracle):
    """BASELINE.json configs[3] in miniature: a bank of channels, N=4096, 1024 new samples per
    hop, one frame per channel per hop from device-resident history."""
    n, hop, n_ch, hops = 4096, 1024, 8, 14  # 14 hops: three and a half trips around the N-sample history ring
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, (n_ch, n + hops * hop)).astype(np.float32)
    a = S.Analyser(fftSize=n, n_channels=n_ch, window="hann")
    a.smoothingTimeConstant = 0
    a.push(x[:, :n])
    out = np.zeros((n_ch, n // 2), np.float32)
    cfg = oracle.OracleConfig.make(n, hop, scale=1, out_kind=2, floor_db=-400.0)
    for h in range(hops + 1):
        if h:
            a.push(x[:, n + (h - 1) * hop: n + h * hop])
        a.getFloatFrequencyData(out)
        for c in (0, n_ch - 1):
            ref, _ = oracle.stft(cfg, x[c, h * hop: h * hop + n])
            assert_power_close(10.0 ** (out[c].astype(np.float64) / 10), 10.0 ** (ref[0] / 10),
                               "hop %d ch %d" % (h, c))
    a.close()


@pytest.mark.parametrize("align", ["centre", "vertex"])
@pytest.mark.parametrize("n,hop,out", [(1024, 256, "db_u8"), (2048, 512, "db_u8"), (1024, 256, "power"),
                                       (256, 128, "db"), (4096, 1024, "mag"), (1024, 300, "db_u8")])
def test_logfreq_epilogue(S, oracle, n, hop, out, align):
    """the x = 256^(u-1) column lookup of the reference's shaders
    (sonogram-vertex.shader:51-52, LINEAR / CLAMP_TO_EDGE) fused into the STFT epilogue, at pixel centres (the fragment
    shader) and at the vertices of the 3-D sonogram's mesh, u = c / (cols - 1) (3D/visualizer.js:213-214)."""
    rng = np.random.default_rng(n + hop)
    x = (rng.uniform(-1, 1, n + 30 * hop) * 0.5).astype(np.float32)
    kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out)
    cols = 256
    with S.Engine(**kw) as e:
        rows = e.stft_host(x).astype(np.float64)
    with S.Engine(logfreq_cols=cols, logfreq_align=align, **kw) as e:
        assert e.bins == cols
        got = e.stft_host(x)
    assert got.shape == (rows.shape[0], cols)
    # (a) the resampling step itself, exactly: oracle lookup applied to the GPU's own bin rows
    exp = np.stack([oracle.logfreq_resample(r, cols, align) for r in rows])
    scale = max(1.0, np.abs(exp).max())
    if out == "db_u8":
        # byte columns: the LINEAR blend of two byte texels rounded to the nearest byte
        assert got.dtype == np.uint8
        d = np.abs(got.astype(np.float64) - exp)
        assert d.max() <= 0.5 + 1e-3, d.max()
        got = got.astype(np.float64)
    else:
        assert got.dtype == np.float32
        assert np.abs(got - exp).max() <= 2e-5 * scale
    # (b) end to end against the oracle
    okind = {"power": 0, "mag": 1, "db": 2, "db_u8": 3}[out]
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=okind)
    ref, ref_u8 = oracle.stft(cfg, x)
    ref_rows = ref_u8.astype(np.float64) if out == "db_u8" else ref
    exp2 = np.stack([oracle.logfreq_resample(r, cols, align) for r in ref_rows])
    if out == "db_u8":
        # every byte column inside the image of the power tolerance: blend of the two texels' byte intervals, rounded
        P, _ = oracle.stft(oracle.OracleConfig.make(n, hop, window=1, scale=1), x)
        assert_bytes_in_bounds(got, *logfreq_byte_bounds(*byte_bounds(P, floor_db=-200.0), cols, align), "byte columns",
                               np.floor(exp2 + 0.5))
    elif out == "db":
        m = exp2 > -150
        assert np.abs(got[m] - exp2[m]).max() <= 5e-2
    else:
        assert np.abs(got - exp2).max() <= 1e-4 * np.abs(exp2).max()
    with pytest.raises(S.SpectroError):
        S.Engine(logfreq_cols=256, nyquist=True)
    # the column table must fit in LDS behind the exchange buffers: rejected at create time, not at launch
    with pytest.raises(S.SpectroError) as ei:
        S.Engine(fft_size=16384, hop=4096, logfreq_cols=4096)
    assert ei.value.code == -4  # SG_ERR_UNSUPPORTED


def test_sharded_matches_unsharded(S):
    """Multi-GPU partition (spectrogram_amd/shard.py) exercised on one device: the ranks' rows,
    concatenated, are bit-identical to the unsharded call -- frame shards (one long stream, with
    halo) and channel shards."""
    import torch
    from spectrogram_amd import shard
    g = torch.Generator(device="cuda").manual_seed(5)
    long_x = torch.rand(1, 1_000_000, device="cuda", generator=g) * 2 - 1
    many = torch.rand(7, 50_000, device="cuda", generator=g) * 2 - 1
    with S.Engine(fft_size=1024, hop=256) as e:
        full = e.stft(long_x)
        parts = [shard.stft_sharded(e, long_x, 4, r) for r in range(4)]
        assert torch.equal(torch.cat([p[0] for p in parts], dim=1), full)
        assert sum(p[1].frame_count for p in parts) == full.shape[1]
        fullm = e.stft(many)
        partsm = [shard.stft_sharded(e, many, 4, r) for r in range(4)]
        assert torch.equal(torch.cat([p[0] for p in partsm if p[0] is not None], dim=0), fullm)


def test_history_ring(S):
    """the 256-row byte history the reference keeps as a texture
    (3D/visualizer.js:399-417: row at yoffset, yoffset = (yoffset + 1) % 256), kept on device."""
    n, H, n_ch = 512, 8, 2
    rng = np.random.default_rng(9)
    a = S.Analyser(fftSize=n, n_channels=n_ch)
    a.smoothingTimeConstant = 0
    a.startHistory(H)
    assert a.historyOffset == 0 and not a.readHistory().any()   # zero-filled like the reference's tmp array
    rows = []
    buf = np.zeros((n_ch, n // 2), np.uint8)
    for i in range(H + 3):                                       # wraps around
        a.push(rng.uniform(-1, 1, (n_ch, 200)).astype(np.float32) * (i + 1) / 12)
        a.getByteFrequencyData(buf)
        rows.append(buf.copy())
        assert a.historyOffset == (i + 1) % H
    img = a.readHistory()
    assert img.shape == (n_ch, H, n // 2)
    want = np.stack(rows[-H:], axis=1)                           # oldest of the last H rows first
    assert np.array_equal(img, want)
    short = np.zeros((n_ch, 10), np.uint8)                       # a short caller array still records the full row
    a.push(rng.uniform(-1, 1, (n_ch, 200)).astype(np.float32))
    a.getByteFrequencyData(short)
    full = a.readHistory()[:, -1, :]
    assert np.array_equal(full[:, :10], short) and full[:, 10:].any()
    a.close()


@pytest.mark.parametrize("n,hop", [(1024, 256), (1024, 512), (2048, 512), (256, 128), (1024, 333), (4096, 1024)])
def test_int16_ingest(S, oracle, n, hop):
    """16-bit PCM converted inside the framing load (x / 32768)."""
    import torch
    rng = np.random.default_rng(n ^ hop)
    xi = rng.integers(-32768, 32767, (2, n + 45 * hop + 3), dtype=np.int16)
    with S.Engine(fft_size=n, hop=hop, pcm="s16") as e:
        got_h = e.stft_host(xi)
        got_d = e.stft(torch.from_numpy(xi).cuda()).cpu().numpy()
    assert np.array_equal(got_h, got_d)
    cfg = oracle.OracleConfig.make(n, hop)
    for c in range(2):
        ref, _ = oracle.stft(cfg, xi[c].astype(np.float32) / 32768.0)   # exact in float32
        assert_power_close(got_h[c], ref, "s16 n=%d hop=%d ch=%d" % (n, hop, c))
    # interleaved stereo int16 (sample_stride = 2)
    with S.Engine(fft_size=n, hop=hop, pcm="s16") as e:
        inter = np.ascontiguousarray(xi.T).reshape(-1)
        L = e.stft_host(inter, sample_stride=2)
    assert_power_close(L, oracle.stft(cfg, xi[0].astype(np.float32) / 32768.0)[0], "s16 interleaved")


@pytest.mark.parametrize("out", ["mag", "db", "db_u8", "power"])
def test_batch_smoothing(S, oracle, out):
    """tau != 0 in batch mode (the reference's bar-graph mode sets 0.75, 3D/visualizer.js:351):
    X^ = tau X^_prev + (1 - tau)|X| along the frames of each channel, then the output kind."""
    n, hop, tau = 1024, 256, 0.75
    rng = np.random.default_rng(21)
    x = (rng.uniform(-1, 1, (2, n + 60 * hop)) * np.linspace(0.05, 1, n + 60 * hop)).astype(np.float32)
    with S.Engine(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out, smoothing=tau) as e:
        got = e.stft_host(x)
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=1)
    for c in range(2):
        mag, _ = oracle.stft(cfg, x[c])
        sm = np.zeros(n // 2)
        ref = np.empty_like(mag)
        for f in range(mag.shape[0]):
            sm = tau * sm + (1 - tau) * mag[f]
            ref[f] = sm
        if out == "mag":
            assert_power_close(got[c].astype(np.float64) ** 2, ref ** 2, "smoothed mag")
        elif out == "power":
            assert_power_close(got[c], ref ** 2, "smoothed power")
        elif out == "db":
            assert_power_close(10.0 ** (got[c].astype(np.float64) / 10), ref ** 2, "smoothed db")
        else:
            with np.errstate(divide="ignore"):
                b = np.clip(np.floor(255.0 / 70.0 * (20 * np.log10(ref) + 100.0)), 0, 255)
            assert_bytes_in_bounds(got[c], *byte_bounds(mag ** 2, tau, floor_db=-200.0), "smoothed bytes", b)


@pytest.mark.parametrize("tau", [0.5, 0.75, 0.93])
def test_batch_smoothing_chunked_scan(S, oracle, tau):
    """Long enough for the frame axis to be cut into chunks with a warm-up overlap (tau^warm < 2^-30):
    every chunk must agree with the serial recurrence on the oracle's magnitudes."""
    n, hop = 256, 64
    rng = np.random.default_rng(31)
    x = (rng.uniform(-1, 1, n + 4000 * hop) * (0.2 + 0.8 * np.abs(np.sin(np.arange(n + 4000 * hop) / 9000.0)))).astype(np.float32)
    with S.Engine(fft_size=n, hop=hop, window="blackman", scale="inv_n", out="mag", smoothing=tau) as e:
        got = e.stft_host(x)
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=1)
    mag, _ = oracle.stft(cfg, x)
    sm = np.zeros(n // 2)
    ref = np.empty_like(mag)
    for f in range(mag.shape[0]):
        sm = tau * sm + (1 - tau) * mag[f]
        ref[f] = sm
    assert got.shape == ref.shape and got.shape[0] == 4001
    assert_power_close(got.astype(np.float64) ** 2, ref ** 2, "chunked scan tau=%g" % tau)


def _serial_smooth(mag, tau):
    sm = np.zeros(mag.shape[1])
    ref = np.empty_like(mag)
    for f in range(mag.shape[0]):
        sm = tau * sm + (1 - tau) * mag[f]
        ref[f] = sm
    return ref


@pytest.mark.parametrize("n,hop,n_ch,out", [(1024, 256, 2048, "db_u8"), (1024, 300, 2100, "mag"), (2048, 512, 1024, "power"),
                                            (256, 64, 8192, "db")])
def test_fused_batch_smoothing_many_channels(S, oracle, n, hop, n_ch, out):
    """With at least half as many channels as the chip has frame slots, tau != 0 in batch mode is ONE launch: a run is
    a whole channel and the smoothed magnitudes ride in registers (no |X| scratch; configs 3 and 5 have this shape).
    Sampled channels against the float64 recurrence on the oracle's magnitudes; every channel against the two-pass
    path (scratch + chunked scan) under the same tolerance."""
    import torch
    tau, frames = 0.75, 45
    rng = np.random.default_rng(n + n_ch)
    ns = n + (frames - 1) * hop + 7
    x = (rng.uniform(-1, 1, (n_ch, ns)) * rng.uniform(0.01, 1.0, (n_ch, 1))).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out, smoothing=tau)
    with S.Engine(**kw) as e:
        got = e.stft(xd)
        torch.cuda.synchronize()
        variant, run_len, single, total_runs = e.last_launch()
        assert (run_len, total_runs) == (frames, n_ch), "fused smoothing walks a whole channel per run"
    with S.Engine(debug={"no_fused_smoothing": 1}, **kw) as e:
        two = e.stft(xd)
        torch.cuda.synchronize()
    got, two = got.cpu().numpy(), two.cpu().numpy()

    def as_power(a):
        a = a.astype(np.float64)
        return {"mag": a ** 2, "power": a, "db": 10.0 ** (a / 10)}[out]

    if out != "db_u8":   # (bytes: both paths against the oracle's intervals on the sampled channels below)
        assert power_error(as_power(got), as_power(two)) <= 2 * REL
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=1)
    for c in (0, n_ch // 3, n_ch - 1):
        mag, _ = oracle.stft(cfg, x[c])
        ref = _serial_smooth(mag, tau)
        if out == "db_u8":
            with np.errstate(divide="ignore"):
                b = np.clip(np.floor(255.0 / 70.0 * (20 * np.log10(ref) + 100.0)), 0, 255)
            bounds = byte_bounds(mag ** 2, tau, floor_db=-200.0)
            assert_bytes_in_bounds(got[c], *bounds, "fused smoothing, bytes, ch %d" % c, b)
            assert_bytes_in_bounds(two[c], *bounds, "two-pass smoothing, bytes, ch %d" % c)
        else:
            assert_power_close(as_power(got[c]), ref ** 2, "fused smoothing %s ch %d" % (out, c))


@pytest.mark.parametrize("n,hop,n_ch,out,align", [(1024, 256, 3, "db_u8", "vertex"), (2048, 512, 2, "power", "centre"),
                                                  (512, 128, 5, "db", "centre"), (1024, 300, 4, "mag", "vertex"),
                                                  (1024, 256, 2048, "db_u8", "vertex"), (2048, 512, 1024, "mag", "centre"),
                                                  (256, 64, 8192, "power", "vertex"), (2048, 735, 1100, "db", "centre")])
def test_batch_smoothing_with_logfreq_columns(S, oracle, n, hop, n_ch, out, align):
    """tau != 0 together with cfg.logfreq_cols in batch mode (refused until round 6).  The order is the
    reference's: the analyser smooths its bins (3D/visualizer.js:351), the shader samples the row it hands out
    (sonogram-vertex.shader:51-52).  Few channels: |X| scratch + chunked scan that blends the two bins of every column; many
    channels: ONE launch, the column lookup behind the smoothing recurrence that rides in registers (OUT_SMOOTH | OUT_LOGFREQ).
    (a) the oracle's lookup applied to the SAME engine's smoothed bin rows (validated against the float64 recurrence by the
    tests above); (b) sampled channels end to end against the float64 oracle."""
    import torch
    tau, frames, cols = 0.75, 40, 256
    rng = np.random.default_rng(n + n_ch + hop)
    ns = n + (frames - 1) * hop + 5
    x = (rng.uniform(-1, 1, (n_ch, ns)) * rng.uniform(0.01, 1.0, (n_ch, 1))).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out, smoothing=tau)
    with S.Engine(**kw) as e:
        rows = e.stft(xd).cpu().numpy().astype(np.float64)
    with S.Engine(logfreq_cols=cols, logfreq_align=align, **kw) as e:
        assert e.bins == cols
        got = e.stft(xd)
        torch.cuda.synchronize()
        fused = e.last_launch()[1:] == (frames, 0, n_ch)     # a run is a whole channel
        assert fused == (n_ch >= 1000), e.last_launch()
    got = got.cpu().numpy()
    assert got.shape == (n_ch, frames, cols)
    chans = sorted({0, n_ch // 3, n_ch - 1})
    okind = {"power": 0, "mag": 1, "db": 2, "db_u8": 3}[out]
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=1)
    for c in chans:
        exp = np.stack([oracle.logfreq_resample(r, cols, align) for r in rows[c]])
        if out == "db_u8":
            assert got.dtype == np.uint8 and np.abs(got[c].astype(np.float64) - exp).max() <= 0.5 + 1e-3
        else:
            assert np.abs(got[c] - exp).max() <= 2e-5 * max(1.0, np.abs(exp).max())
        mag, _ = oracle.stft(cfg, x[c])
        sm = _serial_smooth(mag, tau)
        with np.errstate(divide="ignore"):
            y = np.maximum(20 * np.log10(sm), -200.0)
        val = {"power": sm ** 2, "mag": sm, "db": y, "db_u8": np.clip(np.floor(255.0 / 70.0 * (y + 100.0)), 0, 255)}[out]
        exp2 = np.stack([oracle.logfreq_resample(r, cols, align) for r in val])
        g = got[c].astype(np.float64)
        if out == "db_u8":
            bounds = logfreq_byte_bounds(*byte_bounds(mag ** 2, tau, floor_db=-200.0), cols, align)
            assert_bytes_in_bounds(g, *bounds, "smoothed byte columns ch %d" % c, np.floor(exp2 + 0.5))
        elif out == "db":
            m = exp2 > -150
            assert np.abs(g[m] - exp2[m]).max() <= 5e-2
        else:
            assert np.abs(g - exp2).max() <= 1e-4 * np.abs(exp2).max()
    # the two implementations agree with each other on every channel (many channels: fused against the forced two-pass path)
    if fused:
        with S.Engine(logfreq_cols=cols, logfreq_align=align, debug={"no_fused_smoothing": 1}, **kw) as e:
            two = e.stft(xd).cpu().numpy()
        d = np.abs(got.astype(np.float64) - two.astype(np.float64))
        if out == "db_u8":
            assert d.max() <= 1 and (d > 0).mean() < 2e-3
        elif out == "db":
            assert d[two > -150].max() <= 5e-2
        else:
            assert d.max() <= 2e-4 * np.abs(two).max()


def test_two_pass_smoothing_scratch_on_two_streams(S):
    """with cfg.smoothing > 0 and few channels sg_stft runs |X| into the handle's scratch and a scan over it, and with
    device output it only enqueues.  Two such calls back to back on two streams share that scratch: the second must wait
    (device-side) for the first call's scan, or the first call's rows are smoothed from the second call's magnitudes."""
    import ctypes as C
    import torch
    from spectrogram_amd import _lib as L
    lib = L.load()
    n, hop, n_ch, ns = 1024, 256, 6, 1_500_000
    with S.Engine(fft_size=n, hop=hop, smoothing=0.6, out="mag") as e:
        frames = e.num_frames(ns)
        xs = [(torch.rand(n_ch, ns, device="cuda", generator=torch.Generator(device="cuda").manual_seed(90 + i)) * 2 - 1) for i in range(3)]
        want = []
        for x in xs:
            o = torch.empty(n_ch, frames, n // 2, device="cuda")
            L.check(lib.sg_stft(e._h, x.data_ptr(), n_ch, ns, ns, 1, o.data_ptr(), 0, L.PTR_DEVICE, None))
            torch.cuda.synchronize()
            want.append(o)
        outs = [torch.zeros(n_ch, frames, n // 2, device="cuda") for _ in range(3)]
        streams = [torch.cuda.Stream() for _ in range(3)]
        torch.cuda.synchronize()
        for x, o, st in zip(xs, outs, streams):        # back to back, no synchronisation in between
            L.check(lib.sg_stft(e._h, x.data_ptr(), n_ch, ns, ns, 1, o.data_ptr(), 0, L.PTR_DEVICE, C.c_void_p(st.cuda_stream)))
        torch.cuda.synchronize()
        for i in range(3):
            assert torch.equal(outs[i], want[i]), "call %d was smoothed from another call's magnitudes" % i


@pytest.mark.parametrize("cols", [256, 2048])
def test_logfreq_columns_at_the_largest_fft_size(S, oracle, cols):
    """cfg.logfreq_cols at fft_size 32768 (refused until round 6 for "16-bit row offsets": the largest
    offset, 4 * 16383, still fits).  Batch rows against the oracle's lookup on the same engine's bin rows; a poll of the
    same size against its own bins."""
    n, hop = 32768, 8192
    rng = np.random.default_rng(cols)
    x = (rng.uniform(-1, 1, (2, n + 5 * hop + 3)) * 0.5).astype(np.float32)
    for out in ("power", "db_u8"):
        kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out)
        with S.Engine(**kw) as e:
            rows = e.stft_host(x).astype(np.float64)
        with S.Engine(logfreq_cols=cols, logfreq_align="vertex", **kw) as e:
            got = e.stft_host(x)
        assert got.shape == (2, 6, cols)
        exp = np.stack([[oracle.logfreq_resample(r, cols, "vertex") for r in ch] for ch in rows])
        if out == "db_u8":
            assert np.abs(got.astype(np.float64) - exp).max() <= 0.5 + 1e-3
        else:
            assert np.abs(got - exp).max() <= 2e-5 * max(1.0, np.abs(exp).max())
    with pytest.raises(S.SpectroError) as ei:       # the table must still fit behind the 144 KB of exchange buffers
        S.Engine(fft_size=n, hop=hop, logfreq_cols=2049)
    assert ei.value.code == -4
    a = S.Analyser(fftSize=n, logfreq_cols=cols)
    b = S.Analyser(fftSize=n)
    for an in (a, b):
        an.smoothingTimeConstant = 0
        an.push(x[0][:n])
    fa, fb = np.zeros(cols, np.float32), np.zeros(n // 2, np.float32)
    a.getFloatFrequencyData(fa)
    b.getFloatFrequencyData(fb)
    exp = oracle.logfreq_resample(np.maximum(fb.astype(np.float64), -200.0), cols, "vertex")
    assert np.abs(fa - exp).max() <= 2e-3
    a.close()
    b.close()


def _pack24(q):
    u = q & 0xFFFFFF
    return np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=-1).astype(np.uint8).reshape(q.shape[0], -1)


@pytest.mark.parametrize("k", [1, 2])
def test_poll_mode_takes_packed_int24_pushes(S, oracle, k):
    """sg_push with cfg.pcm_format = SG_PCM_S24 (refused until round 6): packed little-endian 24-bit
    samples are converted on the way into the float history (the window carries the 1 / 8388608, as in the batch path);
    bytes and dB rows against the oracle on x / 8388608, the time-domain getter exact, ragged blocks, K-channel mix."""
    n, n_ch = 1024, 2
    rng = np.random.default_rng(240 + k)
    a = S.Analyser(fftSize=n, n_channels=n_ch, pcm="s24", downmix=k)
    a.smoothingTimeConstant = 0
    hist = [np.zeros(0) for _ in range(n_ch)]
    for blk in (700, 333, 1024, 90):
        q = rng.integers(-(1 << 23), 1 << 23, (n_ch, blk * k), dtype=np.int64)
        q[0, :4] = [-(1 << 23), (1 << 23) - 1, -1, 255]
        a.push(_pack24(q))
        for c in range(n_ch):
            mono = (q[c].reshape(blk, k).astype(np.float32).sum(axis=1, dtype=np.float32) * np.float32(1.0 / k)) if k > 1 else q[c].astype(np.float32)
            hist[c] = np.concatenate([hist[c], mono / np.float32(8388608.0)])[-n:]
        by, fl, td = np.zeros((n_ch, n // 2), np.uint8), np.zeros((n_ch, n // 2), np.float32), np.zeros((n_ch, n), np.float32)
        a.getByteFrequencyData(by)
        a.getFloatFrequencyData(fl)
        a.getFloatTimeDomainData(td)
        for c in range(n_ch):
            last = np.concatenate([np.zeros(n - len(hist[c])), hist[c]]).astype(np.float32)
            assert np.array_equal(td[c], last), "time-domain samples of int24 pushes are x / 8388608 exactly"
            st = np.zeros(n // 2)
            db, ref_by = oracle.analyser_poll(n, last, 0.0, -100.0, -30.0, st)
            assert_bytes_in_bounds(by[c], *byte_bounds(st ** 2), "int24 poll bytes", ref_by)
            m = db > -140
            assert np.abs(fl[c][m] - db[m]).max() <= 2e-2
    with pytest.raises(TypeError):
        a.push(np.zeros((n_ch, 30), np.float32))
    a.close()


@pytest.mark.parametrize("k", [2, 3, 4, 6])
def test_speakers_downmix(S, oracle, k):
    """SG_DOWNMIX_SPEAKERS, the W3C mono down-mix of an AnalyserNode's input -- K = 6 (L R C LFE SL SR):
    sqrt(1/2) (L + R) + C + 0.5 (SL + SR); K = 2 / 4: the mean (bit for bit the default's rows); any other K: "discrete",
    channel 0 alone.  Batch (host and device input, int16 too) and poll mode, against the oracle on the float64 mix."""
    import torch
    n, hop = 1024, 256
    rng = np.random.default_rng(600 + k)
    ns = n + 20 * hop
    x = (rng.uniform(-1, 1, (2, ns * k)) * 0.4).astype(np.float32)
    w = {2: [0.5, 0.5], 3: [1, 0, 0], 4: [0.25] * 4, 6: [np.sqrt(0.5), np.sqrt(0.5), 1, 0, 0.5, 0.5]}[k]
    mixed = (x.reshape(2, ns, k).astype(np.float64) * np.asarray(w)).sum(axis=2)
    cfg = oracle.OracleConfig.make(n, hop)
    with S.Engine(fft_size=n, hop=hop, downmix=k, downmix_mode="speakers") as e:
        got = e.stft_host(x)
        gd = e.stft(torch.from_numpy(x).cuda()).cpu().numpy()
    with S.Engine(fft_size=n, hop=hop, downmix=k) as e:
        mean = e.stft_host(x)
    assert np.array_equal(got, gd)
    assert np.array_equal(got, mean) == (k in (2, 4))
    for c in range(2):
        ref, _ = oracle.stft(cfg, mixed[c].astype(np.float32))
        assert_power_close(got[c], ref, "speakers mix K=%d ch %d" % (k, c))
    xi = (x * 32767).astype(np.int16)
    with S.Engine(fft_size=n, hop=hop, downmix=k, downmix_mode="speakers", pcm="s16") as e:
        g16 = e.stft_host(xi)
    mixed16 = ((xi.reshape(2, ns, k).astype(np.float64) / 32768.0) * np.asarray(w)).sum(axis=2)
    ref, _ = oracle.stft(cfg, mixed16[0].astype(np.float32))
    assert_power_close(g16[0], ref, "speakers mix of int16 K=%d" % k)
    # poll mode: sg_push applies the same weights on the way into the history
    a = S.Analyser(fftSize=n, n_channels=2, downmix=k, downmix_mode="speakers")
    a.smoothingTimeConstant = 0
    a.push(x[:, :n * k])
    td = np.zeros((2, n), np.float32)
    a.getFloatTimeDomainData(td)
    assert np.abs(td - mixed[:, :n]).max() <= 4e-7
    fl = np.zeros((2, n // 2), np.float32)
    a.getFloatFrequencyData(fl)
    st = np.zeros(n // 2)
    db, _ = oracle.analyser_poll(n, mixed[0, :n].astype(np.float32), 0.0, -100.0, -30.0, st)
    m = db > -140
    assert np.abs(fl[0][m] - db[m]).max() <= 2e-2
    a.close()
    with pytest.raises(KeyError):
        S.Engine(downmix=2, downmix_mode="surround")


def test_randomized_batch_smoothing(S, oracle):
    """Seeded sweep over tau != 0 in batch mode: both implementations (few channels: |X| scratch + chunked scan, long
    enough to cross chunk boundaries; many channels: the recurrence fused into the transform, a whole channel per run),
    every load variant underneath, float32 / int16 input, every output kind, and (round 6) the row handed out as
    log-frequency columns of the smoothed bins, at either alignment -- sampled channels against the float64 recurrence on
    the oracle's magnitudes."""
    import os
    import torch
    rng = np.random.default_rng(int(os.environ.get("SG_FUZZ_SEED", "20261004")) + 31)
    fused_seen = scan_seen = 0
    for case in range(max(4, int(os.environ.get("SG_FUZZ_CASES", "48")) // 4)):
        n = int(2 ** rng.integers(8, 13))                        # 256 .. 4096
        hop = int(rng.choice([n // 4, n // 2, max(1, n // 3), 1 + int(rng.integers(0, n - 4)), n, n + 3]))
        tau = float(rng.uniform(0.05, 0.97))
        out = ["power", "mag", "db", "db_u8"][int(rng.integers(0, 4))]
        pcm = ["f32", "s16"][int(rng.integers(0, 2))]
        many = bool(rng.integers(0, 2))
        cols = int(rng.choice([0, 0, 37, 256])) if 512 <= n <= 4096 else 0
        align = ["centre", "vertex"][int(rng.integers(0, 2))]
        if many:
            n_ch = 2_097_152 // n + int(rng.integers(0, 50))     # >= half the chip's frame slots: the fused path
            frames = int(rng.integers(1, 24))
        else:
            n_ch = int(rng.integers(1, 4))
            frames = int(rng.choice([1, 2, 37, 300, 1500 if n <= 1024 else 400]))
        ns = n + (frames - 1) * hop + int(rng.integers(0, hop))
        amp = rng.uniform(0.01, 1.0, (n_ch, 1))
        if pcm == "s16":
            x = (rng.uniform(-1, 1, (n_ch, ns)) * amp * 32767).astype(np.int16)
            as_f = x.astype(np.float32) / np.float32(32768.0)
        else:
            x = (rng.uniform(-1, 1, (n_ch, ns)) * amp).astype(np.float32)
            as_f = x
        what = "case %d: n=%d hop=%d tau=%.3f %s %s ch=%d frames=%d cols=%d %s" % (case, n, hop, tau, out, pcm, n_ch, frames, cols, align)
        lf = dict(logfreq_cols=cols, logfreq_align=align) if cols else {}
        with S.Engine(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out, smoothing=tau, pcm=pcm, **lf) as e:
            got = e.stft(torch.from_numpy(x).cuda())
            torch.cuda.synchronize()
            _, run_len, _, total_runs = e.last_launch()
            got = got.cpu().numpy()
        if (run_len, total_runs) == (frames, n_ch) and frames > 1:
            fused_seen += 1
        else:
            scan_seen += 1
        assert got.shape == (n_ch, frames, cols or n // 2), what
        cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=1)
        for c in sorted({0, n_ch // 2, n_ch - 1}):
            mag, _ = oracle.stft(cfg, as_f[c])
            ref = _serial_smooth(mag, tau)
            g = got[c].astype(np.float64)
            if cols:
                # the analyser smooths its bins, the shader samples the row it hands out (test_batch_smoothing_with_logfreq_columns)
                with np.errstate(divide="ignore"):
                    y = np.maximum(20 * np.log10(ref), -200.0)
                val = {"power": ref ** 2, "mag": ref, "db": y, "db_u8": np.clip(np.floor(255.0 / 70.0 * (y + 100.0)), 0, 255)}[out]
                exp = np.stack([oracle.logfreq_resample(r, cols, align) for r in val])
                if out == "db_u8":
                    bounds = logfreq_byte_bounds(*byte_bounds(mag ** 2, tau, floor_db=-200.0), cols, align)
                    assert_bytes_in_bounds(g, *bounds, "%s ch %d" % (what, c), np.floor(exp + 0.5))
                elif out == "db":
                    mk = exp > -150
                    assert not mk.any() or np.abs(g[mk] - exp[mk]).max() <= 5e-2, (what, c)
                else:
                    assert np.abs(g - exp).max() <= 1e-4 * np.abs(exp).max() + 1e-30, (what, c)
                continue
            if out == "db_u8":
                with np.errstate(divide="ignore"):
                    b = np.clip(np.floor(255.0 / 70.0 * (20 * np.log10(ref) + 100.0)), 0, 255)
                assert_bytes_in_bounds(g, *byte_bounds(mag ** 2, tau, floor_db=-200.0), "%s ch %d" % (what, c), b)
            else:
                gp = {"mag": g ** 2, "power": g, "db": 10.0 ** (g / 10)}[out]
                assert_power_close(gp, ref ** 2, "%s ch %d" % (what, c))
    if int(os.environ.get("SG_FUZZ_CASES", "48")) >= 48 and "SG_FUZZ_SEED" not in os.environ:   # coverage of the default draw only: another seed may miss a variant
        assert fused_seen >= 2 and scan_seen >= 2, (fused_seen, scan_seen)


def test_batch_smoothing_after_a_loud_passage(S, oracle):
    """a loud passage followed by exact silence.  The serial recurrence keeps decaying from the loud state
    (peak * tau^n); the fused path IS the serial recurrence (relative tolerance all the way down), while the chunked
    scan restarts each chunk from zero `warm` frames early and therefore carries an ABSOLUTE error of at most
    2^-30 of the largest earlier magnitude (aux_kernels.hip) -- asserted as such."""
    import torch
    n, hop, tau = 256, 64, 0.75
    loud, quiet = 900, 400
    rng = np.random.default_rng(5)
    x1 = np.concatenate([rng.uniform(-1, 1, n + (loud - 1) * hop), np.zeros(quiet * hop)]).astype(np.float32)
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=1)
    mag, _ = oracle.stft(cfg, x1)
    ref = _serial_smooth(mag, tau)
    peak = ref.max()
    kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n", out="mag", smoothing=tau)
    # few channels: two launches, chunked scan
    with S.Engine(**kw) as e:
        got = e.stft_host(np.stack([x1, x1]))[1].astype(np.float64)
    err = np.abs(got - ref)
    assert (err <= 1e-4 * ref + 2.0 ** -29 * peak).all(), "chunked scan: beyond 1e-4 relative + 2^-30 of the earlier peak"
    # many channels: fused, exact recurrence -- relative error only, 100+ dB below the loud passage
    xd = torch.from_numpy(np.tile(x1, (8192, 1))).cuda()
    with S.Engine(**kw) as e:
        gf = e.stft(xd)
        assert e.last_launch()[3] == 8192
        gf = gf[4097].double().cpu().numpy()
    tail = slice(loud + 40, loud + 110)                    # tau^40 .. tau^110 of the loud state: -100 .. -270 dB
    live = ref[tail] > 1e-30                               # float32 underflows below ~1e-38
    assert (np.abs(gf[tail] - ref[tail])[live] <= 2e-4 * ref[tail][live]).all()
    assert (np.abs(gf[:loud] - ref[:loud]) <= 1e-4 * np.maximum(ref[:loud], 1e-3 * peak)).all()


def test_batch_smoothing_config2_length(S):
    """The scan at BASELINE configs[1] length (112,497 frames of one channel, tau = 0.75, byte output): the GPU's
    smoothed bytes against the float64 recurrence run over the GPU's own unsmoothed magnitudes (themselves
    parity-tested above), so the check costs no 112k-frame oracle pass."""
    import torch
    n, hop, tau = 1024, 256, 0.75
    g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    x = (torch.rand(1, 48000 * 600, device="cuda", generator=g) * 2 - 1).contiguous()
    kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n")
    with S.Engine(out="mag", **kw) as e:
        mag = e.stft(x)
    with S.Engine(out="db_u8", smoothing=tau, **kw) as e:
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        got = e.stft(x)
        t0.record(); got = e.stft(x, out=got); t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1)
    frames = mag.shape[1]
    assert frames == 112497
    print("smoothed db_u8, 112497 frames: %.3f ms = %.0f GB/s algorithmic (hop*4 in + 512 B out per frame)" %
          (ms, frames * (hop * 4 + 512) / ms / 1e6))
    m = mag[0].double()
    # float64 recurrence on device, blockwise to stay vectorised: s_f = tau s_{f-1} + (1 - tau) m_f
    sm = torch.zeros(n // 2, dtype=torch.float64, device="cuda")
    ref = torch.empty_like(m)
    for f0 in range(0, frames, 64):
        blk = m[f0:f0 + 64]
        L = blk.shape[0]
        pw = tau ** torch.arange(1, L + 1, dtype=torch.float64, device="cuda")[:, None]
        # s_{f0+i} = tau^{i+1} s_in + sum_{j<=i} tau^{i-j} (1 - tau) m_j  (cumulative sum in the scaled domain)
        scaled = torch.cumsum(blk * (1 - tau) / pw, dim=0) * pw
        ref[f0:f0 + L] = scaled + pw * sm
        sm = ref[f0 + L - 1].clone()
    # Same magnitudes on both sides, so what may differ is the float32 recurrence (a few 1e-7 relative) and the chunked
    # scan's restart (2^-30 of the largest earlier magnitude): that interval of the smoothed
    # magnitude, mapped through the (monotone) byte map, is the set of bytes allowed -- one value almost everywhere.
    slack = 2.0 ** -30 * ref.max()

    def byte_of(a, shift):
        return torch.clamp(torch.floor(255.0 / 70.0 * (20 * torch.log10(a.clamp_min(0)) + 100.0) + shift), 0, 255)

    b_lo, b_hi = byte_of(ref * (1 - 4e-7) - slack, -2e-4), byte_of(ref * (1 + 4e-7) + slack, 2e-4)
    gb = got[0].double()
    assert bool(((gb >= b_lo) & (gb <= b_hi)).all()), "smoothed byte outside the interval the float32 scan allows"
    assert (b_hi > b_lo).double().mean().item() < 1e-3


def _check_random_frames(S, P, x, n, hop, n_check=48, seed=0):
    """Random (channel, frame) rows of the GPU result against torch's float64 FFT of the same samples."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    C, F = P.shape[0], P.shape[1]
    ci = torch.randint(0, C, (n_check,), device="cuda", generator=g)
    fi = torch.randint(0, F, (n_check,), device="cuda", generator=g)
    ci[0], fi[0] = C - 1, F - 1                      # the very last row: highest addresses
    ci[1], fi[1] = 0, 0
    w = 0.5 - 0.5 * torch.cos(2 * torch.pi * torch.arange(n, device="cuda", dtype=torch.float64) / n)
    idx = fi[:, None] * hop + torch.arange(n, device="cuda")[None, :]
    fr = x[ci[:, None], idx].double()
    if x.dtype == torch.int16:
        fr = fr / 32768.0
    ref = (torch.fft.rfft(fr * w, dim=1).abs() ** 2)[:, : n // 2]
    err = power_error(P[ci, fi].cpu().numpy(), ref.cpu().numpy())
    assert err <= REL, err


def test_full_size_config3_batched(S):
    """BASELINE.json configs[2] at full size: 1024 x 30 s stereo 48 kHz = 2048 channels x 1.44 M samples
    (11.8 GB in), N=2048, hop 512 -> 5,752,832 frames, 23.6 GB out.  Offsets exceed 2^32 bytes."""
    import torch
    n, hop, C, Ls = 2048, 512, 2048, 1_440_000
    g = torch.Generator(device="cuda").manual_seed(0x5EED0003)
    x = torch.empty(C, Ls, device="cuda")
    for c0 in range(0, C, 256):                      # generate in slabs to bound temporaries
        x[c0:c0 + 256] = torch.rand(256, Ls, device="cuda", generator=g) * 2 - 1
    with S.Engine(fft_size=n, hop=hop) as e:
        assert e.num_frames(Ls) == 2809
        P = e.stft(x)
        torch.cuda.synchronize()
        assert P.shape == (C, 2809, 1024)
        _check_random_frames(S, P, x, n, hop, seed=3)
        # a checksum of checksums: per-channel sums equal those of the same channels run alone
        sub = e.stft(x[C - 3:])
        assert torch.equal(sub, P[C - 3:])
        # and the whole 5.75 M-frame batch again: bit-identical (two-wave frames exchange through LDS under barriers;
        # a race between a fast and a slow wave would show as a run-to-run difference)
        P2 = e.stft(x)
        torch.cuda.synchronize()
        assert torch.equal(P2, P)
        del P2
    del P
    # the same batch with the reference's bar-graph smoothing (tau = 0.75, 3D/visualizer.js:351) and byte rows: ONE
    # launch of 2048 whole-channel runs -- no 23.6 GB |X| scratch.  Property check at full size: the recurrence inverted
    # on the float path, (X^[f] - tau X^[f-1]) / (1 - tau) = |X[f]|, on sampled channels against the unsmoothed output.
    torch.cuda.empty_cache()
    kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n")
    with S.Engine(out="mag", smoothing=0.75, **kw) as e:
        free0 = torch.cuda.mem_get_info()[0]
        Sm = e.stft(x)
        torch.cuda.synchronize()
        _, run_len, _, total_runs = e.last_launch()
        assert (run_len, total_runs) == (2809, C)
        # nothing but the output was allocated (a |X| scratch would be another 23.6 GB)
        assert free0 - torch.cuda.mem_get_info()[0] < Sm.numel() * 4 + (1 << 30)
    with S.Engine(out="mag", **kw) as e:
        for c in (0, 777, C - 1):
            mag = e.stft(x[c:c + 1])[0].double()
            sm = Sm[c].double()
            rec = torch.cat([sm[:1], sm[1:] - 0.75 * sm[:-1]]) / 0.25
            rel = (rec - mag).abs() / (mag.abs().amax(dim=1, keepdim=True) * 1e-3 + mag.abs())
            assert rel.max().item() < 2e-3, (c, rel.max().item())   # the inversion amplifies float32 rounding by ~1/(1 - tau) * 2
    del Sm, x
    torch.cuda.empty_cache()


def test_full_size_config5_chunk_int16(S):
    """One per-GPU chunk of BASELINE.json configs[4]: 62,500 utterances x 10 s @ 16 kHz, N=1024, hop 256
    (622 frames each), here with int16 input (20 GB in f32 would be 10 GB as s16) -> 38.9 M frames."""
    import torch
    n, hop, C, Ls = 1024, 256, 62_500, 160_000
    g = torch.Generator(device="cuda").manual_seed(0x5EED0005)
    x = torch.empty(C, Ls, device="cuda", dtype=torch.int16)
    for c0 in range(0, C, 12_500):
        x[c0:c0 + 12_500] = torch.randint(-32768, 32767, (12_500, Ls), device="cuda", generator=g, dtype=torch.int16)
    with S.Engine(fft_size=n, hop=hop, pcm="s16") as e:
        assert e.num_frames(Ls) == 622
        P = e.stft(x)
        torch.cuda.synchronize()
        assert P.shape == (C, 622, 512)
        _check_random_frames(S, P, x, n, hop, seed=5)
        assert bool(torch.isfinite(P[::997]).all())
    del P, x
    torch.cuda.empty_cache()


def test_config5_chunked_runner_f32(S, oracle):
    """BASELINE.json configs[4] through the chunked runner: float32 utterances (10 s at 16 kHz = 160,000 samples,
    622 frames each), H2D of chunk i+1 overlapping the transform of chunk i, against the oracle on sampled
    (utterance, frame) rows of every chunk, and bit-identical to the same utterances pushed through one plain call."""
    import torch
    from spectrogram_amd import chunked
    n, hop, utt_samples, n_utts, chunk_utts = 1024, 256, 160_000, 1100, 384
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0x5EED0005)
    host = (torch.rand(n_utts, utt_samples, generator=g) * 2 - 1).pin_memory()
    plan = chunked.plan_chunks(n_utts, 1, 0, chunk_utts)
    assert [c.utt_count for c in plan] == [384, 384, 332]
    cfg = oracle.OracleConfig.make(n, hop)
    picks = {}
    with S.Engine(fft_size=n, hop=hop) as e:
        r = chunked.ChunkedRunner(e, utt_samples, chunk_utts, dev)
        assert r.frames == 622

        def source(ch):
            return host[ch.utt_start:ch.utt_start + ch.utt_count]

        def sink(ch, rows):      # runs on the compute stream: pick rows before the next chunk overwrites them
            for u in (0, ch.utt_count // 2, ch.utt_count - 1):
                picks[ch.utt_start + u] = rows[u].clone()

        total = r.run(source, plan, sink)
        assert total == n_utts * 622
        whole = e.stft(host[plan[1].utt_start:plan[1].utt_start + 8].to(dev))
        torch.cuda.synchronize()
        mid = plan[1].utt_start
    assert len(picks) == 9
    for u, rows in picks.items():
        ref, _ = oracle.stft(cfg, host[u].numpy())
        assert_power_close(rows.cpu().numpy(), ref, "utterance %d" % u)
    assert torch.equal(picks[mid], whole[0])


def test_call_time_argument_errors_are_statuses_not_crashes(S):
    """the C ABI answers with a status and sg_last_error(), never a throw or a fault.  Every entry
    point with arguments the library can judge on the host: NULL handles and buffers, negative sizes, strides shorter
    than a row, polls and rings before sg_stream_begin / sg_ring_begin, out-of-range settings, buffers claimed to be
    page-locked that are not -- each refused with the documented code and a message, and the handle still works after."""
    import ctypes as C
    import torch
    from spectrogram_amd import _lib as L
    lib = L.load()
    x = np.zeros((2, 5000), np.float32)
    out = np.zeros((2, 16, 512), np.float32)
    by = np.zeros((2, 512), np.uint8)
    nul = None

    def refused(rc, code=L.ERR_INVALID):
        assert rc == code, (rc, lib.sg_last_error())
        assert lib.sg_last_error(), "an error status must leave a message"

    with S.Engine(fft_size=1024, hop=256) as e:
        h = e._h
        # ---- every entry point on a NULL handle
        for call in (lambda: lib.sg_stft(nul, x.ctypes.data, 2, 5000, 5000, 1, out.ctypes.data, 0, L.PTR_HOST, nul),
                     lambda: lib.sg_stream_begin(nul, 2), lambda: lib.sg_push(nul, x.ctypes.data, 10, 10, 0, nul),
                     lambda: lib.sg_poll_u8(nul, by.ctypes.data, 512, 512, 0, nul),
                     lambda: lib.sg_poll_f32(nul, out.ctypes.data, 512, 512, 0, nul),
                     lambda: lib.sg_poll_time_u8(nul, by.ctypes.data, 512, 512, 0, nul),
                     lambda: lib.sg_poll_time_f32(nul, out.ctypes.data, 512, 512, 0, nul),
                     lambda: lib.sg_ring_begin(nul, 4), lambda: lib.sg_ring_set_pinned(nul, 1),
                     lambda: lib.sg_ring_read(nul, by.ctypes.data, 0, 0, nul), lambda: lib.sg_set_smoothing(nul, 0.5),
                     lambda: lib.sg_set_db_range(nul, -90.0, -20.0), lambda: lib.sg_get_config(nul, C.byref(L.SgConfig())),
                     lambda: lib.sg_last_launch(nul, C.byref(L.SgLaunchInfo())), lambda: lib.sg_num_bins(nul),
                     lambda: lib.sg_out_elem_size(nul), lambda: lib.sg_num_frames(nul, 5000)):
            refused(call())
        assert lib.sg_destroy(nul) == 0                       # like free(NULL)
        # ---- batch call
        refused(lib.sg_stft(h, x.ctypes.data, -1, 5000, 5000, 1, out.ctypes.data, 0, L.PTR_HOST, nul))
        refused(lib.sg_stft(h, x.ctypes.data, 2, -5, 5000, 1, out.ctypes.data, 0, L.PTR_HOST, nul))
        refused(lib.sg_stft(h, x.ctypes.data, 2, 5000, 5000, 0, out.ctypes.data, 0, L.PTR_HOST, nul))
        refused(lib.sg_stft(h, x.ctypes.data, 2, 5000, -1, 1, out.ctypes.data, 0, L.PTR_HOST, nul))
        refused(lib.sg_stft(h, x.ctypes.data, 2, 5000, 5000, 1, out.ctypes.data, 100, L.PTR_HOST, nul))    # rows would overlap
        refused(lib.sg_stft(h, nul, 2, 5000, 5000, 1, out.ctypes.data, 0, L.PTR_HOST, nul))
        refused(lib.sg_stft(h, x.ctypes.data, 2, 5000, 5000, 1, nul, 0, L.PTR_HOST, nul))
        refused(lib.sg_stft(h, x.ctypes.data, 2, 5000, 5000, 1, out.ctypes.data, 0, L.IN_PINNED, nul))     # pageable, not pinned
        refused(lib.sg_stft(h, x.ctypes.data, 2, 5000, 5000, 1, out.ctypes.data, 0, L.OUT_PINNED, nul))
        assert lib.sg_stft(h, nul, 2, 100, 100, 1, nul, 0, L.PTR_HOST, nul) == 0       # shorter than a window: no frames, no buffers needed
        assert lib.sg_stft(h, nul, 0, 5000, 5000, 1, nul, 0, L.PTR_HOST, nul) == 0
        # ---- poll mode before / after sg_stream_begin
        for call in (lambda: lib.sg_push(h, x.ctypes.data, 10, 10, 0, nul), lambda: lib.sg_poll_u8(h, by.ctypes.data, 512, 512, 0, nul),
                     lambda: lib.sg_poll_time_f32(h, out.ctypes.data, 512, 512, 0, nul), lambda: lib.sg_ring_begin(h, 4)):
            refused(call(), L.ERR_STATE)
        refused(lib.sg_stream_begin(h, 0))
        assert lib.sg_stream_begin(h, 2) == 0
        refused(lib.sg_push(h, x.ctypes.data, -3, 10, 0, nul))
        refused(lib.sg_push(h, nul, 10, 10, 0, nul))
        refused(lib.sg_push(h, x.ctypes.data, 10, 5, 0, nul))                          # stride shorter than the block
        assert lib.sg_push(h, nul, 0, 0, 0, nul) == 0                                  # nothing to push
        refused(lib.sg_poll_u8(h, nul, 512, 512, 0, nul))
        refused(lib.sg_poll_u8(h, by.ctypes.data, -1, 512, 0, nul))
        refused(lib.sg_poll_u8(h, by.ctypes.data, 512, 100, 0, nul))
        refused(lib.sg_poll_u8(h, by.ctypes.data, 512, 512, L.OUT_PINNED, nul))
        refused(lib.sg_poll_time_u8(h, by.ctypes.data, 512, 100, 0, nul))
        refused(lib.sg_ring_begin(h, 0))
        refused(lib.sg_ring_begin(h, 1 << 20))
        refused(lib.sg_set_smoothing(h, 1.5))
        refused(lib.sg_set_smoothing(h, -0.1))
        refused(lib.sg_set_db_range(h, -30.0, -30.0))
        # ---- and the handle is still good: batch and poll results as if nothing had happened
        xr = np.random.default_rng(1).uniform(-1, 1, (2, 5000)).astype(np.float32)
        with S.Engine(fft_size=1024, hop=256) as fresh:
            want = fresh.stft_host(xr)
        assert np.array_equal(e.stft_host(xr), want)
        assert lib.sg_push(h, xr.ctypes.data, 2000, 5000, 0, nul) == 0
        assert lib.sg_poll_u8(h, by.ctypes.data, 512, 512, 0, nul) == 0 and by.max() > 0
    # ---- helpers without a handle
    refused(lib.sg_alloc_pinned(16, None))
    refused(lib.sg_wav_parse(None, 0, C.byref(L.SgWavInfo())))
    refused(lib.sg_axis_log_scale(0.0, 100.0, 20.0, 20000.0, None))
    torch.cuda.synchronize()


def test_handles_give_back_their_device_memory(S):
    """Every device allocation a handle makes (tables, staging, history, smoothing state, byte ring, |X| scratch, the
    log-frequency table) goes back at sg_destroy: 300 create / use / destroy cycles -- batch engines on host pointers
    with smoothing, analysers with a history ring, an fftSize re-plan in between -- leave the device's free memory
    where it was."""
    import torch
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (4, 40_000)).astype(np.float32)
    by = np.zeros((4, 512), np.uint8)

    def cycle(i):
        n = (512, 1024, 2048, 4096)[i % 4]
        with S.Engine(fft_size=n, hop=n // 4 + (i % 3), smoothing=0.5 if i % 2 else 0.0, out="db_u8" if i % 3 else "power",
                      logfreq_cols=0 if (i % 2) else 128) as e:
            e.stft_host(x)
        a = S.Analyser(fftSize=1024, n_channels=4, logfreq_cols=0)
        a.startHistory(64)
        a.push(x[:, :3000])
        a.getByteFrequencyData(by)
        if i % 5 == 0:
            a.fftSize = 2048                    # re-plan: new native engine, the old one destroyed
            a.push(x[:, :500])
        a.close()

    for i in range(20):                          # warm the allocator, the code objects, rocFFT-free as we are
        cycle(i)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(300):
        cycle(i)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 << 20, "device memory shrank by %.1f MiB over 300 handle lifetimes" % ((free0 - free1) / 2 ** 20)


def test_concurrent_host_threads_on_their_own_handles(S):
    """include/spectro.h: a handle is not thread-safe, the LIBRARY is -- one handle per thread.  Six host threads, each
    with its own engine (different sizes, load variants, formats, one an analyser in poll mode), hammer the library at
    once (ctypes drops the GIL inside every call): per-handle staging, the per-device LDS-cap cache, the thread-local
    error message and the lazily built launch tables must not bleed into one another.  Every result must equal what
    the same engine produces alone."""
    import threading
    import torch
    rng = np.random.default_rng(77)
    jobs = [dict(fft_size=1024, hop=256), dict(fft_size=2048, hop=735, out="db_u8", window="blackman", scale="inv_n"),
            dict(fft_size=8192, hop=2048), dict(fft_size=512, hop=100, out="db"), dict(fft_size=4096, hop=1000, pcm="s16"),
            dict(fft_size=16384, hop=4096, out="mag")]
    data, want = [], []
    for kw in jobs:
        n, hop = kw["fft_size"], kw["hop"]
        ns = n + 40 * hop + 3
        if kw.get("pcm") == "s16":
            x = rng.integers(-30000, 30000, (3, ns), dtype=np.int64).astype(np.int16)
        else:
            x = rng.uniform(-1, 1, (3, ns)).astype(np.float32)
        data.append(x)
        with S.Engine(**kw) as e:
            want.append(e.stft_host(x))
    errors = []

    def worker(i):
        try:
            with S.Engine(**jobs[i]) as e:
                xd = torch.from_numpy(data[i]).cuda()
                st = torch.cuda.Stream()
                for it in range(25):
                    got = e.stft_host(data[i]) if it % 2 else None          # host-pointer path (per-handle staging)
                    with torch.cuda.stream(st):
                        dev = e.stft(xd)                                    # device path on the thread's own stream
                    st.synchronize()
                    if got is not None and not np.array_equal(got, want[i]):
                        errors.append("thread %d iteration %d: host path differs" % (i, it))
                    if not np.array_equal(dev.cpu().numpy(), want[i]):
                        errors.append("thread %d iteration %d: device path differs" % (i, it))
                # an error raised on this thread carries this thread's message
                try:
                    S.Engine(fft_size=1000 + i)
                except S.SpectroError as ex:
                    if str(1000 + i) not in str(ex):
                        errors.append("thread %d: foreign error text %r" % (i, str(ex)))
        except Exception as ex:  # noqa: BLE001
            errors.append("thread %d: %r" % (i, ex))

    def poller():
        try:
            a = S.Analyser(fftSize=2048, n_channels=2)
            a.smoothingTimeConstant = 0
            ref = S.Analyser(fftSize=2048, n_channels=2)
            ref.smoothingTimeConstant = 0
            g, w = np.zeros((2, 1024), np.uint8), np.zeros((2, 1024), np.uint8)
            r = np.random.default_rng(5)
            for _ in range(60):
                blk = r.uniform(-1, 1, (2, 735)).astype(np.float32)
                a.push(blk)
                ref.push(blk)
                a.getByteFrequencyData(g)
                ref.getByteFrequencyData(w)
                if not np.array_equal(g, w):
                    errors.append("poller: two analysers fed the same samples disagree")
            a.close()
            ref.close()
        except Exception as ex:  # noqa: BLE001
            errors.append("poller: %r" % (ex,))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))] + [threading.Thread(target=poller)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_single_process_multi_device(S, oracle):
    """one host thread + handle + stream per device inside ONE process.  On a one-GPU box the
    device list repeats ordinal 0 (two concurrent workers on one card: same thread / device-guard logic); with
    more devices visible every ordinal gets a worker and handles with device_id > 0 are exercised."""
    import torch
    from spectrogram_amd.multidev import MultiDeviceEngine
    ndev = torch.cuda.device_count()
    devices = list(range(ndev)) if ndev > 1 else [0, 0, 0]
    n, hop = 1024, 256
    rng = np.random.default_rng(12)
    x = rng.uniform(-1, 1, (7, n + 33 * hop)).astype(np.float32)
    cfg = oracle.OracleConfig.make(n, hop)
    with MultiDeviceEngine(devices, fft_size=n, hop=hop) as me:
        assert [e.cfg.device_id for e in me.engines] == devices
        before = torch.cuda.current_device()
        parts = me.stft(torch.from_numpy(x))
        assert torch.cuda.current_device() == before          # the caller's device is restored
    seen = 0
    for i, (sh, rows) in enumerate(parts):
        if rows is None:
            continue
        assert rows.device.index == devices[i]
        for c in range(sh.count):
            ref, _ = oracle.stft(cfg, x[sh.start + c])
            assert_power_close(rows[c].cpu().numpy(), ref, "worker %d channel %d" % (i, sh.start + c))
        seen += sh.count
    assert seen == 7
    if ndev > 1:   # every visible device ran a worker of its own: handles with device_id > 0 are exercised, not skipped
        assert {rows.device.index for _, rows in parts if rows is not None} == set(range(min(ndev, 7)))
        assert max(e for e in devices) > 0
    if ndev > 1:   # a handle on the last device, driven while another device is current
        with S.Engine(fft_size=n, hop=hop, device_id=ndev - 1) as e:
            got = e.stft_host(x[0])
        assert_power_close(got, oracle.stft(cfg, x[0])[0], "device_id %d from device 0" % (ndev - 1))
    # sizes whose workgroup asks for more than 64 KiB of dynamic LDS (N >= 8192: 73,728 B, more with the
    # sample ring) raise the kernel's cap with hipFuncSetAttribute, which HIP keeps per device -- the cache of that
    # call is per (device, instantiation) now, and every worker of a multi-device engine must get its own grant.
    n8, hop8 = 8192, 2048
    x8 = rng.uniform(-1, 1, (len(devices) + 1, n8 + 9 * hop8)).astype(np.float32)
    cfg8 = oracle.OracleConfig.make(n8, hop8)
    for hop_used in (hop8, 3000):                      # register sliding window, and the LDS ring beside the buffers
        with MultiDeviceEngine(devices, fft_size=n8, hop=hop_used) as me:
            parts8 = me.stft(torch.from_numpy(x8))
        cfgu = oracle.OracleConfig.make(n8, hop_used)
        for sh, rows in parts8:
            if rows is None:
                continue
            ref, _ = oracle.stft(cfgu, x8[sh.start])
            assert_power_close(rows[0].cpu().numpy(), ref, "N=8192 hop %d on device %d" % (hop_used, rows.device.index))


def test_waveform_mode_as_the_reference_drives_it(S, oracle):
    """3D/visualizer.js:361-364: ANALYSISTYPE_WAVEFORM sets smoothingTimeConstant = 0.1 and polls
    getByteTimeDomainData into the frequencyBinCount-sized array (the first N/2 of the N newest samples)."""
    n = 2048
    rng = np.random.default_rng(8)
    x = np.clip(rng.normal(0, 0.6, n), -1.3, 1.3).astype(np.float32)     # includes clipped samples
    a = S.Analyser(fftSize=n)
    a.smoothingTimeConstant = 0.1
    a.push(x)
    buf = np.zeros(a.frequencyBinCount, np.uint8)
    assert a.getByteTimeDomainData(buf) is None
    assert np.array_equal(buf, oracle.time_domain_bytes(x[:n // 2]))
    # interleaved with a byte-frequency poll, as switching analysis type between draws does
    fb = np.zeros(a.frequencyBinCount, np.uint8)
    a.getByteFrequencyData(fb)
    a.getByteTimeDomainData(buf)
    assert np.array_equal(buf, oracle.time_domain_bytes(x[:n // 2]))
    # the float form of the same call (AnalyserNode.getFloatTimeDomainData): the samples themselves, bit for bit; a short
    # array takes the first len, a long one keeps its tail
    fl = np.full(n + 5, 7.0, np.float32)
    assert a.getFloatTimeDomainData(fl) is None
    assert np.array_equal(fl[:n], x) and np.all(fl[n:] == 7.0)
    sh = np.zeros(100, np.float32)
    a.getFloatTimeDomainData(sh)
    assert np.array_equal(sh, x[:100])
    a.push(x[:300] * np.float32(0.5))
    a.getFloatTimeDomainData(fl)
    assert np.array_equal(fl[:n], np.concatenate([x[300:], x[:300] * np.float32(0.5)]))
    a.close()


def test_bench_spawns_two_ranks(S):
    """`python bench.py --gpus 2` with no launcher: the parent (which never touches the GPU) starts two rank
    processes, they rendezvous on 127.0.0.1, each runs its own stream, rank 0 prints one line whose value is the
    sum over ranks / max-over-ranks time.  On this one-GPU box the ranks share the device (SG_BENCH_REHEARSE)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, SG_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--no-cpu-baseline", "--no-streaming"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    # the stdout line is the compact summary (BENCH_r04.json: a 37 KB line came back from the driver unparsed); the full
    # object it summarises is in the file it names
    assert len(lines[0]) <= 6000
    brief = json.loads(lines[0])
    j = json.load(open(os.path.join(ROOT, brief["detail"])))
    assert brief["value"] == j["value"] and brief["ms_per_step"] == j["ms_per_step"] and brief["n_gpus"] == 2
    assert len(brief["roofline"]["per_rank_frames_per_s"]) == 2 and brief["roofline"]["kernel"] == j["roofline"]["kernel"]
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["scaling"] == "weak"
    assert abs(j["value"] - 2 * 112497 * 4 / (j["ms_per_step"] * 4e-3)) <= 1e-6 * j["value"]
    assert j["value"] > 1e8 and "cpu_baseline" not in j and "streaming" not in j
    # value = frames of both ranks / the slowest rank's own (synchronize-then-stamp) time: with the two ranks sharing
    # this one device it is about twice what rank 0's HIP events saw for its own launches -- and no closing
    # collective is billed to it (tests/test_bench_contract.py pins the order of calls)
    ev = j["roofline"]["rank0_event_frames_per_s"]
    # every rank's own figures ride in the line (gathered after the timed region), so that a bent hardware scaling curve
    # can be attributed to a rank; the slowest rank's own interval is the one `value` is computed from
    rf = j["roofline"]
    assert len(rf["per_rank_frames_per_s"]) == 2 == len(rf["per_rank_ms_per_step"]) == len(rf["per_rank_event_ms_per_launch"])
    assert rf["per_rank_frames_per_s"][0] == ev
    assert abs(max(rf["per_rank_ms_per_step"]) - j["ms_per_step"]) <= 1e-6 * j["ms_per_step"]
    assert abs(rf["frac"] - rf["frac_wall"]) < 1e-12 and rf["frac_events"] > 0
    assert 0.4 * 2 * ev <= j["value"] <= 1.25 * 2 * ev, (j["value"], ev)


@pytest.mark.parametrize("workload", ["config2", "config5"])
def test_bench_four_ranks_rehearsal(S, workload):
    """The N-rank line of both workloads with four ranks sharing this one device over gloo (SG_BENCH_REHEARSE): the most a
    one-GPU box allows -- six processes may use the card at once and this test process is one of them; the eight-rank
    rendezvous / gathers / max-reduce run on the CPU (tests/test_shard.py::test_bench_collectives_gloo[8]).  value = the
    frames of all ranks / the slowest rank's own interval, one per-rank entry per rank, and the line says which device
    every rank sat on (here: one PCI address under four ranks, `distinct_devices` 1 -- on the driver's node it must be N)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, SG_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    extra = ["--steps", "4", "--warmup", "1"] if workload == "config2" else \
        ["--workload", "config5", "--chunk-utts", "256", "--steps", "3", "--warmup", "1", "--no-h2d"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline", "--no-streaming"] + extra,
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 6000, out.stdout
    brief = json.loads(lines[0])
    j = json.load(open(os.path.join(ROOT, brief["detail"])))
    assert brief["n_gpus"] == 4 == j["n_gpus"] and j["scaling"] == "weak" and brief["value"] == j["value"]
    rf = j["roofline"]
    assert len(rf["per_rank_frames_per_s"]) == 4 == len(rf["per_rank_ms_per_step"])
    frames_per_rank_step = 112497 if workload == "config2" else 256 * 622
    assert abs(j["value"] - 4 * frames_per_rank_step * j["steps"] / (j["ms_per_step"] * j["steps"] * 1e-3)) <= 1e-6 * j["value"]
    assert abs(max(rf["per_rank_ms_per_step"]) - j["ms_per_step"]) <= 1e-6 * j["ms_per_step"]   # the slowest rank's own interval
    rk = j["ranks"]
    assert rk["ranks_seen"] == 4 and len(rk["devices"]) == 4 and [d["rank"] for d in rk["devices"]] == [0, 1, 2, 3]
    assert rk["distinct_devices"] == 1 and len({d["pci"] for d in rk["devices"]}) == 1
    assert brief["ranks"]["ranks_seen"] == 4 and len(brief["ranks"]["devices"]) == 4


def test_randomized_layouts(S, oracle):
    """Seeded sweep over awkward but legal layouts: sample strides, padded channel strides, unaligned
    device bases (forces the generic load path), hop > N, exactly one frame, out_channel_stride with
    slack, every window / scale / output kind."""
    import os
    import torch
    # SG_FUZZ_SEED / SG_FUZZ_CASES: the same sweep from other seeds (tools/fuzz_layouts.sh runs a handful per round)
    rng = np.random.default_rng(int(os.environ.get("SG_FUZZ_SEED", "20261003")))
    kinds = ["power", "mag", "db", "db_u8"]
    for case in range(int(os.environ.get("SG_FUZZ_CASES", "48"))):
        n = int(2 ** rng.integers(5, 13))                       # 32 .. 4096
        hop = int(rng.choice([n // 4, n // 2, n, 2 * n + 3, max(1, n // 3), 1 + int(rng.integers(1, 3 * n))]))
        n_ch = int(rng.integers(1, 4))
        ss = int(rng.choice([1, 1, 2, 3]))
        frames = int(rng.integers(1, 40))
        n_samples = n + (frames - 1) * hop + int(rng.integers(0, hop))
        cs = n_samples * ss + int(rng.integers(0, 7)) if ss > 1 or rng.random() < 0.5 else n_samples
        if ss == 2 and n_ch == 2 and rng.random() < 0.5:
            cs = 1                                               # true interleaved stereo
        base_off = int(rng.integers(0, 4))                       # misaligns the device pointer by 0..3 floats
        window = ["hann", "blackman", "rect"][int(rng.integers(0, 3))]
        scale = ["none", "inv_n"][int(rng.integers(0, 2))]
        out = kinds[int(rng.integers(0, 4))]
        nyq = bool(rng.integers(0, 2))
        extent = (n_ch - 1) * cs + (n_samples - 1) * ss + 1
        buf = rng.uniform(-1, 1, base_off + extent + 8).astype(np.float32)
        dbuf = torch.from_numpy(buf).cuda()
        what = "case %d: n=%d hop=%d ch=%d ss=%d cs=%d off=%d %s %s %s nyq=%d" % (
            case, n, hop, n_ch, ss, cs, base_off, window, scale, out, nyq)
        forced = int(rng.choice([0, 0, 16, 40]))                 # frames per run: the launcher's choice, or forced long
        what += " run_len=%d" % forced
        with S.Engine(debug={"run_len": forced}, fft_size=n, hop=hop, window=window, scale=scale, out=out, nyquist=nyq) as e:
            nf = e.num_frames(n_samples)
            bins = e.bins
            assert nf == frames, what
            slack = int(rng.integers(0, 5)) * bins
            ocs = nf * bins + slack
            dt = torch.uint8 if out == "db_u8" else torch.float32
            dout = torch.full((n_ch * ocs + 16,), 77, dtype=dt, device="cuda")
            e.stft_device_ptr(dbuf.data_ptr() + 4 * base_off, n_ch, n_samples, cs, ss, dout.data_ptr(), ocs,
                              torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got_all = dout.cpu().numpy()
        okind = kinds.index(out)
        cfg = oracle.OracleConfig.make(n, hop, window=["hann", "blackman", "rect"].index(window),
                                       scale=["none", "inv_n"].index(scale), out_kind=okind, nyquist=int(nyq))
        for c in range(n_ch):
            xc = buf[base_off + c * cs: base_off + c * cs + (n_samples - 1) * ss + 1]
            ref, ref_u8 = oracle.stft(cfg, xc, sample_stride=ss)
            got = got_all[c * ocs: c * ocs + nf * bins].reshape(nf, -1)
            if out == "power":
                assert_power_close(got, ref, what)
            elif out == "mag":
                assert_power_close(got.astype(np.float64) ** 2, ref ** 2, what)
            elif out == "db":
                assert_power_close(10.0 ** (got.astype(np.float64) / 10), 10.0 ** (ref / 10), what)
            else:
                cfg.out_kind = 0
                P, _ = oracle.stft(cfg, xc, sample_stride=ss)
                cfg.out_kind = okind
                assert_bytes_in_bounds(got, *byte_bounds(P, floor_db=-200.0), what, ref_u8)
            # nothing outside the rows this channel owns was touched
            tail = got_all[c * ocs + nf * bins: (c + 1) * ocs] if c + 1 < n_ch else got_all[c * ocs + nf * bins:]
            assert np.all(tail == 77), what


def test_randomized_formats_strides_and_epilogues(S, oracle):
    """The second seeded sweep: sample format (float32 / int16 / packed int24) x K-channel down-mix x sample stride x
    padded rows x misaligned bases x hops on every load variant (register window at N/4 and N/2, the LDS ring up to N - 4,
    generic beyond) x N = 32 ... 32768 x output kind, a third of the cases through the log-frequency columns -- each
    against the float64 oracle fed the same samples as floats.  SG_FUZZ_SEED / SG_FUZZ_CASES as above."""
    import os
    import torch
    rng = np.random.default_rng(int(os.environ.get("SG_FUZZ_SEED", "20261004")) + 7)
    seen = set()
    for case in range(int(os.environ.get("SG_FUZZ_CASES", "48"))):
        fmt = ["f32", "s16", "s24"][int(rng.integers(0, 3))]
        n = int(2 ** rng.integers(5, 16))                        # 32 .. 32768
        hop = int(rng.choice([n // 4, n // 2, max(1, n // 3), n - 4, n - 3, 1 + int(rng.integers(0, n)), 1 + int(rng.integers(0, n // 2)), n + 5]))
        k = int(rng.choice([1, 1, 1, 2, 3]))
        ss = int(rng.choice([1, 1, 1, 2, 3]))
        n_ch = int(rng.integers(1, 5))
        frames = int(rng.integers(1, 30)) if n <= 2048 else int(rng.integers(1, 12 if n <= 8192 else 5))
        n_samples = n + (frames - 1) * hop + int(rng.integers(0, hop))       # analysed (down-mixed) samples
        row = n_samples * k * ss                                             # input elements one channel spans
        cs = row + int(rng.integers(0, 9))
        off = int(rng.integers(0, 6))
        out = ["power", "db_u8", "mag", "db"][int(rng.integers(0, 4))]
        cols = int(rng.choice([0, 0, 37, 256])) if 512 <= n <= 8192 else 0
        total = off + n_ch * cs + 4
        if fmt == "f32":
            raw = rng.uniform(-1, 1, total).astype(np.float32)
            as_f, dev = raw, torch.from_numpy(raw).cuda()
            ptr, esz = dev.data_ptr() + 4 * off, 4
        elif fmt == "s16":
            raw = rng.integers(-32768, 32768, total, dtype=np.int64).astype(np.int16)
            as_f, dev = raw.astype(np.float32) / np.float32(32768.0), torch.from_numpy(raw).cuda()
            ptr, esz = dev.data_ptr() + 2 * off, 2
        else:
            q = rng.integers(-(1 << 23), 1 << 23, total, dtype=np.int64)
            u = q & 0xFFFFFF
            raw = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=-1).astype(np.uint8).reshape(-1)
            as_f, dev = (q / 8388608.0).astype(np.float32), torch.from_numpy(raw).cuda()
            ptr, esz = dev.data_ptr() + 3 * off, 3
        what = "case %d: %s n=%d hop=%d k=%d ss=%d ch=%d cs=%d off=%d %s cols=%d" % (case, fmt, n, hop, k, ss, n_ch, cs, off, out, cols)
        kw = dict(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out, pcm=fmt, downmix=k)
        if cols:
            kw["logfreq_cols"] = cols
        # frames per run: the launcher's choice (small launches: 4 frames and up), or forced long, so that register
        # windows rotate and LDS rings wrap many times within a run
        forced = int(rng.choice([0, 0, 16, 33, 64]))
        what += " run_len=%d" % forced
        with S.Engine(debug={"run_len": forced}, **kw) as e:
            assert e.num_frames(n_samples) == frames, what
            bins = e.bins
            dt = torch.uint8 if out == "db_u8" else torch.float32
            dout = torch.full((n_ch * frames * bins + 8,), 77, dtype=dt, device="cuda")
            e.stft_device_ptr(ptr, n_ch, n_samples, cs, ss, dout.data_ptr(), frames * bins, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            seen.add(e.last_launch()[0])
            got_all = dout.cpu().numpy()
        assert np.all(got_all[n_ch * frames * bins:] == 77), what
        okind = ["power", "mag", "db", "db_u8"].index(out)
        cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1, out_kind=okind)
        for c in range(n_ch):
            xs = as_f[off + c * cs: off + c * cs + row: ss]
            mono = xs.reshape(n_samples, k).astype(np.float64).mean(axis=1).astype(np.float32)
            ref, ref_u8 = oracle.stft(cfg, mono)
            got = got_all[c * frames * bins: (c + 1) * frames * bins].reshape(frames, bins)
            if cols:
                src = ref_u8.astype(np.float64) if out == "db_u8" else (np.maximum(ref, -200.0) if out == "db" else ref)
                exp = np.stack([oracle.logfreq_resample(r, cols) for r in src])
                if out == "db_u8":
                    P, _ = oracle.stft(oracle.OracleConfig.make(n, hop, window=1, scale=1), mono)
                    assert_bytes_in_bounds(got, *logfreq_byte_bounds(*byte_bounds(P, floor_db=-200.0), cols), what,
                                           np.floor(exp + 0.5))
                elif out == "db":
                    m = exp > -150
                    assert np.abs(got[m] - exp[m]).max() <= 5e-2, what
                else:
                    assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max() + 1e-30, what
            elif out == "power":
                assert_power_close(got, ref, what)
            elif out == "mag":
                assert_power_close(got.astype(np.float64) ** 2, ref ** 2, what)
            elif out == "db":
                assert_power_close(10.0 ** (got.astype(np.float64) / 10), 10.0 ** (ref / 10), what)
            else:
                P, _ = oracle.stft(oracle.OracleConfig.make(n, hop, window=1, scale=1), mono)
                assert_bytes_in_bounds(got, *byte_bounds(P, floor_db=-200.0), what, ref_u8)
    if int(os.environ.get("SG_FUZZ_CASES", "48")) >= 48 and "SG_FUZZ_SEED" not in os.environ:   # coverage of the default draw only: another seed may miss a variant
        assert len({v & 7 for v in seen}) >= 3 and len({v >> 3 for v in seen}) == 3, sorted(seen)   # load variants and formats reached


@pytest.mark.parametrize("k,fmt", [(2, "f32"), (2, "s16"), (6, "f32")])
def test_downmix_to_mono(S, oracle, k, fmt):
    """AnalyserNode down-mixes a multi-channel input to mono before windowing (stereo -> 0.5 (L + R));
    here fused into the framing load for K interleaved channels."""
    n, hop = 2048, 512
    rng = np.random.default_rng(100 + k)
    nsamp = n + 25 * hop + 7
    if fmt == "s16":
        inter = rng.integers(-20000, 20000, (2, nsamp * k), dtype=np.int16)
        as_f = inter.astype(np.float32) / 32768.0
    else:
        inter = rng.uniform(-1, 1, (2, nsamp * k)).astype(np.float32)
        as_f = inter
    with S.Engine(fft_size=n, hop=hop, downmix=k, pcm=fmt) as e:
        got = e.stft_host(inter)
    assert got.shape[1] == oracle.num_frames(nsamp, n, hop)
    cfg = oracle.OracleConfig.make(n, hop)
    for c in range(2):
        mono = as_f[c].reshape(nsamp, k).astype(np.float64).mean(axis=1).astype(np.float32)
        ref, _ = oracle.stft(cfg, mono)
        assert_power_close(got[c], ref, "downmix k=%d %s ch=%d" % (k, fmt, c))


@pytest.mark.parametrize("k", [2, 6])
def test_downmix_device_path(S, oracle, k):
    """Engine.stft (torch device buffers) with a K-channel down-mix: n_samples handed to the C ABI counts
    analysed samples, so the kernel reads exactly the tensor's width (it used to read K times past it)."""
    import torch
    n, hop = 1024, 256
    rng = np.random.default_rng(200 + k)
    nsamp = n + 19 * hop + 3
    inter = rng.uniform(-1, 1, (3, nsamp * k)).astype(np.float32)
    with S.Engine(fft_size=n, hop=hop, downmix=k) as e:
        out = e.stft(torch.from_numpy(inter).cuda())
        torch.cuda.synchronize()
        got = out.cpu().numpy()
    assert got.shape == (3, oracle.num_frames(nsamp, n, hop), n // 2)
    cfg = oracle.OracleConfig.make(n, hop)
    for c in range(3):
        mono = inter[c].reshape(nsamp, k).astype(np.float64).mean(axis=1).astype(np.float32)
        ref, _ = oracle.stft(cfg, mono)
        assert_power_close(got[c], ref, "device downmix k=%d ch=%d" % (k, c))


def test_host_path_padded_output_keeps_gaps(S, oracle):
    """sg_stft with host pointers and out_channel_stride > frames*bins writes only the rows: the caller's
    bytes between channels keep their sentinel (a flat copy-back used to clobber them)."""
    import ctypes as C
    from spectrogram_amd import _lib as L
    n, hop, n_ch = 256, 128, 3
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (n_ch, n + 9 * hop)).astype(np.float32)
    with S.Engine(fft_size=n, hop=hop) as e:
        frames, bins = e.num_frames(x.shape[1]), e.bins
        stride = frames * bins + 37
        out = np.full(n_ch * stride, -7.25, np.float32)
        L.check(e._lib.sg_stft(e._live(), x.ctypes.data, n_ch, x.shape[1], x.shape[1], 1, out.ctypes.data,
                               stride, L.PTR_HOST, None))
    cfg = oracle.OracleConfig.make(n, hop)
    for c in range(n_ch):
        ref, _ = oracle.stft(cfg, x[c])
        assert_power_close(out[c * stride: c * stride + frames * bins].reshape(frames, bins), ref, "padded ch %d" % c)
        assert np.all(out[c * stride + frames * bins: (c + 1) * stride] == -7.25), "gap after channel %d was written" % c


def test_single_round_geometry_is_bit_identical(S, oracle):
    """A launch the chip holds at once is cut into one exact round of equal runs with the waves ranked by remaining
    work (spectro_api.hip); the "no_single_round" switch (sg_debug_set) keeps the 16-64-frame runs.  How a channel's
    frames are cut into runs must not change a single bit of the result.  The two arms really differ (sg_last_launch:
    28- / 30- / 35-frame runs with the single-round flag against 16-frame runs without), the first case is the bench
    launch itself (BASELINE.json configs[1]: 112,497 frames), and the frames either side of every 64th run boundary of
    the single-round geometry are held to the float64 oracle."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(77)
    cases = [(1024, 256, 1, 28_800_000, "power", 28), (1024, 256, 4, 1024 + 29_999 * 256, "db_u8", 30),
             (2048, 512, 2, 2048 + 34_999 * 512, "power", 35), (1024, 512, 2, 1024 + 49_999 * 512, "db", 25)]
    for n, hop, ch, ns, out, want_rl in cases:
        x = torch.rand(ch, ns, device="cuda", generator=g) * 2 - 1
        with S.Engine(fft_size=n, hop=hop, out=out) as e:
            a = e.stft(x)
            va, rla, single_a, _ = e.last_launch()
        with S.Engine(debug={"no_single_round": 1}, fft_size=n, hop=hop, out=out) as e:
            b = e.stft(x)
            vb, rlb, single_b, _ = e.last_launch()
        torch.cuda.synchronize()
        assert (rla, single_a) == (want_rl, 1) and single_b == 0 and rlb != rla and va == vb, (n, hop, ch, rla, rlb, single_a, single_b)
        assert torch.equal(a, b), (n, hop, ch, out)
        if out == "power":
            frames = (ns - n) // hop + 1
            cfg = oracle.OracleConfig.make(n, hop)
            xc = x[ch - 1].cpu().numpy()
            got = a[ch - 1]
            pick = sorted({0, frames - 1} | {k * rla + d for k in range(64, frames // rla + 1, 64) for d in (-1, 0)})
            for f in pick:
                ref, _ = oracle.stft(cfg, xc[f * hop:f * hop + n])
                assert_power_close(got[f:f + 1].cpu().numpy(), ref, "n=%d frame %d (run boundary of %d-frame runs)" % (n, f, rla))


@pytest.mark.parametrize("align", ["vertex", "centre"])
@pytest.mark.parametrize("tau", [0.0, 0.75])
def test_poll_mode_logfreq_rows(S, oracle, tau, align):
    """Poll mode with cfg.logfreq_cols: the row a frequency poll hands out (and the history ring keeps) is the 256
    log-frequency columns the reference's shader samples out of the 1024-byte row (sonogram-vertex.shader:51-52 over
    3D/visualizer.js:399-417) -- one fused launch, 256 bytes per channel per hop instead of 1024."""
    n, hop, cols = 2048, 735, 256
    rng = np.random.default_rng(int(tau * 100) + 1)
    x = (rng.uniform(-1, 1, (2, n + 12 * hop)) * np.array([[0.3], [0.01]])).astype(np.float32)
    a = S.Analyser(fftSize=n, n_channels=2, logfreq_cols=cols, logfreq_align=align)
    if align == "vertex":   # the Analyser objects default to the reference's default display, the 3-D sonogram's mesh
        a.close()
        a = S.Analyser(fftSize=n, n_channels=2, logfreq_cols=cols)
    a.smoothingTimeConstant = tau
    assert (a.frequencyBinCount, a.rowLength) == (n // 2, cols)
    a.startHistory(8)
    st = [np.zeros(n // 2), np.zeros(n // 2)]
    bb = [ByteBounds(n // 2), ByteBounds(n // 2)]
    by = np.zeros((2, cols), np.uint8)
    fl = np.zeros((2, cols), np.float32)
    rows = []
    for f in range(13):
        a.push(x[:, :n] if f == 0 else x[:, n + (f - 1) * hop:n + f * hop])
        a.getByteFrequencyData(by)           # transform + smoothing + columns: one launch
        a.getFloatFrequencyData(fl)          # same block again: re-emitted from the state, columns included
        by2 = np.zeros_like(by)
        a.getByteFrequencyData(by2)
        assert np.array_equal(by, by2)
        rows.append(by.copy())
        for c in range(2):
            db, ref_b = oracle.analyser_poll(n, x[c, f * hop:f * hop + n], tau, -100.0, -30.0, st[c])
            exp_b = oracle.logfreq_resample(ref_b.astype(np.float64), cols, align)
            lo, hi = bb[c].step(poll_power(oracle, x[c, f * hop:f * hop + n]), tau)
            assert_bytes_in_bounds(by[c], *logfreq_byte_bounds(lo, hi, cols, align), "poll %d channel %d" % (f, c), np.floor(exp_b + 0.5))
            exp_f = oracle.logfreq_resample(np.maximum(db, -200.0), cols, align)
            m = exp_f > -110
            assert np.abs(fl[c][m] - exp_f[m]).max() < 5e-2, (f, c)
    # short / long caller arrays: min(len, row) values, the rest untouched
    short = np.full((2, 10), 7, np.uint8)
    a.getByteFrequencyData(short)
    assert np.array_equal(short, by[:, :10])
    long_ = np.full((2, 300), 7, np.uint8)
    a.getByteFrequencyData(long_)
    assert np.array_equal(long_[:, :cols], by) and np.all(long_[:, cols:] == 7)
    # the ring holds rows of `cols` bytes: the last 8 byte polls that advanced it
    hist = a.readHistory()
    assert hist.shape == (2, 8, cols)
    a.close()


@pytest.mark.parametrize("k", [2, 3])
def test_poll_mode_downmix_on_push(S, oracle, k):
    """cfg.downmix = K in poll mode: sg_push averages K interleaved channels into the analyser's mono history, as the
    reference's `mix` GainNode does with a stereo file or microphone (UI/player.js:25,110-115,125-131).  The bytes are
    those of a mono analyser fed the float32 mean -- bit for bit -- through host pushes, device pushes, short blocks
    and blocks longer than the window."""
    import ctypes as C
    import torch
    from spectrogram_amd import _lib as L
    n = 1024
    rng = np.random.default_rng(40 + k)
    blocks = [300, 1024, 5, 2500, 735, 735, 4096, 128]
    multi = [rng.uniform(-1, 1, (2, b * k)).astype(np.float32) for b in blocks]
    a = S.Analyser(fftSize=n, n_channels=2, downmix=k)
    ref = S.Analyser(fftSize=n, n_channels=2)
    for an in (a, ref):
        an.smoothingTimeConstant = 0.5
    got, want = np.zeros((2, n // 2), np.uint8), np.zeros((2, n // 2), np.uint8)
    st = np.zeros(n // 2)
    bb = ByteBounds(n // 2)
    hist = np.zeros(n, np.float32)
    lib = L.load()
    for i, blk in enumerate(multi):
        mono = np.zeros((2, blocks[i]), np.float32)
        for c in range(k):                              # the kernel's order of additions, then the 1/K
            mono += blk[:, c::k]
        mono *= np.float32(1.0 / k)
        if i % 2:
            d = torch.from_numpy(blk).cuda()            # device-pointer push
            L.check(lib.sg_push(a._h, d.data_ptr(), blocks[i], blocks[i] * k, L.IN_DEVICE, None))
            torch.cuda.synchronize()
        else:
            a.push(blk)
        ref.push(mono)
        a.getByteFrequencyData(got)
        ref.getByteFrequencyData(want)
        assert np.array_equal(got, want), "block %d (%d samples)" % (i, blocks[i])
        hist = np.concatenate([hist, mono[1]])[-n:]
        _, ref_b = oracle.analyser_poll(n, hist, 0.5, -100.0, -30.0, st)
        assert_bytes_in_bounds(got[1], *bb.step(poll_power(oracle, hist), 0.5), "block %d" % i, ref_b)
    a.close()
    ref.close()


def test_randomized_poll_sequences(S, oracle):
    """Seeded sweep over poll mode: fftSize 32 ... 8192, 1-3 analysers, smoothing 0 / 0.3 / 0.8, dB range, int16, packed int24
    (round 6) or float pushes, a K-channel down-mix, push lengths from 1 sample to beyond the window (the history ring wraps at every position), polls
    of every kind in random order, repeated polls without a push (the smoother must not advance), the byte history ring
    with wrap-around -- every poll against the float64 oracle on a host-side model of the last N samples."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SG_FUZZ_SEED", "20261004")) + 99)
    for case in range(int(os.environ.get("SG_FUZZ_CASES", "48")) // 4):
        n = int(2 ** rng.integers(5, 14))
        n_ch = int(rng.integers(1, 4))
        tau = float(rng.choice([0.0, 0.3, 0.8]))
        k = int(rng.choice([1, 1, 2, 3]))
        pcm = ["f32", "s16", "s24"][int(rng.integers(0, 3))]
        min_db, max_db = (-100.0, -30.0) if rng.random() < 0.5 else (-80.0, -10.0)
        ring_h = int(rng.integers(2, 6))
        what = "case %d: n=%d ch=%d tau=%.1f k=%d %s [%g, %g] ring %d" % (case, n, n_ch, tau, k, pcm, min_db, max_db, ring_h)
        a = S.Analyser(fftSize=n, n_channels=n_ch, pcm=pcm, downmix=k)
        a.smoothingTimeConstant = tau
        a.maxDecibels = max_db                                    # upwards first: min < max must hold at every step
        a.minDecibels = min_db
        a.startHistory(ring_h)
        m = n // 2
        st = [np.zeros(m) for _ in range(n_ch)]
        bb = [ByteBounds(m, min_db, max_db) for _ in range(n_ch)]
        hist = np.zeros((n_ch, n), np.float32)
        rows = []                                                 # byte rows in poll order (the ring's content)
        by, fl = np.zeros((n_ch, m), np.uint8), np.zeros((n_ch, m), np.float32)
        td, tf = np.zeros((n_ch, n), np.uint8), np.zeros((n_ch, n), np.float32)
        for step in range(10):
            blk = int(rng.choice([1, int(rng.integers(1, n)), int(rng.integers(1, n)), n, n + int(rng.integers(1, 2 * n))]))
            if pcm == "s16":
                q = rng.integers(-30000, 30000, (n_ch, blk * k), dtype=np.int64).astype(np.int16)
            elif pcm == "s24":
                q = rng.integers(-(1 << 23), 1 << 23, (n_ch, blk * k), dtype=np.int64)
            else:
                q = (rng.uniform(-1, 1, (n_ch, blk * k)) * rng.choice([1.0, 1e-3])).astype(np.float32)
            a.push(_pack24(q) if pcm == "s24" else q)
            # the push kernel's own float32 arithmetic (time-domain polls are compared bit for bit): the K samples added
            # in order -- int16 as the integers they are -- then one multiply by float32(1/K); the 1/32768 afterwards
            acc = np.zeros((n_ch, blk), np.float32)
            src = q.astype(np.float32).reshape(n_ch, blk, k)
            for j in range(k):
                acc = acc + src[:, :, j]
            mono = acc * np.float32(1.0 / k) if k > 1 else acc
            if pcm == "s16":
                mono = mono * np.float32(1.0 / 32768.0)
            elif pcm == "s24":
                mono = mono * np.float32(1.0 / 8388608.0)
            hist = np.concatenate([hist, mono], axis=1)[:, -n:]
            ref = [oracle.analyser_poll(n, hist[c], tau, min_db, max_db, st[c]) for c in range(n_ch)]   # advances st once
            bounds = [bb[c].step(poll_power(oracle, hist[c]), tau) for c in range(n_ch)]                  # ... and so do these
            for kind in rng.permutation(["u8", "f32", "u8", "td", "tf"]):
                if kind == "u8":
                    a.getByteFrequencyData(by)
                    rows.append(by.copy())
                    for c in range(n_ch):
                        assert_bytes_in_bounds(by[c], *bounds[c], "%s step %d channel %d" % (what, step, c), ref[c][1])
                elif kind == "f32":
                    a.getFloatFrequencyData(fl)
                    for c in range(n_ch):
                        db = ref[c][0]
                        mk = db > -120
                        assert not mk.any() or np.abs(fl[c][mk] - db[mk]).max() < 2e-2, (what, step, c)
                        assert np.all(fl[c][~mk] < -119), (what, step, c)
                elif kind == "td":
                    a.getByteTimeDomainData(td)
                    for c in range(n_ch):
                        assert np.array_equal(td[c], oracle.time_domain_bytes(hist[c])), (what, step, c)
                else:
                    a.getFloatTimeDomainData(tf)
                    assert np.array_equal(tf, hist), (what, step)
        # the ring holds the last ring_h byte rows, oldest first once unrolled
        img = a.readHistory()
        want = np.stack(rows[-ring_h:], axis=1) if len(rows) >= ring_h else None   # [channels, ring_h, bins]
        if want is not None:
            assert np.array_equal(np.asarray(img).reshape(n_ch, ring_h, m), want), what
        a.close()


def test_poll_mode_int16_pushes(S, oracle):
    """cfg.pcm_format = S16 in poll mode: sg_push takes int16 samples (a 16-bit capture device; half the PCIe bytes),
    converts them on the way into the float history, the 1/32768 rides in the window; frequency bytes, float dB and
    time-domain bytes against the oracle on value / 32768, with a stereo down-mix on top."""
    n = 2048
    rng = np.random.default_rng(1616)
    a = S.Analyser(fftSize=n, n_channels=2, pcm="s16", downmix=2)
    a.smoothingTimeConstant = 0.75
    st = [np.zeros(n // 2), np.zeros(n // 2)]
    bb = [ByteBounds(n // 2), ByteBounds(n // 2)]
    hist = np.zeros((2, n), np.float32)
    by, fl, td = np.zeros((2, n // 2), np.uint8), np.zeros((2, n // 2), np.float32), np.zeros((2, n), np.uint8)
    tf = np.zeros((2, n), np.float32)
    for blk in (735, 4000, 64, 735):
        q = rng.integers(-20000, 20000, (2, blk * 2), dtype=np.int16)
        a.push(q)
        mono = ((q[:, 0::2].astype(np.float32) + q[:, 1::2].astype(np.float32)) * np.float32(0.5)) / np.float32(32768.0)
        hist = np.concatenate([hist, mono], axis=1)[:, -n:]
        a.getByteFrequencyData(by)
        a.getFloatFrequencyData(fl)
        a.getByteTimeDomainData(td)
        for c in range(2):
            db, ref_b = oracle.analyser_poll(n, hist[c], 0.75, -100.0, -30.0, st[c])
            assert_bytes_in_bounds(by[c], *bb[c].step(poll_power(oracle, hist[c]), 0.75), "block %d channel %d" % (blk, c), ref_b)
            m = db > -120
            assert np.abs(fl[c][m] - db[m]).max() < 2e-2
            assert np.array_equal(td[c], oracle.time_domain_bytes(hist[c]))
        a.getFloatTimeDomainData(tf)
        assert np.array_equal(tf, hist)       # the mean of two int16 values over 32768 is exact in float32
    a.close()


def test_pinned_host_buffers_are_read_and_written_in_place(S, oracle):
    """SG_IN_PINNED / SG_OUT_PINNED: page-locked host memory is mapped into the device's address space, so sg_push reads
    the new samples with a kernel straight over PCIe and the fused poll kernel stores its rows straight into the
    caller's buffer -- no staging copy either way (config 4's per-hop path).  Same bytes as pageable pushes and
    device-side polls, for float32 and int16 pushes, bins and log-frequency rows, odd block lengths included."""
    import ctypes as C
    import torch
    from spectrogram_amd import _lib as L
    lib = L.load()
    n, n_ch = 1024, 5
    rng = np.random.default_rng(808)
    for pcm, cols in (("f32", 0), ("s16", 0), ("f32", 64), ("s16", 64)):
        row = cols or n // 2
        a = S.Analyser(fftSize=n, n_channels=n_ch, pcm=pcm, logfreq_cols=cols)     # pinned in, pinned out
        b = S.Analyser(fftSize=n, n_channels=n_ch, pcm=pcm, logfreq_cols=cols)     # pageable in, host out (staged)
        for an in (a, b):
            an.smoothingTimeConstant = 0.6
        out_pin = torch.zeros(n_ch, row, dtype=torch.uint8).pin_memory()
        flt_pin = torch.zeros(n_ch, row, dtype=torch.float32).pin_memory()
        want, wantf = np.zeros((n_ch, row), np.uint8), np.zeros((n_ch, row), np.float32)
        for blk in (256, 735, 1024, 3001, 4, 1000):
            x = rng.uniform(-1, 1, (n_ch, blk))
            x = (x * 20000).astype(np.int16) if pcm == "s16" else x.astype(np.float32)
            xp = torch.from_numpy(x).pin_memory()
            L.check(lib.sg_push(a._h, xp.data_ptr(), blk, blk, L.IN_PINNED, None))
            b.push(x)
            L.check(lib.sg_poll_u8(a._h, out_pin.data_ptr(), row, row, L.OUT_PINNED, None))
            L.check(lib.sg_poll_f32(a._h, flt_pin.data_ptr(), row, row, L.OUT_PINNED, None))
            torch.cuda.synchronize()
            b.getByteFrequencyData(want)
            b.getFloatFrequencyData(wantf)
            assert np.array_equal(out_pin.numpy(), want), (pcm, cols, blk)
            assert np.array_equal(flt_pin.numpy(), wantf), (pcm, cols, blk)
        a.close()
        b.close()
    # a pageable buffer passed as SG_OUT_PINNED is refused, not written through a bad pointer
    a = S.Analyser(fftSize=n)
    a.push(np.zeros(n, np.float32))
    plain = np.zeros(n // 2, np.uint8)
    rc = lib.sg_poll_u8(a._h, plain.ctypes.data, n // 2, n // 2, L.OUT_PINNED, None)
    assert rc == L.ERR_INVALID and b"page-locked" in lib.sg_last_error()
    a.close()


def test_randomized_host_pointer_paths(S):
    """Seeded sweep over how sg_stft is handed its buffers: pageable host memory (staged both ways), page-locked memory
    read / written in place (SG_IN_PINNED / SG_OUT_PINNED), device memory, and every mix of the three -- with padded
    input rows, a sample stride, misaligned bases, slack between the output channels, all three sample formats, with
    and without batch smoothing.  Each combination must produce the bits of the all-device call (itself held to the
    oracle by the other sweeps) and leave every byte outside its rows alone."""
    import os
    import torch
    from spectrogram_amd import _lib as L
    rng = np.random.default_rng(int(os.environ.get("SG_FUZZ_SEED", "20261004")) + 63)
    for case in range(max(6, int(os.environ.get("SG_FUZZ_CASES", "48")) // 3)):
        fmt = ["f32", "s16", "s24"][int(rng.integers(0, 3))]
        n = int(2 ** rng.integers(5, 13))
        hop = int(rng.choice([n // 4, n // 2, 1 + int(rng.integers(0, n)), n + 3]))
        n_ch, ss = int(rng.integers(1, 4)), int(rng.choice([1, 1, 2]))
        frames = int(rng.integers(1, 25))
        n_samples = n + (frames - 1) * hop + int(rng.integers(0, hop))
        row = n_samples * ss
        cs = row + int(rng.integers(0, 6))
        off = int(rng.integers(0, 4))
        out = ["power", "db_u8", "db"][int(rng.integers(0, 3))]
        tau = float(rng.choice([0.0, 0.0, 0.6]))
        esz_in = {"f32": 4, "s16": 2, "s24": 3}[fmt]
        total = off + n_ch * cs + 2
        if fmt == "f32":
            raw = rng.uniform(-1, 1, total).astype(np.float32).view(np.uint8)
        elif fmt == "s16":
            raw = rng.integers(-32768, 32768, total, dtype=np.int64).astype(np.int16).view(np.uint8)
        else:
            raw = rng.integers(0, 256, total * 3, dtype=np.int64).astype(np.uint8)
        raw = np.ascontiguousarray(raw)
        what = "case %d: %s n=%d hop=%d ch=%d ss=%d cs=%d off=%d %s tau=%.1f" % (case, fmt, n, hop, n_ch, ss, cs, off, out, tau)
        with S.Engine(fft_size=n, hop=hop, out=out, pcm=fmt, smoothing=tau, window="blackman", scale="inv_n") as e:
            bins, lib = e.bins, e._lib
            osz = 1 if out == "db_u8" else 4
            ocs = frames * bins + int(rng.integers(0, 3)) * 5
            nbytes_out = (n_ch * ocs + 3) * osz
            in_dev = torch.from_numpy(raw).cuda()
            in_pin = torch.from_numpy(raw).pin_memory()
            stream = torch.cuda.current_stream().cuda_stream

            def call(in_kind, out_kind):
                src = {"host": raw.ctypes.data, "pin": in_pin.data_ptr(), "dev": in_dev.data_ptr()}[in_kind] + off * esz_in
                if out_kind == "host":
                    dst = np.full(nbytes_out, 77, np.uint8)
                    dptr = dst.ctypes.data
                elif out_kind == "pin":
                    dst = torch.full((nbytes_out,), 77, dtype=torch.uint8).pin_memory()
                    dptr = dst.data_ptr()
                else:
                    dst = torch.full((nbytes_out,), 77, dtype=torch.uint8, device="cuda")
                    dptr = dst.data_ptr()
                flags = {"host": 0, "pin": L.IN_PINNED, "dev": L.IN_DEVICE}[in_kind] | {"host": 0, "pin": L.OUT_PINNED, "dev": L.OUT_DEVICE}[out_kind]
                L.check(lib.sg_stft(e._h, src, n_ch, n_samples, cs, ss, dptr, ocs, flags, stream))
                torch.cuda.synchronize()
                return dst if out_kind == "host" else dst.cpu().numpy()

            want = call("dev", "dev")
            rows = np.zeros(nbytes_out, bool)
            for c in range(n_ch):
                rows[c * ocs * osz:(c * ocs + frames * bins) * osz] = True
            assert np.all(want[~rows] == 77), what
            for ik, ok in (("host", "host"), ("pin", "pin"), ("host", "dev"), ("dev", "host"), ("pin", "host"), ("host", "pin"), ("dev", "pin"), ("pin", "dev")):
                got = call(ik, ok)
                assert np.array_equal(got, want), (what, ik, ok)


def test_batch_call_on_pinned_host_buffers_in_place(S, oracle):
    """sg_stft with SG_IN_PINNED | SG_OUT_PINNED: samples read and rows stored in place over PCIe by the kernel (every
    load variant: sliding window, LDS ring, generic), same bits as the staged host-pointer call."""
    import torch
    from spectrogram_amd import _lib as L
    rng = np.random.default_rng(99)
    for n, hop, ss in ((1024, 256, 1), (2048, 735, 1), (512, 100, 2)):
        x = rng.uniform(-1, 1, (3, (n + 50 * hop) * ss)).astype(np.float32)
        with S.Engine(fft_size=n, hop=hop) as e:
            want = e.stft_host(x, sample_stride=ss)
            xp = torch.from_numpy(x).pin_memory()
            op = torch.zeros(want.shape, dtype=torch.float32).pin_memory()
            n_samples = x.shape[1] // ss
            L.check(e._lib.sg_stft(e._h, xp.data_ptr(), 3, n_samples, x.shape[1], ss, op.data_ptr(), 0,
                                   L.IN_PINNED | L.OUT_PINNED, None))
            torch.cuda.synchronize()
            assert np.array_equal(op.numpy(), want), (n, hop)
            if ss == 1:
                ref, _ = oracle.stft(oracle.OracleConfig.make(n, hop), x[2])
                assert_power_close(op.numpy()[2], ref, "pinned in place n=%d" % n)


# ---- the stated tolerance, anchored --------------------------------------------------------------
# north_star: "<= 1e-4 max relative error vs the JS reference" -- a float32 radix-2 FFT.  oracle/yardstick.py measures the
# HIP output and the two float32 radix-2 twins (the JS file's semantics, bit-equal to it under node; and the all-float32
# evaluation) against the float64 oracle on the SAME frames.  The first formulation of this test (largest error <= the
# all-float32 twin's per case, rms <= its rms at every size) failed on 7 of 12 cases on the GPU: the per-case maximum is an
# order statistic, and the rms at N >= 4096 was 4-7 % above the twin's until the kernel's derived twiddles were given
# chosen triplets.  What holds now -- rms <= the twin's at every size, 256 ... 32768 -- and what is asserted is stated
# assertion by assertion below and in DESIGN.md section 6.
TWIN_CASES = [(n, w, k) for n in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768) for w, k in (("hann", "uniform"), ("blackman", "int16"))]
_twin_results = {}


@pytest.mark.parametrize("n,window,kind", TWIN_CASES)
def test_error_is_no_worse_than_a_float32_radix2_fft(S, oracle, n, window, kind):
    import torch
    from oracle import yardstick
    from conftest import FLOOR_REL
    hop = n // 2 + 5                                            # odd hop: every window alignment, LDS-ring / generic loads
    frames = 10_000_000 // (n // 2) + 1                          # >= 1e7 bins
    ns = n + (frames - 1) * hop
    rng = np.random.default_rng(n * 7 + len(window))
    if kind == "uniform":
        x = rng.uniform(-1, 1, ns).astype(np.float32)
        dev, pcm = torch.from_numpy(x).cuda(), "f32"
    else:                                                        # a 16-bit source through the int16 load path
        q = rng.integers(-32768, 32768, ns, dtype=np.int64).astype(np.int16)
        x = q.astype(np.float32) / np.float32(32768.0)
        dev, pcm = torch.from_numpy(q).cuda(), "s16"
    with S.Engine(fft_size=n, hop=hop, window=window, pcm=pcm) as e:
        got = e.stft(dev)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
    r = yardstick.error_stats(x, n, hop, 1 if window == "blackman" else 0, {"hip": got})
    _twin_results[(n, window, kind)] = r
    hip, js, f32 = r["hip"], r["twin_js"], r["twin_f32"]
    line = "N=%5d %-8s %-7s %9d bins:" % (n, window, kind, r["bins"])
    for name, v in (("hip", hip), ("twin_js", js), ("twin_f32", f32)):
        line += "  %s F=1e-5 %.2e F=2e-5 %.2e peak max %.2e rms %.2e;" % (name, v["F=1e-05"], v["F=2e-05"], v["peak_max"], v["peak_rms"])
    print(line)
    # the statement itself
    assert hip["F=%g" % FLOOR_REL] <= REL, line
    # ... and its anchor.  Float32 FFT noise scales with the frame's peak, so the comparison is made on |dP| / max_k P.
    # In the mean square over >= 1e7 bins (a stable statistic: the three columns repeat to 1 % between 1e7 and 1e8 bins)
    # the kernel is no noisier than the all-float32 radix-2 FFT at EVERY size -- the reference's own sizes, 1024 and 2048
    # (UI/player.js:10), included; measured 0.92-0.97 of it.  At N >= 4096 four of the seven twiddles of each radix-8
    # pass are products of the other three (registers: DESIGN.md section 5.1); with the three taken from the W_N^k table
    # that cost a rounding and the ratio was 1.04-1.07.  The host now picks the stored triplet so that the products land
    # nearest the true twiddles (spectro_api.hip, derived_triplets): 0.92-0.94, below the 0.94 of holding all seven
    # (profiles/r04_tolerance_tail.txt, section D).
    assert hip["peak_rms"] <= 1.0 * f32["peak_rms"], line
    # The largest error over 1e7 bins is an order statistic (it moves by +-20 % between seeds for all three columns), so
    # it is held to a gross bound only: an indexing or twiddle bug moves it by orders of magnitude, not by half.
    assert hip["peak_max"] <= 1.5 * f32["peak_max"], line
    # The JS file's own semantics (double butterflies, one rounding per store) are quieter than any all-float32 FFT:
    # measured 0.85-0.88 of the kernel's rms.  Recorded, and bounded from the other side as well.
    assert hip["peak_rms"] <= 1.25 * js["peak_rms"], line


def test_fuzz_case_of_round_3_that_crossed_the_old_floor(S, oracle):
    """Seed 82, case 83 of test_randomized_formats_strides_and_epilogues (gpurun_out/seed82.log, round 3): int16 input,
    N = 512, hop 517, |X| output, 1.052e-4 at the old floor of 1e-5 on a bin 49.8 dB below its frame's peak.  Replayed
    from the same generator state, pinned here with BOTH floors reported next to what the float32 radix-2 twins do on the
    same frames (keep the failing case as a regression test)."""
    import torch
    rng = np.random.default_rng(82 + 7)
    found = None
    for case in range(84):   # replay the sweep's draws up to the case (same order of rng calls as the sweep)
        fmt = ["f32", "s16", "s24"][int(rng.integers(0, 3))]
        n = int(2 ** rng.integers(5, 16))
        hop = int(rng.choice([n // 4, n // 2, max(1, n // 3), n - 4, n - 3, 1 + int(rng.integers(0, n)), 1 + int(rng.integers(0, n // 2)), n + 5]))
        k = int(rng.choice([1, 1, 1, 2, 3]))
        ss = int(rng.choice([1, 1, 1, 2, 3]))
        n_ch = int(rng.integers(1, 5))
        frames = int(rng.integers(1, 30)) if n <= 2048 else int(rng.integers(1, 12 if n <= 8192 else 5))
        n_samples = n + (frames - 1) * hop + int(rng.integers(0, hop))
        row = n_samples * k * ss
        cs = row + int(rng.integers(0, 9))
        off = int(rng.integers(0, 6))
        out = ["power", "db_u8", "mag", "db"][int(rng.integers(0, 4))]
        cols = int(rng.choice([0, 0, 37, 256])) if 512 <= n <= 8192 else 0
        total = off + n_ch * cs + 4
        if fmt == "f32":
            raw = rng.uniform(-1, 1, total).astype(np.float32)
        elif fmt == "s16":
            raw = rng.integers(-32768, 32768, total, dtype=np.int64).astype(np.int16)
        else:
            raw = rng.integers(-(1 << 23), 1 << 23, total, dtype=np.int64)
        if case == 83:   # (the sweep of round 3 did not yet draw a forced run length per case)
            found = (fmt, n, hop, k, ss, n_ch, frames, n_samples, row, cs, off, out, cols, raw)
    fmt, n, hop, k, ss, n_ch, frames, n_samples, row, cs, off, out, cols, raw = found
    assert (fmt, n, hop, out) == ("s16", 512, 517, "mag"), found[:12]
    as_f = raw.astype(np.float32) / np.float32(32768.0)
    dev = torch.from_numpy(raw).cuda()
    with S.Engine(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out, pcm=fmt, downmix=k,
                  **({"logfreq_cols": cols} if cols else {})) as e:
        dout = torch.empty(n_ch * frames * e.bins, dtype=torch.float32, device="cuda")
        e.stft_device_ptr(dev.data_ptr() + 2 * off, n_ch, n_samples, cs, ss, dout.data_ptr(), frames * e.bins,
                          torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got_all = dout.cpu().numpy().astype(np.float64).reshape(n_ch, frames, -1)
    assert cols == 0
    cfg = oracle.OracleConfig.make(n, hop, window=1, scale=1)
    worst = {1e-5: 0.0, 2e-5: 0.0}
    twin = {1e-5: 0.0, 2e-5: 0.0}
    for c in range(n_ch):
        xs = as_f[off + c * cs: off + c * cs + row: ss]
        mono = xs.reshape(n_samples, k).astype(np.float64).mean(axis=1).astype(np.float32)
        ref, _ = oracle.stft(cfg, mono)
        t32 = oracle.stft_f32(cfg, mono).astype(np.float64)
        mx = ref.max(axis=1, keepdims=True)
        for f in worst:
            den = np.maximum(ref, f * mx)
            worst[f] = max(worst[f], float((np.abs(got_all[c] ** 2 - ref) / den).max()))
            twin[f] = max(twin[f], float((np.abs(t32 - ref) / den).max()))
    print("seed 82 case 83: hip F=1e-5 %.3e F=2e-5 %.3e; float32 radix-2 twin on the same frames F=1e-5 %.3e F=2e-5 %.3e"
          % (worst[1e-5], worst[2e-5], twin[1e-5], twin[2e-5]))
    assert worst[2e-5] <= REL


def test_fft_size_change_keeps_the_sample_history(S, oracle):
    """analyser.fftSize = n on a live analyser (UI/player.js:10 sets it once, before any audio; the platform object allows
    it at any time): the smoothing state restarts, the samples stay -- an AnalyserNode holds its latest 32768 samples
    whatever fftSize is, so a LARGER size set later transforms audio that arrived before the change (VERDICT r3 item 8:
    the re-plan used to drop the history)."""
    rng = np.random.default_rng(808)
    x = (rng.uniform(-1, 1, 6000) * 0.4).astype(np.float32)
    a = S.Analyser(fftSize=2048)
    a.smoothingTimeConstant = 0.5
    a.push(x[:5000])
    by = np.zeros(1024, np.uint8)
    a.getByteFrequencyData(by)                       # smoothing state now non-zero
    for n in (512, 4096, 8192, 2048):
        a.fftSize = n
        assert a.frequencyBinCount == n // 2
        last = np.concatenate([np.zeros(max(0, n - 5000), np.float32), x[max(0, 5000 - n):5000]])
        td = np.zeros(n, np.float32)
        a.getFloatTimeDomainData(td)
        assert np.array_equal(td, last), "fftSize %d: the history before the change is gone" % n
        by = np.zeros(n // 2, np.uint8)
        a.getByteFrequencyData(by)                   # first poll after the change: smoothing from zero
        st = np.zeros(n // 2)
        _, ref_b = oracle.analyser_poll(n, last, 0.5, -100.0, -30.0, st)
        assert_bytes_in_bounds(by, *ByteBounds(n // 2).step(poll_power(oracle, last), 0.5), "fftSize %d" % n, ref_b)
    a.push(x[5000:])                                  # and the feed goes on
    td = np.zeros(2048, np.float32)
    a.getFloatTimeDomainData(td)
    assert np.array_equal(td, x[-2048:])
    with pytest.raises(ValueError):
        a.fftSize = 1000
    a.close()


def test_debug_switches_are_an_explicit_entry(S):
    """sg_debug_set: the tuning / test switches of a handle (they used to be read from the environment)."""
    import ctypes as C
    from spectrogram_amd import _lib as L
    with S.Engine(fft_size=1024, hop=300, debug={"no_lring": 1}) as e:
        lib = L.load()
        assert lib.sg_debug_set(e._h, b"no_such_switch", 1) == L.ERR_INVALID
        assert lib.sg_debug_set(e._h, b"run_len", -1) == L.ERR_INVALID
        assert lib.sg_debug_set(None, b"run_len", 1) == L.ERR_INVALID
        assert lib.sg_debug_set(e._h, b"run_len", 19) == 0
        import torch
        e.stft(torch.zeros(1, 1024 + 300 * 60, device="cuda"))
        variant, run_len, _, _ = e.last_launch()
        assert (variant, run_len) == (0, 19)


@pytest.mark.parametrize("name", golden_names("noise_") + ["sweep_n256_h128", "two_tone_n1024"])
def test_golden_vectors_also_hold_at_the_old_floor(S, name):
    """one assertion stays at the floor of rounds 1-2 (1e-5 of the frame's peak) over the committed golden
    vectors, so that a real precision regression of a later kernel change is still seen even though the stated floor
    is 2e-5 (at which the float32 radix-2 twin itself is the yardstick, tests above and DESIGN.md section 6)."""
    g = golden(name)
    n, hop = int(g["fft_size"]), int(g["hop"])
    if not supported(n):
        pytest.skip("size not built")
    with S.Engine(fft_size=n, hop=hop) as e:
        P = e.stft_host(g["pcm"]).astype(np.float64)
    ref = g["power"]
    den = np.maximum(ref, 1e-5 * ref.max(axis=1, keepdims=True)) + 1e-26
    assert (np.abs(P - ref) / den).max() <= 1e-4, name


@pytest.mark.parametrize("pcm", ["f32", "s16"])
@pytest.mark.parametrize("n,cols", [(4096, 0), (2048, 256), (1024, 0), (256, 0), (8192, 0), (64, 0), (32768, 0), (1024, 256)])
def test_fused_push_poll_is_the_two_calls_bit_for_bit(S, oracle, n, cols, pcm):
    """sg_push_poll_*: the per-animation-frame step of the reference as ONE launch (new audio reaches the analyser, then
    getByteFrequencyData: UI/spectrogram.js:153-161 -> 3D/visualizer.js:346-358).  The fused transform reads the newest
    samples straight from the caller's page-locked (or device) block and files them in the history on its way; a twin
    handle fed by sg_push + sg_poll_* must see the same bytes, the same float dB rows, the same history ring and the same
    time-domain data after every block -- shapes the fused launch serves (even blocks up to one window, pinned or device
    input) and shapes it hands to the two-call path (odd blocks, blocks longer than the window, pageable input) alike."""
    import ctypes as C
    import torch
    from spectrogram_amd import _lib as L
    lib = L.load()
    n_ch, m = 3, n // 2
    row = cols if cols else m
    rng = np.random.default_rng(n + cols + len(pcm))
    cfg = S.make_config(fft_size=n, hop=n, window="blackman", scale="inv_n", out="mag", smoothing=0.6, logfreq_cols=cols, pcm=pcm)
    hs = []
    for _ in range(2):
        h = C.c_void_p()
        L.check(lib.sg_create(C.byref(cfg), C.byref(h)))
        L.check(lib.sg_stream_begin(h, n_ch))
        L.check(lib.sg_ring_begin(h, 3))
        hs.append(h)
    fused, plain = hs
    dt = torch.int16 if pcm == "s16" else torch.float32
    blocks = [n, 2, 64, 734, 735, 1024 if n >= 1024 else n // 2, n + 6, 128, n // 2 + 2, n]
    kinds = ["pinned", "pinned", "device", "pinned", "pinned", "pinned", "pinned", "pageable", "device", "pinned"]
    a8, b8 = np.zeros((n_ch, row), np.uint8), np.zeros((n_ch, row), np.uint8)
    af, bf = np.zeros((n_ch, row), np.float32), np.zeros((n_ch, row), np.float32)
    at, bt = np.zeros((n_ch, n), np.float32), np.zeros((n_ch, n), np.float32)
    hist = np.zeros((n_ch, n), np.float32)
    bb = [ByteBounds(m) for _ in range(n_ch)]
    try:
        for i, (blk, kind) in enumerate(zip(blocks, kinds)):
            if pcm == "s16":
                x = torch.from_numpy(rng.integers(-20000, 20000, (n_ch, blk + 3), dtype=np.int64).astype(np.int16))
                as_f = x[:, :blk].numpy().astype(np.float32) / np.float32(32768.0)
            else:
                x = torch.from_numpy(rng.uniform(-1, 1, (n_ch, blk + 3)).astype(np.float32))
                as_f = x[:, :blk].numpy()
            stride = blk + 3                                     # padded rows: channel_stride != n_new
            if kind == "pinned":
                buf, flags = x.pin_memory(), L.IN_PINNED
            elif kind == "device":
                buf, flags = x.cuda(), L.IN_DEVICE
            else:
                buf, flags = x, L.PTR_HOST
            if i % 2 == 0:
                L.check(lib.sg_push_poll_u8(fused, buf.data_ptr(), blk, stride, a8.ctypes.data, row, row, flags, None))
                L.check(lib.sg_push(plain, buf.data_ptr(), blk, stride, flags, None))
                L.check(lib.sg_poll_u8(plain, b8.ctypes.data, row, row, L.PTR_HOST, None))
                assert np.array_equal(a8, b8), "block %d (%d samples, %s): bytes differ" % (i, blk, kind)
            else:
                L.check(lib.sg_push_poll_f32(fused, buf.data_ptr(), blk, stride, af.ctypes.data, row, row, flags, None))
                L.check(lib.sg_push(plain, buf.data_ptr(), blk, stride, flags, None))
                L.check(lib.sg_poll_f32(plain, bf.ctypes.data, row, row, L.PTR_HOST, None))
                assert np.array_equal(af.view(np.uint32), bf.view(np.uint32)), "block %d (%d samples, %s): dB rows differ" % (i, blk, kind)
            torch.cuda.synchronize()
            L.check(lib.sg_poll_time_f32(fused, at.ctypes.data, n, n, L.PTR_HOST, None))
            L.check(lib.sg_poll_time_f32(plain, bt.ctypes.data, n, n, L.PTR_HOST, None))
            assert np.array_equal(at, bt), "block %d: the fused launch filed the samples wrongly" % i
            hist = np.concatenate([hist, as_f], axis=1)[:, -n:]
            assert np.array_equal(at, hist), "block %d: history is not the last N samples" % i
            if cols == 0:                                         # and against the oracle, not only against the twin
                for c in range(n_ch):
                    lo, hi = bb[c].step(poll_power(oracle, hist[c]), 0.6)
                    if i % 2 == 0:
                        assert_bytes_in_bounds(a8[c], lo, hi, "block %d channel %d" % (i, c))
        ia, ib = np.zeros((n_ch, 3, row), np.uint8), np.zeros((n_ch, 3, row), np.uint8)
        L.check(lib.sg_ring_read(fused, ia.ctypes.data, 0, L.PTR_HOST, None))
        L.check(lib.sg_ring_read(plain, ib.ctypes.data, 0, L.PTR_HOST, None))
        assert np.array_equal(ia, ib)
        # which path ran: blocks of an even length up to one window from pinned / device memory are one launch
        # (sg_last_launch says run_len 1, one run per channel either way; the state check is the history equality above)
    finally:
        for h in hs:
            lib.sg_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("n,keep", [(256, 257), (256, 699), (1024, 1024), (1024, 3070), (4096, 32768)])
def test_history_ring_of_any_length(S, keep, n):
    """The poll-mode history is a double-written ring of H = cfg.history_samples samples per channel (2 H floats: sample L at
    L mod H and at L mod H + H), so the newest N <= H samples always lie contiguous and no push ever moves old samples
    (rounds 2-4 compacted 3 H-float rows: 65 MB of device copies every 64th hop of a 512-channel bank).  Random
    push lengths -- odd, even, longer than the ring -- through every way in (pageable, page-locked in place, device, the
    fused push+poll launch) on ring lengths that are odd, not multiples of four, and not multiples of the pushes: after
    every block the time-domain getter must return exactly the last N samples, the frequency bytes must equal a twin
    handle's whose ring is N long, and a LARGER fftSize set at the end must find the older samples (the reason H > N
    exists: UI/player.js:10 on a live analyser)."""
    import ctypes as C
    import torch
    from spectrogram_amd import _lib as L
    lib = L.load()
    n_ch = 3
    rng = np.random.default_rng(keep * 7 + n)
    hs = []
    for hist in (keep, 0):
        cfg = S.make_config(fft_size=n, hop=n, window="blackman", scale="inv_n", out="mag", smoothing=0.3, history_samples=hist)
        h = C.c_void_p()
        L.check(lib.sg_create(C.byref(cfg), C.byref(h)))
        L.check(lib.sg_stream_begin(h, n_ch))
        hs.append(h)
    ring, twin = hs
    total = np.zeros((n_ch, 0), np.float32)
    a8, b8 = np.zeros((n_ch, n // 2), np.uint8), np.zeros((n_ch, n // 2), np.uint8)
    at = np.zeros((n_ch, n), np.float32)
    sizes = [1, 2, 3, 5, 64, 100, 255, 256, 734, 735, n - 1, n, n + 1, keep - 1, keep, keep + 5, 2 * keep + 3]
    try:
        pushed, i = 0, 0
        while pushed < 3 * keep + n or i < 24:
            blk = int(rng.choice(sizes)) if i % 5 else int(rng.integers(1, n + 1)) & ~1 or 2
            kind = ("pinned", "device", "pageable", "fused", "fused_dev")[int(rng.integers(0, 5))]
            x = torch.from_numpy(rng.uniform(-1, 1, (n_ch, blk + 1)).astype(np.float32))
            stride = blk + 1
            buf, flags = (x.pin_memory(), L.IN_PINNED) if kind in ("pinned", "fused") else \
                ((x.cuda(), L.IN_DEVICE) if kind in ("device", "fused_dev") else (x, L.PTR_HOST))
            if kind.startswith("fused"):   # one launch when blk is even and <= N; the two calls otherwise -- same results
                L.check(lib.sg_push_poll_u8(ring, buf.data_ptr(), blk, stride, a8.ctypes.data, n // 2, n // 2, flags, None))
            else:
                L.check(lib.sg_push(ring, buf.data_ptr(), blk, stride, flags, None))
                L.check(lib.sg_poll_u8(ring, a8.ctypes.data, n // 2, n // 2, L.PTR_HOST, None))
            L.check(lib.sg_push(twin, buf.data_ptr(), blk, stride, flags, None))
            L.check(lib.sg_poll_u8(twin, b8.ctypes.data, n // 2, n // 2, L.PTR_HOST, None))
            torch.cuda.synchronize()
            total = np.concatenate([total, x[:, :blk].numpy()], axis=1)[:, -max(keep, n):]
            want = np.concatenate([np.zeros((n_ch, n), np.float32), total], axis=1)[:, -n:]
            L.check(lib.sg_poll_time_f32(ring, at.ctypes.data, n, n, L.PTR_HOST, None))
            assert np.array_equal(at, want), "block %d (%d samples, %s): not the last N samples" % (i, blk, kind)
            assert np.array_equal(a8, b8), "block %d (%d samples, %s): bytes differ from the N-sample ring's" % (i, blk, kind)
            pushed += blk
            i += 1
        # a larger fftSize finds the audio that arrived before the change -- as much of it as the ring keeps
        big = 1 << (int(keep).bit_length() - 1)
        if big > n:
            L.check(lib.sg_set_fft_size(ring, big))
            tb = np.zeros((n_ch, big), np.float32)
            L.check(lib.sg_poll_time_f32(ring, tb.ctypes.data, big, big, L.PTR_HOST, None))
            assert np.array_equal(tb, total[:, -big:])
        assert lib.sg_set_fft_size(ring, 2 * big if 2 * big <= 32768 and 2 * big > keep else 65536) != 0   # beyond the ring / the range: refused,
        tn = np.zeros((n_ch, max(big, n)), np.float32)                                                     # ... and the handle is as it was
        L.check(lib.sg_poll_time_f32(ring, tn.ctypes.data, max(big, n), max(big, n), L.PTR_HOST, None))
        assert np.array_equal(tn, total[:, -max(big, n):])
    finally:
        for h in hs:
            lib.sg_destroy(h)


def test_staged_pinned_input_on_two_streams(S):
    """sg_stft with page-locked input that goes through the handle's staging buffer (a shape whose frames
    would be re-fetched over PCIe if read in place: interleaved rows here) and device output returns without waiting.  A
    second such call on ANOTHER stream rewrites the same staging buffer: it must wait (device-side) for the first call's
    kernel, or the first call transforms the second call's samples.  Three calls on three streams, three different inputs,
    a long first kernel: every output equals the all-device call's, bit for bit."""
    import ctypes as C
    import torch
    from spectrogram_amd import _lib as L
    lib = L.load()
    n, hop, n_ch, ns = 1024, 300, 48, 400_000
    with S.Engine(fft_size=n, hop=hop) as e:
        frames = e.num_frames(ns)
        xs = [(torch.rand(n_ch, 2 * ns, generator=torch.Generator().manual_seed(70 + i)) * 2 - 1) for i in range(3)]
        pins = [x.pin_memory() for x in xs]
        want = []
        for x in xs:                                   # the reference call: device memory both sides
            o = torch.empty(n_ch, frames, n // 2, device="cuda")
            L.check(lib.sg_stft(e._h, x.cuda().data_ptr(), n_ch, ns, 2 * ns, 2, o.data_ptr(), 0, L.PTR_DEVICE, None))
            torch.cuda.synchronize()
            want.append(o)
        assert e.last_launch()[0] == 0                 # generic loads: the page-locked rows are staged, not read in place
        outs = [torch.zeros(n_ch, frames, n // 2, device="cuda") for _ in range(3)]
        streams = [torch.cuda.Stream() for _ in range(3)]
        torch.cuda.synchronize()
        for x, o, st in zip(pins, outs, streams):      # back to back, no synchronisation in between
            L.check(lib.sg_stft(e._h, x.data_ptr(), n_ch, ns, 2 * ns, 2, o.data_ptr(), 0, L.IN_PINNED | L.OUT_DEVICE,
                                C.c_void_p(st.cuda_stream)))
        torch.cuda.synchronize()
        for i in range(3):
            assert torch.equal(outs[i], want[i]), "call %d transformed another call's samples" % i
        # ... and a pageable call afterwards (synchronous, a fourth stream) still sees its own samples
        o4 = np.zeros((n_ch, frames, n // 2), np.float32)
        st4 = torch.cuda.Stream()
        L.check(lib.sg_stft(e._h, xs[1].numpy().ctypes.data, n_ch, ns, 2 * ns, 2, o4.ctypes.data, 0, L.PTR_HOST,
                            C.c_void_p(st4.cuda_stream)))
        assert np.array_equal(o4, want[1].cpu().numpy())


def test_analyser_objects_run_a_frame_as_one_call(S, oracle):
    """Analyser.pushAndGetByteFrequencyData / pushAndGetFloatFrequencyData: one animation frame of the reference
    (UI/spectrogram.js:153-161 -> 3D/visualizer.js:346-358) as one call.  From pageable numpy arrays the C ABI stages small
    blocks through page-locked memory it owns and runs ONE fused launch (sg_last_launch: run length 1, one run per channel,
    and no push kernel in between is visible as identical results); bytes, dB rows and the samples left behind must equal
    a twin analyser's that is fed by push() + get...() -- multi-channel, int16 feed, log-frequency rows, a caller's array
    shorter than the row, odd and over-long blocks (which fall back to the two calls) included."""
    rng = np.random.default_rng(4242)
    for kw, n_ch in ((dict(fftSize=2048), 1), (dict(fftSize=1024, n_channels=3), 3), (dict(fftSize=1024, pcm="s16"), 1),
                     (dict(fftSize=2048, logfreq_cols=256, n_channels=2), 2)):
        one, two = S.Analyser(**kw), S.Analyser(**kw)
        one.smoothingTimeConstant = two.smoothingTimeConstant = 0.4
        n, row = one.fftSize, one.rowLength
        hist = np.zeros((n_ch, 0), np.float32)
        for i, blk in enumerate([736, 128, 735, n, 2, n + 6, 512, 64]):
            if kw.get("pcm") == "s16":
                x = rng.integers(-20000, 20000, (n_ch, blk)).astype(np.int16)
                xf = x.astype(np.float32) / np.float32(32768.0)
            else:
                x = rng.uniform(-1, 1, (n_ch, blk)).astype(np.float32)
                xf = x
            hist = np.concatenate([hist, xf], axis=1)[:, -n:]
            short = row - 5 if i == 3 else row             # a caller's array shorter than the row: the rest stays untouched
            if i % 2 == 0:
                a, b = np.full((n_ch, short), 7, np.uint8), np.full((n_ch, short), 7, np.uint8)
                one.pushAndGetByteFrequencyData(x if n_ch > 1 else x[0], a if n_ch > 1 else a[0])
                two.push(x)
                two.getByteFrequencyData(b)
            else:
                a, b = np.full((n_ch, short), 7, np.float32), np.full((n_ch, short), 7, np.float32)
                one.pushAndGetFloatFrequencyData(x, a)
                two.push(x)
                two.getFloatFrequencyData(b)
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (kw, i, blk)
            ta, tb = np.zeros((n_ch, n), np.float32), np.zeros((n_ch, n), np.float32)
            one.getFloatTimeDomainData(ta)
            two.getFloatTimeDomainData(tb)
            want = np.concatenate([np.zeros((n_ch, n), np.float32), hist], axis=1)[:, -n:]
            assert np.array_equal(ta, tb) and np.array_equal(ta, want), (kw, i, blk)
        with pytest.raises(TypeError):
            one.pushAndGetByteFrequencyData(np.zeros((n_ch, 64), np.float32 if kw.get("pcm") != "s16" else np.int16), np.zeros((n_ch, row), np.float32))
        one.close()
        two.close()


@pytest.mark.parametrize("n,hop", [(1024, 256), (1024, 300), (2048, 512), (2048, 735), (2048, 1024)])
@pytest.mark.parametrize("out", ["db_u8", "power", "db"])
def test_256_column_rows_have_their_own_instantiation(S, n, hop, out):
    """Round 5: the reference's row -- 256 log-frequency columns (sonogram3DWidth, 3D/visualizer.js:54) at N = 1024 / 2048 -- runs
    through an instantiation that carries the consecutive-column path alone (OUT_LFFAST, chosen at launch; +8-12 % at N = 1024,
    four waves per SIMD instead of three).  It must hand out exactly what the general log-frequency kernel does
    (sg_debug_set "no_lffast"), for every load variant underneath and both alignments; other column counts never take it."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(n + hop)
    x = torch.rand(5, n + 37 * hop + 3, device="cuda", generator=g) * 2 - 1
    for align in ("centre", "vertex"):
        kw = dict(fft_size=n, hop=hop, out=out, window="blackman", scale="inv_n", logfreq_cols=256, logfreq_align=align)
        with S.Engine(debug={}, **kw) as fast, S.Engine(debug={"no_lffast": 1}, **kw) as general:
            a, b = fast.stft(x), general.stft(x)
            torch.cuda.synchronize()
            assert a.shape == (5, 38, 256) and torch.equal(a, b), (n, hop, out, align)
    with S.Engine(fft_size=n, hop=hop, out=out, logfreq_cols=128, debug={}) as e, \
            S.Engine(fft_size=n, hop=hop, out=out, logfreq_cols=128, debug={"no_lffast": 1}) as f:
        assert torch.equal(e.stft(x), f.stft(x))     # 128 columns: the general kernel either way


@pytest.mark.parametrize("n,hop", [(512, 100), (512, 517), (16384, 5000), (32768, 9001)])
def test_plain_rows_on_the_generic_loads_have_their_own_instantiation(S, n, hop):
    """Round 5: on the generic loads, plain rows (unit stride, float32, no down-mix -- known at launch) run an instantiation that
    carries the one 8-byte load per point alone (load variant 5: 100-112 VGPRs against 148-154; N >= 16384 no longer spills:
    +50 / +68 % at N = 16384 / 32768).  Same bits as the general generic kernel (sg_debug_set "no_gen_unit"), every output kind;
    strided rows keep variant 0."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(n + hop)
    x = torch.rand(3, n + 9 * hop + 5, device="cuda", generator=g) * 2 - 1
    for out in ("power", "db_u8", "mag"):
        with S.Engine(fft_size=n, hop=hop, out=out, debug={}) as a, S.Engine(fft_size=n, hop=hop, out=out, debug={"no_gen_unit": 1}) as b:
            ra, rb = a.stft(x), b.stft(x)
            torch.cuda.synchronize()
            assert (a.last_launch()[0], b.last_launch()[0]) == (5, 0)
            assert torch.equal(ra, rb), (n, hop, out)
            a.stft(x, sample_stride=2)
            assert a.last_launch()[0] == 0


def test_engine_created_before_torch_is_imported():
    """A PyTorch-ROCm wheel ships its own HIP runtime; libspectro.so names the system's.  Whichever initialises first serves the
    process, and torch on top of the system runtime reports no device (round 5: an analyser created before anything had imported
    torch left `torch.cuda.is_available()` False for every later test).  `_lib.load()` therefore loads torch's runtime first when
    torch is installed but not imported; in a fresh process: engine first, torch second, both work."""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "assert 'torch' not in sys.modules\n"
        "import spectrogram_amd as S\n"
        "a = S.Analyser(fftSize=1024); a.push(np.ones(1024, np.float32) * 0.5); b = np.zeros(512, np.uint8)\n"
        "a.getByteFrequencyData(b); a.close(); assert b[0] > 0\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "x = torch.rand(1, 4096, device='cuda')\n"
        "e = S.Engine(fft_size=1024, hop=256); assert float(e.stft(x).sum()) > 0; e.close()\n"
        "i = S.runtime_info()\n"
        "import os\n"
        "assert i['policy'] == 'auto' and i['preloaded'] and os.path.dirname(os.path.realpath(i['library_path'])) == "
        "os.path.join(os.path.dirname(os.path.realpath(torch.__file__)), 'lib'), i\n"
        "print('both-orders-ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "both-orders-ok" in r.stdout, r.stderr[-2000:]


def test_a_refused_fused_poll_leaves_the_history_alone(S):
    """sg_push_poll_* checks the poll's arguments before the history pointer moves: a refused call (bad length, a pageable
    buffer passed as SG_OUT_PINNED) must not leave a block-sized hole of unwritten samples behind."""
    import ctypes as C
    import torch
    from spectrogram_amd import _lib as L
    lib = L.load()
    n = 1024
    cfg = S.make_config(fft_size=n, hop=n, window="blackman", scale="inv_n", out="mag")
    h = C.c_void_p()
    L.check(lib.sg_create(C.byref(cfg), C.byref(h)))
    try:
        L.check(lib.sg_stream_begin(h, 1))
        x = torch.rand(1, n).pin_memory()
        out = np.zeros(n // 2, np.uint8)
        L.check(lib.sg_push_poll_u8(h, x.data_ptr(), n, n, out.ctypes.data, n // 2, n // 2, L.IN_PINNED, None))
        before = np.zeros(n, np.float32)
        L.check(lib.sg_poll_time_f32(h, before.ctypes.data, n, n, L.PTR_HOST, None))
        y = torch.rand(1, 256).pin_memory()
        assert lib.sg_push_poll_u8(h, y.data_ptr(), 256, 256, out.ctypes.data, -1, 0, L.IN_PINNED, None) == L.ERR_INVALID
        assert lib.sg_push_poll_u8(h, y.data_ptr(), 256, 256, None, 8, 8, L.IN_PINNED, None) == L.ERR_INVALID
        assert lib.sg_push_poll_u8(h, y.data_ptr(), 256, 256, out.ctypes.data, 8, 8, L.IN_PINNED | L.OUT_PINNED, None) == L.ERR_INVALID
        after = np.zeros(n, np.float32)
        L.check(lib.sg_poll_time_f32(h, after.ctypes.data, n, n, L.PTR_HOST, None))
        assert np.array_equal(before, after) and np.array_equal(before, x.numpy()[0])
    finally:
        lib.sg_destroy(h)
