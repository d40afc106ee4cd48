"""GPU: the LDS-sample-ring load variants (arbitrary hops at unit stride) -- the path the reference's own cadence takes
(fftSize 2048, one poll per animation frame ~ 735 samples: UI/player.js:10, UI/spectrogram.js:153-161).

Two kinds of check.  Against the float64 oracle, under the stated tolerance.  And against the generic-load variant of
the same kernel (sg_debug_set "no_lring", the Engine's debug= switches): both feed the same float32 samples to the same
butterflies in the same order, so the outputs must agree BIT FOR BIT -- any difference is a ring-addressing bug
(wrap row, parity, split layout, staging predicate, end-of-buffer granule), however small it looks."""
import numpy as np
import pytest

from conftest import assert_power_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import spectrogram_amd as S
    return S


def _both(S, x_dev, n, hop, **kw):
    """(ring output, generic output, ring launch info) for one device tensor [channels, samples].  The switches go
    through sg_debug_set: one engine with "lring_all" (the 4-row ring variant also at the sizes where the
    launcher prefers the generic loads for hops above N/2), one with "no_lring" (generic loads)."""
    import torch
    with S.Engine(debug={"lring_all": 1}, fft_size=n, hop=hop, **kw) as e:
        a = e.stft(x_dev)
        torch.cuda.synchronize()
        info = e.last_launch()
    # small launches are cut into runs of 4 frames and up; the ring's wrap rows, staging predicates and mirror writes
    # only all come into play on long runs, so the same call again in 48-frame runs ("run_len")
    with S.Engine(debug={"lring_all": 1, "run_len": 48}, fft_size=n, hop=hop, **kw) as e:
        a_long = e.stft(x_dev)
        torch.cuda.synchronize()
        assert e.last_launch()[1] == min(48, e.num_frames(x_dev.shape[1] // (3 if kw.get("pcm") == "s24" else 1)))
    assert torch.equal(a, a_long), "n=%d hop=%d: 48-frame runs and the launcher's runs differ" % (n, hop)
    with S.Engine(debug={"no_lring": 1}, fft_size=n, hop=hop, **kw) as e:
        b = e.stft(x_dev)
        torch.cuda.synchronize()
        info_g = e.last_launch()
    assert info_g[0] & 7 == 0, "no_lring must select the generic loads"
    return a, b, info


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192])
def test_ring_equals_generic_loads_bit_for_bit(S, n):
    """Hops on both sides of every boundary the ring code has: 1, odd / even, the K = 2 / K = 4 staging switch at
    N/2 - 4, the largest ring hop N - 4, hops that make the window wrap in every row; ragged channel strides; bases
    that are not 16-byte aligned; runs that end inside a granule."""
    import torch
    rng = np.random.default_rng(n)
    hops = [1, 3, 100, 300, 301, n // 4 + 1, n // 2 - 5, n // 2 - 4, n // 2 - 3, n // 2 + 1, 735, n - 5, n - 4]
    for hop in hops:
        frames = int(rng.integers(70, 150)) if hop > 8 else 700
        n_ch = int(rng.integers(1, 4))
        n_samples = n + (frames - 1) * hop + int(rng.integers(0, hop))
        cs = n_samples + int(rng.integers(0, 9))
        off = int(rng.integers(0, 8))                       # float offset of the base: 4-byte aligned only
        buf = torch.from_numpy(rng.uniform(-1, 1, off + n_ch * cs).astype(np.float32)).cuda()
        x = buf[off:off + n_ch * cs].view(n_ch, cs)[:, :n_samples]
        a, b, info = _both(S, x, n, hop)
        want = 1 if (hop + 3) // 4 + 1 <= 2 * (n // 16) else 3
        assert info[0] == want, "hop %d ran load variant %d, expected the LDS ring (%d)" % (hop, info[0], want)
        assert torch.equal(a, b), "n=%d hop=%d: ring and generic loads differ" % (n, hop)


@pytest.mark.parametrize("run_len_env", [None, 37])
@pytest.mark.parametrize("n,hop", [(1024, 300), (1024, 301), (2048, 735), (2048, 736), (4096, 1111), (1024, 7)])
def test_ring_vs_oracle_across_run_boundaries(S, oracle, n, hop, run_len_env):
    """Against the float64 oracle, on the frames either side of every run boundary and the first / last frames."""
    import torch
    rng = np.random.default_rng(hop)
    frames = 400
    x = rng.uniform(-1, 1, (2, n + (frames - 1) * hop + 3)).astype(np.float32)
    with S.Engine(debug={"run_len": run_len_env}, fft_size=n, hop=hop) as e:
        got = e.stft(torch.from_numpy(x).cuda()).cpu().numpy()
        variant, run_len, _, _ = e.last_launch()
    assert variant in (1, 3) and (run_len == run_len_env if run_len_env else 4 <= run_len <= 96)
    cfg = oracle.OracleConfig.make(n, hop)
    pick = sorted({0, 1, frames - 2, frames - 1} | {k * run_len + d for k in range(1, frames // run_len + 1) for d in (-1, 0, 1)
                                                     if 0 <= k * run_len + d < frames})
    for c in range(2):
        for f in pick:
            ref, _ = oracle.stft(cfg, x[c, f * hop:f * hop + n])
            assert_power_close(got[c, f:f + 1], ref, "n=%d hop=%d ch=%d frame %d (run_len %d)" % (n, hop, c, f, run_len))


@pytest.mark.parametrize("n,hop", [(1024, 333), (2048, 735), (1024, 256), (4096, 2000)])
def test_ring_int16_input(S, oracle, n, hop):
    """int16 PCM rides the same ring (converted on the way in): against the oracle and against the generic loads."""
    import torch
    rng = np.random.default_rng(16 + hop)
    x = rng.integers(-32768, 32767, (2, n + 90 * hop + 1), dtype=np.int16)
    xd = torch.from_numpy(x).cuda()[:, 1:]                 # 2-byte aligned base: the sliding window needs 4
    a, b, info = _both(S, xd, n, hop, pcm="s16")
    assert info[0] in (9, 11) and torch.equal(a, b)
    cfg = oracle.OracleConfig.make(n, hop)
    ref, _ = oracle.stft(cfg, x[1, 1:].astype(np.float32) / 32768.0)
    assert_power_close(a[1].cpu().numpy(), ref, "int16 ring")


@pytest.mark.parametrize("n,hop", [(1024, 256), (1024, 301), (2048, 735), (2048, 1024), (8192, 4000)])
def test_ring_packed_int24_input(S, oracle, n, hop):
    """Packed little-endian 24-bit PCM (WAV "PCM 24"): twelve-byte granules at byte-aligned addresses through the ring,
    including hop = N/4 and N/2 (int24 has no register-window variant).  Sign / byte-order corners, against the oracle
    and, bit for bit, against the generic loads (three byte loads per sample)."""
    import torch
    rng = np.random.default_rng(24 + hop)
    nsamp = n + 61 * hop + 2
    q = rng.integers(-(1 << 23), 1 << 23, (3, nsamp), dtype=np.int64)
    q[0, :8] = [-(1 << 23), (1 << 23) - 1, -1, 0, 1, 255, 256, -65536]
    u = q & 0xFFFFFF
    packed = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=-1).astype(np.uint8).reshape(3, -1)
    xd = torch.from_numpy(packed).cuda()
    a, b, info = _both(S, xd, n, hop, pcm="s24")
    assert info[0] in (17, 19) and torch.equal(a, b), info
    cfg = oracle.OracleConfig.make(n, hop)
    for c in (0, 2):
        ref, _ = oracle.stft(cfg, (q[c] / 8388608.0).astype(np.float32))
        assert_power_close(a[c].cpu().numpy(), ref, "int24 ring ch %d" % c)


@pytest.mark.parametrize("kw", [dict(out="db_u8", window="blackman", scale="inv_n"), dict(out="db"),
                                dict(out="db_u8", window="blackman", scale="inv_n", logfreq_cols=256), dict(out="mag", nyquist=True)])
def test_ring_every_epilogue(S, kw):
    """Output kinds, the Nyquist bin and the log-frequency columns behind the ring loads (N = 2048, hop 735: the
    reference's operating point) are bit-identical to the same epilogues behind the generic loads."""
    import torch
    rng = np.random.default_rng(3)
    x = torch.from_numpy((rng.uniform(-1, 1, (3, 2048 + 123 * 735)) * 0.3).astype(np.float32)).cuda()
    a, b, info = _both(S, x, 2048, 735, **kw)
    assert info[0] == 1 and torch.equal(a, b)


def test_ring_never_reads_past_the_buffer(S):
    """The run's last granule may end inside the caller's buffer end: the kernel then fetches it sample by sample.
    Values check (against the generic loads) for every residue of the buffer length modulo 4, with the input placed at
    the very end of its allocation."""
    import torch
    n, hop = 1024, 301
    for extra in range(0, 8):
        n_samples = n + 37 * hop + extra
        whole = torch.rand(3 * n_samples + 5, device="cuda") * 2 - 1
        x = whole[5:].view(3, n_samples)                   # ends exactly where the allocation's data ends
        a, b, info = _both(S, x, n, hop)
        assert info[0] == 1 and torch.equal(a, b), extra


def test_reference_cadence_rate_and_traffic_shape(S):
    """hop 735 at N = 2048 takes the split-layout ring; hop 300 at N = 1024 the plain one; hop = N/4 and N/2 stay on the
    register sliding window; strided / down-mixed / packed 24-bit input stays on the generic loads."""
    import torch
    x = torch.rand(2, 200_000, device="cuda")
    for n, hop, want in ((2048, 735, 1), (1024, 300, 1), (1024, 256, 2), (1024, 512, 4), (2048, 1500, 3), (2048, 2045, 0),
                         (512, 100, 5), (1024, 700, 0), (4096, 3000, 0), (8192, 6000, 3), (16384, 5000, 5)):   # 5: generic loads of plain float32 rows (N = 512, N >= 16384)
        with S.Engine(fft_size=n, hop=hop) as e:
            e.stft(x)
            assert e.last_launch()[0] == want, (n, hop, e.last_launch())
    with S.Engine(fft_size=1024, hop=300) as e:
        e.stft(x, sample_stride=2)
        assert e.last_launch()[0] == 0
