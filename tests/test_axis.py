"""The one reference-pinned fixture this path allows: the reference's own log-scale code (util/util.js:32-43),
run under node in the build container by oracle/make_reference_fixtures.js, against libspectro's sg_axis_* helpers;
plus the column <-> bin <-> Hz map of the log-frequency epilogue against the oracle's lookup."""
import json
import os

import numpy as np

from conftest import GOLDEN


def _ref():
    return json.load(open(os.path.join(GOLDEN, "ref_util_logscale.json")))


def test_log_scale_matches_reference_code():
    from spectrogram_amd import axis
    r = _ref()
    s = axis.LogScale(*r["setLogScale"])
    assert abs(s.a - r["a"]) <= 1e-13 * r["a"] and abs(s.b - r["b"]) <= 1e-13 * abs(r["b"])
    for x, y in zip(r["x"], r["lin2log"]):
        assert abs(s.lin2log(x) - y) <= 1e-12 * abs(y), (x, y)
    for y, x in zip(r["y"], r["log2lin"]):
        assert abs(s.log2lin(y) - x) <= 1e-9 * max(1.0, abs(x)), (y, x)
    # round trip and the end points the page relies on (20 Hz .. 20 kHz over the legend)
    assert abs(s.lin2log(20) - 20) < 1e-9 and abs(s.lin2log(20000) - 20000) < 1e-6
    for x in (25.0, 999.0, 15000.0):
        assert abs(s.log2lin(s.lin2log(x)) - x) < 1e-7


def test_legend_positions_match_reference_arithmetic():
    """UI/spectrogram.js:191-212 (yToFreq): percentFromBottom -> freqStart + span * p -> lin2log; values recorded
    from the reference's lin2log for two canvas heights."""
    from spectrogram_amd import axis
    s = axis.LogScale(20, 20, 20000, 20000)
    for e in _ref()["legend"]:
        p = 1 - (e["y"] - 30) / (e["height"] - 30)
        assert abs(s.lin2log(20 + 19980 * p) - e["freq"]) <= 1e-12 * e["freq"]


def test_logfreq_column_map_matches_the_epilogue_table(oracle=None):
    """sg_logfreq_col_bin is the sampling position the kernel's table is built from: resampling a ramp row
    (value = bin index) through the oracle's lookup must give exactly that position (clamped to the row)."""
    import oracle as O
    from spectrogram_amd import axis
    for n, cols in ((1024, 256), (2048, 256), (256, 64)):
        m = n // 2
        got = O.logfreq_resample(np.arange(m, dtype=np.float64), cols)
        pos = np.array([axis.logfreq_col_bin(n, cols, c) for c in range(cols)])
        assert np.abs(got - np.clip(pos, 0, m - 1)).max() < 1e-9
        # Hz: column centres run from ~Nyquist/256 to ~Nyquist on a 44.1 kHz stream (the bundled clips' rate)
        hz = [axis.logfreq_col_hz(n, cols, c, 44100.0) for c in (0, cols - 1)]
        assert hz[0] < 44100 / 2 / 200 and 0.95 * 22050 < hz[1] < 22050
    assert axis.bin_hz(2048, 1024, 48000.0) == 24000.0


def test_vertex_alignment_is_the_reference_mesh():
    """The reference's default display, the 3-D sonogram, samples the row at its mesh's texture coordinates
    u = x / (sonogram3DWidth - 1), x = 0 .. 255 (3D/visualizer.js:54,213-214; lookup in sonogram-vertex.shader:51-52):
    at fftSize 2048 column 0 lands on texel position 3.5 (x = 1/256 of 1024 texels, centres at i + 0.5) and column 255
    on u = 1, past the last texel centre, which CLAMP_TO_EDGE (:313) turns into the last texel."""
    import oracle as O
    from spectrogram_amd import axis
    n, cols, m = 2048, 256, 1024
    assert abs(axis.logfreq_col_bin(n, cols, 0, "vertex") - 3.5) < 1e-12
    assert abs(axis.logfreq_col_bin(n, cols, 255, "vertex") - 1023.5) < 1e-9
    assert axis.logfreq_col_hz(n, cols, 255, 44100.0, "vertex") == axis.bin_hz(n, 1023, 44100.0)   # the clamped last texel
    ramp = np.arange(m, dtype=np.float64)
    got = O.logfreq_resample(ramp, cols, "vertex")
    pos = np.array([axis.logfreq_col_bin(n, cols, c, "vertex") for c in range(cols)])
    assert np.abs(got - np.clip(pos, 0, m - 1)).max() < 1e-9
    assert got[0] == 3.5 and got[-1] == m - 1
    # the two conventions really are different sets of samples (the build only had pixel centres)
    centre = O.logfreq_resample(ramp, cols, "centre")
    assert np.abs(centre - got).max() > 5.0
    # the numpy restatement the GPU tests use agrees with the oracle's loop for both
    from conftest import logfreq_blend
    for align in ("centre", "vertex"):
        assert np.abs(logfreq_blend(ramp, cols, align) - O.logfreq_resample(ramp, cols, align)).max() < 1e-9
    assert O.logfreq_resample(ramp, 1, "vertex")[0] == 3.5    # a single column sits at u = 0


def test_vertex_columns_sit_on_the_reference_meshs_own_texture_coordinates():
    """Reference-pinned (the second sliver of this path that can execute here): the 3-D sonogram's mesh loop,
    3D/visualizer.js:203-218, run under node from the reference's own text by oracle/make_reference_fixtures.js, gives the
    texture coordinate gTexCoord0.x of each of its 256 columns (a Float32Array: fround(x / 255)); the vertex shader samples
    the analyser row at pow(256, that - 1) (sonogram-vertex.shader:51-52).  SG_LOGFREQ_VERTEX must sample there: same u to
    float32 precision, same texel position to a few 1e-4 of a texel, for every column."""
    from spectrogram_amd import axis
    r = json.load(open(os.path.join(GOLDEN, "ref_mesh_texcoords.json")))
    cols = r["sonogram3DWidth"]
    assert cols == 256 and r["texcoord_x_same_in_every_row"]
    u_ref = np.array(r["texcoord_x"], np.float64)
    assert u_ref[0] == 0.0 and u_ref[-1] == 1.0
    assert np.array_equal(u_ref.astype(np.float32), (np.arange(cols) / np.float32(cols - 1)).astype(np.float32))
    n = 2048                                                  # UI/player.js:10
    m = n // 2
    pos_ref = np.power(256.0, u_ref - 1.0) * m - 0.5          # where the shader's x lands in texel units (centres at i + 0.5)
    pos = np.array([axis.logfreq_col_bin(n, cols, c, "vertex") for c in range(cols)])
    assert np.abs(pos - pos_ref).max() < 5e-4                 # float32 u: 6e-8 x ln(256) x 1024 texels
    assert np.abs(u_ref - np.arange(cols) / (cols - 1)).max() <= 2.0 ** -24
    # pixel centres are NOT where this mesh samples: up to several texels away
    centre = np.array([axis.logfreq_col_bin(n, cols, c, "centre") for c in range(cols)])
    assert np.abs(centre - pos_ref).max() > 5.0
