#!/usr/bin/env python3
"""WAV file -> the picture the reference's sonogram modes draw, as a PGM image.

What the reference does per animation frame (3D/visualizer.js:346-417, sonogram-vertex.shader:51-52): poll
getByteFrequencyData into a row of N/2 bytes, keep the last 256 rows as a texture, sample it along
x = 256^(u-1) (log-frequency).  Here the whole file goes through one batch call with the same numerics
(Blackman window, 1/N scale, dB mapped to bytes over [-100, -30] dB) and the log-frequency columns fused
into the STFT epilogue; rows are time, columns are log-frequency, as in the texture.

    python examples/wav_to_spectrogram.py in.wav out.pgm [--fft 2048] [--hop 735] [--cols 256] [--align vertex|centre] [--linear]

Needs an MI355X (there is no CPU path).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_amd import wav  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("wav")
    ap.add_argument("pgm")
    ap.add_argument("--fft", type=int, default=2048, help="fftSize (UI/player.js:10)")
    ap.add_argument("--hop", type=int, default=735, help="samples between rows (44.1 kHz / 60 fps = 735)")
    ap.add_argument("--cols", type=int, default=256, help="log-frequency columns")
    ap.add_argument("--align", default="vertex", choices=["vertex", "centre"],
                    help="where the columns sample the row: the 3-D sonogram's mesh vertices u = x / (cols - 1), the reference's "
                         "default display (3D/visualizer.js:52,213-214), or pixel centres like its 2-D sonogram's fragment shader")
    ap.add_argument("--linear", action="store_true", help="N/2 linear bins per row instead of log-frequency columns")
    a = ap.parse_args(argv)
    data = open(a.wav, "rb").read()
    kw = dict(fft_size=a.fft, hop=a.hop, window="blackman", scale="inv_n", out="db_u8")
    if not a.linear:
        kw["logfreq_cols"] = a.cols
        kw["logfreq_align"] = a.align
    rows, info = wav.stft_wav(data, mono=True, **kw)       # [frames, cols] uint8; a stereo file is mixed to mono
    with open(a.pgm, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (rows.shape[1], rows.shape[0]))
        f.write(rows.tobytes())
    print("%s: %d Hz, %d ch, %s, %d samples -> %d rows x %d columns -> %s" % (
        a.wav, info["sample_rate"], info["channels"], info["pcm"], info["n_frames"], rows.shape[0], rows.shape[1], a.pgm))
    return rows, info


if __name__ == "__main__":
    main()
