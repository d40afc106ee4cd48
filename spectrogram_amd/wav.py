"""WAV ingest: the host-side step in front of the hot path.

The reference obtains samples through `decodeAudioData` (util/util.js:2-28); for uncompressed
files the equivalent here is: parse the RIFF header (libspectro's sg_wav_parse, plain C), then hand
the sample bytes to the STFT exactly as they lie in the file -- int16 / packed int24 / float32 are
converted inside the kernel's framing load, and a multi-channel file is mixed to mono there too
(what an AnalyserNode does to its input).  MP3 decode stays out of scope.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .engine import Engine

_FMT_NAME = {L.PCM_F32: "f32", L.PCM_S16: "s16", L.PCM_S24: "s24"}


def parse(data):
    """bytes-like WAV image -> dict(pcm, channels, sample_rate, bits_per_sample, data_offset, data_bytes, n_frames)."""
    buf = np.frombuffer(data, np.uint8)
    info = L.SgWavInfo()
    L.check(L.load().sg_wav_parse(buf.ctypes.data, buf.size, C.byref(info)))
    return {"pcm": _FMT_NAME[info.pcm_format], "channels": info.channels, "sample_rate": info.sample_rate,
            "bits_per_sample": info.bits_per_sample, "data_offset": info.data_offset,
            "data_bytes": info.data_bytes, "n_frames": info.n_frames}


def stft_wav(data, mono=True, **engine_kw):
    """STFT of a WAV image.  mono=True: one row set, the mean of the file's channels (fused down-mix);
    mono=False: [channels, frames, bins], each interleaved channel analysed on its own."""
    w = parse(data)
    ch, n = w["channels"], w["n_frames"]
    bps = w["bits_per_sample"] // 8
    raw = np.frombuffer(data, np.uint8, count=n * ch * bps, offset=w["data_offset"])
    dt = {"f32": np.float32, "s16": np.int16, "s24": np.uint8}[w["pcm"]]
    samples = raw if w["pcm"] == "s24" else raw.view(dt)
    if mono or ch == 1:
        with Engine(pcm=w["pcm"], downmix=ch if ch > 1 else 0, **engine_kw) as e:
            return e.stft_host(samples), w
    per = []
    with Engine(pcm=w["pcm"], **engine_kw) as e:
        unit = 3 if w["pcm"] == "s24" else 1
        for c in range(ch):   # channel c of an interleaved file: start at sample c, stride = channels
            per.append(e.stft_host(samples[c * unit:], sample_stride=ch))
    return np.stack(per), w
