#!/usr/bin/env python3
"""Writes the tiny WAV fixtures under tests/golden/wav/ (run in the build container; committed with its output).

Every file holds the same seeded signal (two sines + noise, 2400 frames at 48 kHz) in another sample
encoding / header flavour; tests/golden/wav/expected.npz holds the float64 samples each one decodes to
([frames, channels], value = integer / 2^(bits-1) for PCM).  tests/test_wav.py checks libspectro's header
parser against the recorded header fields and the GPU STFT of the file bytes against the oracle's STFT of
the expected samples.  These are script-made data, not reference files (the reference holds MP3s only).
"""
import os
import struct

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "wav")
SR, N = 48000, 2400


def chunk(tag, body):
    return tag + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")


def fmt_chunk(tag, ch, bits, extensible=False):
    block = ch * bits // 8
    base = struct.pack("<HHIIHH", 0xFFFE if extensible else tag, ch, SR, SR * block, block, bits)
    if extensible:
        guid_tail = bytes.fromhex("000000001000800000aa00389b71")
        base += struct.pack("<HHI", 22, bits, 0x3 if ch == 2 else 0x4) + struct.pack("<H", tag) + guid_tail
    return chunk(b"fmt ", base)


def riff(*chunks):
    body = b"WAVE" + b"".join(chunks)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(0x5EED0F4)
    t = np.arange(N) / SR
    sig = np.stack([0.45 * np.sin(2 * np.pi * 1000 * t) + 0.2 * np.sin(2 * np.pi * 7350 * t + 0.3),
                    0.35 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.standard_normal(N)], axis=1)
    sig = np.clip(sig + 0.01 * rng.standard_normal(sig.shape), -0.999, 0.999)
    expected = {}

    def pcm_bytes(x, bits):
        q = np.round(x * (1 << (bits - 1))).astype(np.int64)
        q = np.clip(q, -(1 << (bits - 1)), (1 << (bits - 1)) - 1)
        if bits == 16:
            return q.astype("<i2").tobytes(), q / 32768.0
        b = np.empty(q.shape + (3,), np.uint8)
        u = q & 0xFFFFFF
        b[..., 0], b[..., 1], b[..., 2] = u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF
        return b.tobytes(), q / 8388608.0

    def put(name, data, exp, **fields):
        open(os.path.join(OUT, name + ".wav"), "wb").write(data)
        expected[name] = exp
        for k, v in fields.items():
            expected[name + "__" + k] = np.int64(v)

    mono = sig[:, :1]
    raw, exp = pcm_bytes(mono, 16)
    put("s16_mono", riff(fmt_chunk(1, 1, 16), chunk(b"data", raw)), exp, channels=1, bits=16, offset=44)
    raw, exp = pcm_bytes(sig, 16)
    put("s16_stereo_list", riff(fmt_chunk(1, 2, 16), chunk(b"LIST", b"INFOISFT\x05\0\0\0test\0\0"), chunk(b"data", raw)), exp,
        channels=2, bits=16, offset=44 + 8 + 18)
    raw, exp = pcm_bytes(sig, 24)
    put("s24_stereo", riff(fmt_chunk(1, 2, 24), chunk(b"data", raw)), exp, channels=2, bits=24, offset=44)
    raw, exp = pcm_bytes(mono, 24)
    put("s24_mono_ext", riff(fmt_chunk(1, 1, 24, extensible=True), chunk(b"data", raw)), exp, channels=1, bits=24, offset=68)
    f = sig.astype("<f4")
    put("f32_stereo", riff(fmt_chunk(3, 2, 32), chunk(b"fact", struct.pack("<I", N)), chunk(b"data", f.tobytes())),
        f.astype(np.float64), channels=2, bits=32, offset=44 + 12)
    # a streamed file: data size 0xFFFFFFFF, samples run to the end of the image
    raw, exp = pcm_bytes(mono, 16)
    body = b"WAVE" + fmt_chunk(1, 1, 16) + b"data" + struct.pack("<I", 0xFFFFFFFF) + raw
    put("s16_mono_streamed", b"RIFF" + struct.pack("<I", 0xFFFFFFFF) + body, exp, channels=1, bits=16, offset=44)
    # rejected: 8-bit PCM (valid header, no ingest path)
    put("u8_mono_unsupported", riff(fmt_chunk(1, 1, 8), chunk(b"data", bytes(N))), np.zeros((N, 1)), channels=1, bits=8, offset=44)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
