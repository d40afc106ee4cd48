"""The accuracy yardstick of the stated float32 tolerance -- TEST INFRASTRUCTURE ONLY (like everything under oracle/).

north_star: "<= 1e-4 max relative error vs the JS reference", the reference being a float32 radix-2 Cooley-Tukey FFT in
JavaScript.  Two such FFTs exist here, both textbook iterative radix-2 on Float32 storage:
  * twin "js"  -- oracle/stft_oracle.js run on Float32Array, restated in C bit for bit (so_stft_f32js): a JS engine
                  evaluates each butterfly in double and rounds once per store;
  * twin "f32" -- the same algorithm with every operation in float32 (so_stft_f32): what the JS code computes with
                  Math.fround around each operation, or any C / WebAssembly float implementation.
error_stats() measures an implementation and both twins against the float64 oracle ON THE SAME FRAMES, under the same
normalisations, so that the tolerance statement can be read against what a float32 radix-2 FFT itself achieves instead
of against a floor picked to make a sweep pass.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import oracle as O

FLOORS = (1e-5, 2e-5)


def _chunks(frames, workers):
    per = (frames + workers - 1) // workers
    return [(i * per, min(per, frames - i * per)) for i in range(workers) if i * per < frames]


def error_stats(x, n, hop, window, candidates, workers=8):
    """x: one channel of float32 samples; candidates: {name: frames x N/2 float array} (e.g. the HIP kernel's output).
    Returns {name: {"F=1e-05": max |dP| / max(P, F Pmax), ..., "peak_max": max |dP| / Pmax, "peak_rms": rms of the same}}
    for every candidate and for the two twins, P = the float64 oracle's power of the same frames."""
    x = np.ascontiguousarray(x, np.float32)
    cfg = O.OracleConfig.make(n, hop, window=window)
    frames = O.num_frames(len(x), n, hop)
    names = list(candidates) + ["twin_js", "twin_f32"]
    O.lib()

    def work(span):
        f0, nf = span
        seg = x[f0 * hop: f0 * hop + (nf - 1) * hop + n]
        ref, _ = O.stft(cfg, seg)
        mx = ref.max(axis=1, keepdims=True)
        mx = np.where(mx > 0, mx, 1.0)
        out = {}
        for name in names:
            if name == "twin_js":
                got = O.stft_f32js(cfg, seg)
            elif name == "twin_f32":
                got = O.stft_f32(cfg, seg)
            else:
                got = np.asarray(candidates[name][f0:f0 + nf])
            d = np.abs(got.astype(np.float64) - ref)
            r = {"F=%g" % f: float((d / np.maximum(ref, f * mx)).max()) for f in FLOORS}
            dn = d / mx
            r["peak_max"] = float(dn.max())
            r["sumsq"] = float((dn * dn).sum())
            out[name] = r
        return out, nf * (n // 2)

    with ThreadPoolExecutor(workers) as ex:
        parts = list(ex.map(work, _chunks(frames, workers)))
    total = sum(cnt for _, cnt in parts)
    res = {}
    for name in names:
        r = {k: max(p[name][k] for p, _ in parts) for k in parts[0][0][name] if k != "sumsq"}
        r["peak_rms"] = float(np.sqrt(sum(p[name]["sumsq"] for p, _ in parts) / total))
        res[name] = r
    res["bins"] = total
    return res
