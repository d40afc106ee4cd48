'use strict';
/*
 * stft_oracle.js -- plain-JS restatement of the STFT path: TEST INFRASTRUCTURE ONLY, PARITY
 * UNPINNED (see oracle/stft_oracle.h for what that means and why).
 *
 * The reference has no JS FFT (its analyser is the browser's); this is "the per-frame FFT a JS
 * front-end would run" that BASELINE.md section 4 (B1) asks to be timed beside the GPU, and the
 * checker for the N-API path.  Textbook iterative radix-2 decimation in time, periodic Hann /
 * Blackman window, |X|^2.  Array type is a parameter: Float64Array for checking, Float32Array for
 * the timed baseline.
 */

function makeWindow(kind, n) {
  const w = new Float64Array(n);
  for (let i = 0; i < n; i++) {
    const a = 2 * Math.PI * i / n;
    w[i] = kind === 'hann' ? 0.5 - 0.5 * Math.cos(a)
      : kind === 'blackman' ? 0.42 - 0.5 * Math.cos(a) + 0.08 * Math.cos(2 * a) : 1;
  }
  return w;
}

function makePlan(n, ArrayT) {
  const rev = new Uint32Array(n);
  for (let i = 1, j = 0; i < n; i++) {
    let bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    rev[i] = j;
  }
  const cos = new ArrayT(n / 2), sin = new ArrayT(n / 2);
  for (let i = 0; i < n / 2; i++) { cos[i] = Math.cos(2 * Math.PI * i / n); sin[i] = -Math.sin(2 * Math.PI * i / n); }
  return { n, rev, cos, sin };
}

/* in-place forward FFT on (re, im) */
function fft(plan, re, im) {
  const n = plan.n, rev = plan.rev, cos = plan.cos, sin = plan.sin;
  for (let i = 0; i < n; i++) {
    const j = rev[i];
    if (i < j) { let t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
  }
  for (let len = 2; len <= n; len <<= 1) {
    const half = len >> 1, step = n / len;
    for (let i = 0; i < n; i += len) {
      for (let j = 0, k = 0; j < half; j++, k += step) {
        const wr = cos[k], wi = sin[k], a = i + j, b = a + half;
        const xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - xr; im[b] = im[a] - xi;
        re[a] += xr; im[a] += xi;
      }
    }
  }
}

/* power spectrogram of mono pcm (Float32Array): ArrayT of frames * n/2 */
function stftPower(pcm, n, hop, opts) {
  const o = opts || {};
  const ArrayT = o.arrayType || Float64Array;
  const w = makeWindow(o.window || 'hann', n);
  const scale = o.scale === 'inv_n' ? 1 / n : 1;
  const wt = new ArrayT(n);
  for (let i = 0; i < n; i++) wt[i] = w[i] * scale;
  const plan = makePlan(n, ArrayT);
  const frames = pcm.length < n ? 0 : Math.floor((pcm.length - n) / hop) + 1;
  const bins = n / 2;
  const out = o.out || new ArrayT(frames * bins);
  const re = new ArrayT(n), im = new ArrayT(n);
  for (let f = 0; f < frames; f++) {
    const base = f * hop;
    for (let i = 0; i < n; i++) { re[i] = wt[i] * pcm[base + i]; im[i] = 0; }
    fft(plan, re, im);
    const ob = f * bins;
    for (let k = 0; k < bins; k++) out[ob + k] = re[k] * re[k] + im[k] * im[k];
  }
  return out;
}

/* AnalyserNode byte mapping of one frame of power (X already scaled by 1/N): W3C getByteFrequencyData */
function powerToBytes(p, minDb, maxDb) {
  const out = new Uint8Array(p.length);
  for (let k = 0; k < p.length; k++) {
    const y = 10 * Math.log10(p[k]);
    const v = Math.floor(255 / (maxDb - minDb) * (y - minDb));
    out[k] = v > 255 ? 255 : (v > 0 ? v : 0);
  }
  return out;
}

module.exports = { makeWindow, makePlan, fft, stftPower, powerToBytes };
