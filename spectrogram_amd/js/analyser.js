'use strict';
/*
 * analyser.js -- an AnalyserNode-shaped object backed by the MI355X engine (spectro.node).
 *
 * Drop-in for the object Player creates with context.createAnalyser() (UI/player.js:7) and
 * hands to AnalyserView.setAnalyserNode() (UI/spectrogram.js:121-124, 3D/visualizer.js:567).
 * Same members, same meaning, same error behaviour as the platform object those call sites use:
 *   fftSize (UI/player.js:10)                  power of two 32..32768, else RangeError (IndexSizeError)
 *   frequencyBinCount (3D/visualizer.js:299)    read-only, fftSize / 2
 *   smoothingTimeConstant (UI/player.js:11; 3D/visualizer.js:351,357,362)   0..1
 *   minDecibels / maxDecibels                   defaults -100 / -30 (the reference never sets them)
 *   getByteFrequencyData(Uint8Array) (:352,358), getFloatFrequencyData(Float32Array),
 *   getByteTimeDomainData(Uint8Array) (:363)    write min(array.length, bins) values in place, return undefined
 *   connect() / disconnect() (UI/player.js:25-26,158-167)   no-ops: a native object is not a graph node
 * plus push(Float32Array), the feed that replaces the graph edge mix.connect(analyser).
 */
const path = require('path');

let native = null;
function loadNative() {
  // (SPECTRO_NODE: another build of the same addon -- the sanitizer build of `make -C oracle check-shim-asan`; read here in
  // the wrapper, like SPECTRO_LIB in the Python binding, never by the native code)
  if (!native) native = require(process.env.SPECTRO_NODE || path.join(__dirname, 'spectro.node'));  // throws if not built: no JS fallback
  return native;
}

const WINDOW = { hann: 0, blackman: 1, rect: 2 };
const ALIGN = { centre: 0, vertex: 1 };

class Analyser {
  constructor(options) {
    const o = options || {};
    this._native = loadNative();
    this._window = WINDOW[o.window || 'blackman'];
    this._channels = o.channels || 1;
    // beyond the platform object: rows of `logfreqCols` log-frequency columns (what the reference's shader samples out
    // of each row, sonogram-vertex.shader:51-52) instead of fftSize/2 bins; `downmix` = K interleaved input channels
    // per analyser, averaged on the way in (what the `mix` GainNode does to a stereo source, UI/player.js:25)
    this._logfreqCols = (o.logfreqCols || 0) | 0;
    // where the columns sample the row: 'vertex' (default) = the mesh of the reference's default display, the 3-D
    // sonogram, u = x / (sonogram3DWidth - 1) (3D/visualizer.js:52,54,213-214); 'centre' = pixel centres, what its
    // 2-D sonogram's fragment shader does (sonogram-fragment.shader:16)
    this._logfreqAlign = ALIGN[o.logfreqAlign || 'vertex'];
    if (this._logfreqAlign === undefined) throw new TypeError("logfreqAlign must be 'vertex' or 'centre'");
    this._downmix = (o.downmix || 0) | 0;
    // how the K channels are mixed: 'mean' (the reference's `mix` node on its mono / stereo sources) or 'speakers', the W3C
    // mono down-mix of an AnalyserNode's input (5.1: sqrt(1/2) (L + R) + C + 0.5 (SL + SR))
    this._downmixMode = { mean: 0, speakers: 1 }[o.downmixMode || 'mean'];
    if (this._downmixMode === undefined) throw new TypeError("downmixMode must be 'mean' or 'speakers'");
    this._tau = 0.8;         // W3C default
    this._minDb = -100;
    this._maxDb = -30;
    this._plan(o.fftSize || 2048);
  }

  _plan(n) {
    n = Number(n);
    if (!Number.isInteger(n) || n < 32 || n > 32768 || (n & (n - 1)) !== 0) {
      throw new RangeError('IndexSizeError: fftSize must be a power of two between 32 and 32768');
    }
    if (this._h) {
      // a live analyser keeps its samples (the platform holds the latest 32768 whatever fftSize is); the smoothing
      // state, whose length is fftSize / 2, starts again
      this._native.setFftSize(this._h, n);
    } else {
      this._h = this._native.create(n, this._window, this._channels, this._tau, this._minDb, this._maxDb,
        this._logfreqCols, this._downmix, this._logfreqAlign, this._downmixMode);
    }
    this._fftSize = n;
  }

  /** Frees the native engine; the object must not be used afterwards. */
  close() {
    if (this._h) this._native.destroy(this._h);
    this._h = null;
  }

  get fftSize() { return this._fftSize; }
  set fftSize(v) { this._plan(v); }                       // re-plans; smoothing restarts, the sample history stays
  get frequencyBinCount() { return this._fftSize / 2; }
  /** values per channel a frequency poll writes: frequencyBinCount, or logfreqCols when that option is set */
  get rowLength() { return this._logfreqCols > 0 ? this._logfreqCols : this._fftSize / 2; }
  get smoothingTimeConstant() { return this._tau; }
  set smoothingTimeConstant(v) {
    v = Number(v);
    if (!(v >= 0 && v <= 1)) throw new RangeError('IndexSizeError: smoothingTimeConstant must be in [0, 1]');
    this._native.setSmoothing(this._h, v);
    this._tau = v;
  }
  get minDecibels() { return this._minDb; }
  set minDecibels(v) {
    v = Number(v);
    if (!(v < this._maxDb)) throw new RangeError('IndexSizeError: minDecibels must be < maxDecibels');
    this._native.setDbRange(this._h, v, this._maxDb);
    this._minDb = v;
  }
  get maxDecibels() { return this._maxDb; }
  set maxDecibels(v) {
    v = Number(v);
    if (!(v > this._minDb)) throw new RangeError('IndexSizeError: maxDecibels must be > minDecibels');
    this._native.setDbRange(this._h, this._minDb, v);
    this._maxDb = v;
  }

  connect(dest) { return dest; }
  disconnect() {}

  /** Float32Array of channels * nNew samples (planar; with downmix = K each channel's row is nNew * K interleaved). */
  push(samples) {
    if (!(samples instanceof Float32Array)) throw new TypeError('push expects a Float32Array');
    const k = this._downmix > 1 ? this._downmix : 1;
    if (samples.length % (this._channels * k) !== 0) {
      throw new RangeError('push expects channels * nNew' + (k > 1 ? ' * downmix' : '') + ' samples (a whole number of rows)');
    }
    this._native.push(this._h, samples, samples.length / (this._channels * k));
  }

  /**
   * One animation frame of the reference as one call: the audio that arrived since the last frame reaches the analyser,
   * then getByteFrequencyData(array) (UI/spectrogram.js:153-161 -> 3D/visualizer.js:346-358) -- ONE kernel launch and
   * one synchronise when the block is an even number of samples up to one window (any other block: push, then poll).
   */
  pushAndGetByteFrequencyData(samples, array) {
    if (!(array instanceof Uint8Array)) throw new TypeError('pushAndGetByteFrequencyData expects a Uint8Array');
    this._native.pushPollU8(this._h, samples, this._rows(samples), array, this._channels);
  }
  pushAndGetFloatFrequencyData(samples, array) {
    if (!(array instanceof Float32Array)) throw new TypeError('pushAndGetFloatFrequencyData expects a Float32Array');
    this._native.pushPollF32(this._h, samples, this._rows(samples), array, this._channels);
  }
  _rows(samples) {
    if (!(samples instanceof Float32Array)) throw new TypeError('push expects a Float32Array');
    const k = this._downmix > 1 ? this._downmix : 1;
    if (samples.length % (this._channels * k) !== 0) {
      throw new RangeError('push expects channels * nNew' + (k > 1 ? ' * downmix' : '') + ' samples (a whole number of rows)');
    }
    return samples.length / (this._channels * k);
  }

  getByteFrequencyData(array) {
    if (!(array instanceof Uint8Array)) throw new TypeError('getByteFrequencyData expects a Uint8Array');
    this._native.pollU8(this._h, array, this._channels);
  }
  getFloatFrequencyData(array) {
    if (!(array instanceof Float32Array)) throw new TypeError('getFloatFrequencyData expects a Float32Array');
    this._native.pollF32(this._h, array, this._channels);
  }
  getByteTimeDomainData(array) {
    if (!(array instanceof Uint8Array)) throw new TypeError('getByteTimeDomainData expects a Uint8Array');
    this._native.pollTimeU8(this._h, array, this._channels);
  }
  getFloatTimeDomainData(array) {
    if (!(array instanceof Float32Array)) throw new TypeError('getFloatTimeDomainData expects a Float32Array');
    this._native.pollTimeF32(this._h, array, this._channels);
  }

  // ---- the history texture AnalyserView keeps (3D/visualizer.js:60,293-331,390-417), on the device:
  // after startHistory(h) every getByteFrequencyData() also writes its row at historyOffset (the reference's
  // `yoffset`) and advances it modulo h -- or keeps writing row 0 when pinned, as the reference's 2-D
  // frequency / waveform modes do (:385-390).  readHistory() returns the ring unrolled, oldest row first.
  startHistory(height) {
    this._ringHeight = (height === undefined ? 256 : height) | 0;
    this._native.ringBegin(this._h, this._ringHeight);
  }
  pinHistory(pinned) { this._native.ringSetPinned(this._h, pinned === undefined ? true : !!pinned); }
  get historyOffset() { return this._native.ringOffset(this._h); }
  readHistory(into) {
    const n = this._channels * this._ringHeight * this.rowLength;
    const out = into || new Uint8Array(n);
    if (!(out instanceof Uint8Array)) throw new TypeError('readHistory expects a Uint8Array');
    this._native.ringRead(this._h, out);
    return out;
  }
}

/** Batch STFT of a mono Float32Array on the GPU: Float32Array (or Uint8Array) of frames * fftSize/2. */
function stft(pcm, opts) {
  const o = opts || {};
  const OUT = { power: 0, mag: 1, db: 2, db_u8: 3 };
  return loadNative().stft(pcm, o.fftSize || 1024, o.hop || 256, WINDOW[o.window || 'hann'],
    o.scale === 'inv_n' ? 1 : 0, OUT[o.out || 'power']);
}

module.exports = { Analyser, stft };
