'use strict';
/* The argument-validation half of the N-API shim (spectrogram_amd/js/napi_shim.c: 400 lines of typed-array arithmetic) and
 * of analyser.js, run under node with AddressSanitizer + UBSan: `make -C oracle check-shim-asan` builds the shim and the
 * host side of libspectro.so with the sanitizers on top of a host-memory stand-in for the device (tests/host/) and points
 * SPECTRO_NODE at that addon.  No transform happens here -- what is exercised is every path on which the shim computes a
 * length, a stride or a pointer from what JavaScript handed it: wrong types, short and long arrays, ragged pushes,
 * multi-channel rows, down-mix rows, log-frequency rows, fftSize changes on a live analyser, the history ring, use after
 * close().  It also runs (without the sanitizers' help) against the real addon on a GPU box.  TEST INFRASTRUCTURE ONLY. */
const path = require('path');
const S = require(path.join(__dirname, '..', 'spectrogram_amd', 'js', 'analyser.js'));

let checks = 0;
function ok(cond, what) {
  checks++;
  if (!cond) { console.error('FAILED: ' + what); process.exit(1); }
}
function throws(fn, ctor, what) {
  checks++;
  try { fn(); } catch (e) {
    if (ctor && !(e instanceof ctor)) { console.error('FAILED: ' + what + ': threw ' + e); process.exit(1); }
    return;
  }
  console.error('FAILED: ' + what + ': did not throw'); process.exit(1);
}

// ---- construction
throws(() => new S.Analyser({ fftSize: 1000 }), RangeError, 'fftSize not a power of two');
throws(() => new S.Analyser({ fftSize: 16 }), RangeError, 'fftSize below 32');
throws(() => new S.Analyser({ fftSize: 65536 }), RangeError, 'fftSize above 32768');
throws(() => new S.Analyser({ logfreqAlign: 'left' }), TypeError, 'bad logfreqAlign');
throws(() => new S.Analyser({ downmixMode: 'surround' }), TypeError, 'bad downmixMode');

for (const opts of [{ fftSize: 2048 }, { fftSize: 256, channels: 3 }, { fftSize: 1024, channels: 2, downmix: 2 },
  { fftSize: 512, channels: 2, downmix: 6, downmixMode: 'speakers' }, { fftSize: 1024, logfreqCols: 256 },
  { fftSize: 32768, logfreqCols: 300, logfreqAlign: 'centre' }]) {
  const a = new S.Analyser(opts);
  const ch = opts.channels || 1, k = opts.downmix || 1, n = opts.fftSize, row = a.rowLength;
  ok(a.frequencyBinCount === n / 2 && row === (opts.logfreqCols || n / 2), 'row length');
  a.smoothingTimeConstant = 0;
  throws(() => { a.smoothingTimeConstant = 1.5; }, RangeError, 'smoothing out of range');
  throws(() => { a.minDecibels = -10; }, RangeError, 'minDecibels >= maxDecibels');
  throws(() => { a.maxDecibels = -200; }, RangeError, 'maxDecibels <= minDecibels');
  a.minDecibels = -90; a.maxDecibels = -20;
  ok(a.connect('dest') === 'dest' && a.disconnect() === undefined, 'graph no-ops');
  // ---- pushes: types, ragged rows, blocks longer than the history
  throws(() => a.push(new Float64Array(ch * k * 8)), TypeError, 'push of a Float64Array');
  throws(() => a.push([0, 1, 2]), TypeError, 'push of a plain array');
  if (ch * k > 1) throws(() => a.push(new Float32Array(ch * k * 8 + 1)), RangeError, 'push of a ragged block');
  for (const nNew of [1, 7, 128, n, n + 3, 3 * 32768 + 5]) {
    const x = new Float32Array(ch * k * nNew);
    for (let i = 0; i < x.length; i++) x[i] = ((i * 7919) % 2001 - 1000) / 1000;
    a.push(x);
    // ---- polls: wrong types, short, exact and over-long arrays; excess entries stay as they were
    throws(() => a.getByteFrequencyData(new Float32Array(row)), TypeError, 'byte poll into a Float32Array');
    throws(() => a.getFloatFrequencyData(new Uint8Array(row)), TypeError, 'float poll into a Uint8Array');
    throws(() => a.getByteTimeDomainData(new Float32Array(n)), TypeError, 'byte time poll into a Float32Array');
    for (const len of [0, 1, row >> 1, row, row + 9]) {
      const b = new Uint8Array(ch * len).fill(0xEE);
      ok(a.getByteFrequencyData(b) === undefined, 'getByteFrequencyData returns undefined');
      if (len > row) for (let c = 0; c < ch; c++) for (let i = row; i < len; i++) ok(b[c * len + i] === 0xEE, 'excess bytes untouched');
      const f = new Float32Array(ch * len).fill(-7);
      a.getFloatFrequencyData(f);
      if (len > row) ok(f[len - 1] === -7, 'excess floats untouched');
    }
    for (const len of [0, 3, n, n + 5]) {
      const t = new Uint8Array(ch * len).fill(0xEE), tf = new Float32Array(ch * len).fill(-7);
      a.getByteTimeDomainData(t);
      a.getFloatTimeDomainData(tf);
      if (len > n) ok(t[len - 1] === 0xEE && tf[len - 1] === -7, 'excess time-domain entries untouched');
      if (len === n && k === 1 && nNew >= 1) {   // the newest sample of channel 0 is the last one pushed
        const last = x[(nNew - 1)];
        ok(ch > 1 || Math.abs(tf[n - 1] - last) < 1e-7, 'time-domain tail is the newest sample');
      }
    }
    // ---- one animation frame as one call: typed-array checks in front of the native call
    const by = new Uint8Array(ch * row);
    throws(() => a.pushAndGetByteFrequencyData(x, new Float32Array(ch * row)), TypeError, 'fused byte call into floats');
    throws(() => a.pushAndGetFloatFrequencyData(new Float64Array(4), new Float32Array(ch * row)), TypeError, 'fused call, bad samples');
    a.pushAndGetByteFrequencyData(x, by);
    a.pushAndGetFloatFrequencyData(x, new Float32Array(ch * (row >> 1)));
  }
  // ---- the history ring
  a.startHistory(5);
  const by = new Uint8Array(ch * row);
  for (let i = 0; i < 7; i++) { a.push(new Float32Array(ch * k * 64)); a.getByteFrequencyData(by); ok(a.historyOffset === (i + 1) % 5, 'ring offset'); }
  ok(a.readHistory().length === ch * 5 * row, 'ring image size');
  throws(() => a.readHistory(new Float32Array(ch * 5 * row)), TypeError, 'ring into a Float32Array');
  throws(() => a.readHistory(new Uint8Array(3)), Error, 'ring into a short array');
  a.pinHistory();
  a.getByteFrequencyData(by);
  ok(a.historyOffset === 0, 'pinned ring stays at row 0');
  // ---- fftSize on a live analyser: samples stay, rows change length
  for (const m of [64, 4096, n]) {
    a.fftSize = m;
    ok(a.frequencyBinCount === m / 2, 'frequencyBinCount follows fftSize');
    a.getByteFrequencyData(new Uint8Array(ch * a.rowLength));
    a.getFloatTimeDomainData(new Float32Array(ch * m));
    ok(a.readHistory().length === ch * 5 * a.rowLength, 'ring follows the row length');
  }
  throws(() => { a.fftSize = 100; }, RangeError, 'bad fftSize on a live analyser');
  ok(a.fftSize === n, 'a refused fftSize leaves the analyser as it was');
  a.close();
  a.close();                                       // idempotent
  throws(() => a.getByteFrequencyData(by), Error, 'use after close');
  throws(() => a.push(new Float32Array(ch * k)), Error, 'push after close');
}

// ---- batch entry: types and shapes
throws(() => S.stft(new Float64Array(2048), { fftSize: 1024, hop: 256 }), TypeError, 'stft of a Float64Array');
ok(S.stft(new Float32Array(100), { fftSize: 1024, hop: 256 }).length === 0, 'stft of a short signal is empty');
const rows = S.stft(new Float32Array(1024 + 5 * 256 + 3), { fftSize: 1024, hop: 256, out: 'db_u8', window: 'blackman', scale: 'inv_n' });
ok(rows instanceof Uint8Array && rows.length === 6 * 512, 'byte rows of a batch call');
throws(() => S.stft(new Float32Array(4096), { fftSize: 1000 }), RangeError, 'stft with a bad fftSize');
console.log(JSON.stringify({ ok: true, checks }));
