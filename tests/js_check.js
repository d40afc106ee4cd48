'use strict';
/* Driver used by tests/test_js.py.  Modes:
 *   oracle <pcm.f32> <fft> <hop> <out.f64>      JS oracle power -> file (CPU)
 *   oracle32 <pcm.f32> <fft> <hop> <out.f32> [window]   the same on Float32Array (north_star's float32 radix-2 JS FFT) -> file
 *   gpu    <pcm.f32> <fft> <hop>                native stft() and Analyser vs the JS oracle -> JSON on stdout
 */
const fs = require('fs');
const path = require('path');
const O = require(path.join(__dirname, '..', 'oracle', 'stft_oracle.js'));
const mode = process.argv[2];
const buf = fs.readFileSync(process.argv[3]);
const pcm = new Float32Array(buf.buffer, buf.byteOffset, buf.length / 4);
const n = parseInt(process.argv[4], 10), hop = parseInt(process.argv[5], 10);

if (mode === 'oracle') {
  const p = O.stftPower(pcm, n, hop);
  fs.writeFileSync(process.argv[6], Buffer.from(p.buffer));
} else if (mode === 'oracle32') {
  const p = O.stftPower(pcm, n, hop, { arrayType: Float32Array, window: process.argv[7] || 'hann' });
  fs.writeFileSync(process.argv[6], Buffer.from(p.buffer));
} else {
  const S = require(path.join(__dirname, '..', 'spectrogram_amd', 'js', 'analyser.js'));
  const res = {};
  // batch: GPU power vs JS oracle, tolerance statement of DESIGN.md section 6
  const ref = O.stftPower(pcm, n, hop);
  const got = S.stft(pcm, { fftSize: n, hop });
  const bins = n / 2, frames = ref.length / bins;
  let worst = 0;
  for (let f = 0; f < frames; f++) {
    let mx = 0;
    for (let k = 0; k < bins; k++) mx = Math.max(mx, ref[f * bins + k]);
    for (let k = 0; k < bins; k++) {
      const r = ref[f * bins + k];
      worst = Math.max(worst, Math.abs(got[f * bins + k] - r) / Math.max(r, 2e-5 * mx));
    }
  }
  res.batch_len_ok = got.length === ref.length;
  res.batch_worst_rel = worst;
  // the object the reference's draw loop would poll (UI/player.js:10-11, 3D/visualizer.js:357-358)
  const a = new S.Analyser();
  a.fftSize = 2048;
  a.smoothingTimeConstant = 0;
  res.bin_count = a.frequencyBinCount;
  const bytes = new Uint8Array(a.frequencyBinCount);
  a.push(pcm.subarray(0, 2048));
  res.ret_undefined = a.getByteFrequencyData(bytes) === undefined;
  const pw = O.stftPower(pcm.subarray(0, 2048), 2048, 2048, { window: 'blackman', scale: 'inv_n' });
  const exp = O.powerToBytes(pw, -100, -30);
  // every byte inside the image of the stated power tolerance (tests/conftest.py byte_bounds, restated): the byte
  // map is monotone, so [P - tol, P + tol] maps to an interval of bytes -- one value for all but ~0.4 % of them
  let pmax = 0, outside = 0, wide = 0;
  for (let k = 0; k < 1024; k++) pmax = Math.max(pmax, pw[k]);
  const byteOf = (p, shift) => {
    const v = Math.floor(255 / 70 * (10 * Math.log10(Math.max(p, 0)) + 100) + shift);
    return v > 255 ? 255 : (v > 0 ? v : 0);
  };
  for (let k = 0; k < 1024; k++) {
    const tol = 1e-4 * Math.max(pw[k], 2e-5 * pmax) + 1e-30;
    const lo = byteOf(pw[k] - tol, -2e-4), hi = byteOf(pw[k] + tol, 2e-4);
    if (bytes[k] < lo || bytes[k] > hi || exp[k] < lo || exp[k] > hi) outside++;
    if (hi > lo) wide++;
  }
  res.bytes_outside_tolerance_interval = outside;
  res.byte_intervals_wider_than_one = wide / 1024;
  const shortArr = new Uint8Array(16).fill(9);
  a.getByteFrequencyData(shortArr);
  res.short_ok = shortArr.every((v, i) => v === bytes[i]);
  const longArr = new Uint8Array(1500).fill(9);
  a.getByteFrequencyData(longArr);
  res.long_tail_untouched = longArr.subarray(1024).every((v) => v === 9);
  const td = new Uint8Array(1024);
  a.getByteTimeDomainData(td);
  let tdok = true;
  for (let i = 0; i < 1024; i++) {
    const v = Math.min(255, Math.max(0, Math.floor(128 * (1 + pcm[i]))));
    if (td[i] !== v) tdok = false;
  }
  res.time_domain_ok = tdok;
  const tf = new Float32Array(2048);
  a.getFloatTimeDomainData(tf);
  res.time_domain_f32_ok = tf.every((v, i) => v === pcm[i]);
  let threw = '';
  try { a.getByteFrequencyData(new Float32Array(4)); } catch (e) { threw = e.name; }
  res.type_error = threw;
  try { a.fftSize = 1000; } catch (e) { res.range_error = e.name; }
  res.connect_ok = a.connect('x') === 'x';
  // fftSize on a live analyser: the smoothing state restarts, the sample history stays (the platform keeps its latest
  // 32768 samples whatever fftSize is): the last 512 of the 2048 samples pushed above are still there
  a.fftSize = 512;
  res.replan_bins = a.frequencyBinCount;
  const kept = new Float32Array(512);
  a.getFloatTimeDomainData(kept);
  res.history_kept = kept.every((v, i) => v === pcm[2048 - 512 + i]);
  const small = new Uint8Array(256);
  a.push(pcm.subarray(0, 512));
  a.getByteFrequencyData(small);
  res.replan_polls = small.some((v) => v > 0);
  // the 256-row history texture on the device: rows land at historyOffset, which wraps; pinned keeps row 0
  a.startHistory(3);
  const rows = [];
  for (let i = 0; i < 4; i++) {
    a.push(pcm.subarray(512 * (i + 1), 512 * (i + 2)));
    a.getByteFrequencyData(small);
    rows.push(Uint8Array.from(small));
  }
  res.ring_offset = a.historyOffset;                       // 4 polls into 3 rows -> 1
  const img = a.readHistory();
  const same = (x, y) => x.every((v, k) => v === y[k]);
  res.ring_ok = img.length === 3 * 256 && same(img.subarray(0, 256), rows[1]) && same(img.subarray(256, 512), rows[2]) &&
    same(img.subarray(512, 768), rows[3]);
  a.pinHistory(true);
  a.push(pcm.subarray(0, 512));
  a.getByteFrequencyData(small);
  res.ring_pinned = a.historyOffset === 0 && same(a.readHistory().subarray(0, 256), small);
  a.close();
  let closedThrew = false;
  try { a.getByteFrequencyData(small); } catch (e) { closedThrew = true; }
  res.closed_throws = closedThrew;
  // options beyond the platform object: rows of log-frequency columns, and a stereo source mixed down on push
  // (sonogram-vertex.shader:51-52; the `mix` node of UI/player.js:25)
  const lf = new S.Analyser({ fftSize: 1024, logfreqCols: 64, downmix: 2 });
  lf.smoothingTimeConstant = 0;
  const mono = new S.Analyser({ fftSize: 1024, logfreqCols: 64 });
  mono.smoothingTimeConstant = 0;
  res.row_length = lf.rowLength;
  const stereo = new Float32Array(2048), mix = new Float32Array(1024);
  for (let i = 0; i < 1024; i++) {
    stereo[2 * i] = pcm[i];
    stereo[2 * i + 1] = pcm[2048 + i];
    mix[i] = Math.fround(Math.fround(pcm[i] + pcm[2048 + i]) * 0.5);
  }
  lf.push(stereo);
  mono.push(mix);
  const c1 = new Uint8Array(64), c2 = new Uint8Array(64);
  lf.getByteFrequencyData(c1);
  mono.getByteFrequencyData(c2);
  res.downmix_cols_equal = same(c1, c2) && c1.some((v) => v > 0);
  // the columns sit where the reference's default display samples the row: the vertices of the 3-D sonogram's mesh,
  // u = x / (width - 1) (3D/visualizer.js:213-214, sonogram-vertex.shader:51-52) -- checked against the byte bins of
  // a plain analyser fed the same samples, blended here in double precision
  const plain = new S.Analyser({ fftSize: 1024 });
  plain.smoothingTimeConstant = 0;
  plain.push(mix);
  const binBytes = new Uint8Array(512);
  plain.getByteFrequencyData(binBytes);
  const blendAt = (u) => {
    const pp = Math.pow(256, u - 1) * 512 - 0.5, fl = Math.floor(pp), fr = pp - fl;
    const i0 = Math.min(511, Math.max(0, fl)), i1 = Math.min(511, Math.max(0, fl + 1));
    return binBytes[i0] * (1 - fr) + binBytes[i1] * fr;
  };
  let worstVertex = 0, worstCentre = 0;
  for (let c = 0; c < 64; c++) {
    worstVertex = Math.max(worstVertex, Math.abs(c1[c] - blendAt(c / 63)));
    worstCentre = Math.max(worstCentre, Math.abs(c1[c] - blendAt((c + 0.5) / 64)));
  }
  res.vertex_cols_worst = worstVertex;          // <= 0.5 (rounding to the nearest byte)
  res.centre_cols_worst = worstCentre;          // a different set of samples
  const centred = new S.Analyser({ fftSize: 1024, logfreqCols: 64, logfreqAlign: 'centre' });
  centred.smoothingTimeConstant = 0;
  centred.push(mix);
  const c3 = new Uint8Array(64);
  centred.getByteFrequencyData(c3);
  let worstC3 = 0;
  for (let c = 0; c < 64; c++) worstC3 = Math.max(worstC3, Math.abs(c3[c] - blendAt((c + 0.5) / 64)));
  res.centre_option_worst = worstC3;
  plain.close();
  centred.close();
  // one animation frame as one call: pushAndGetByteFrequencyData(samples, bytes) must hand out exactly what push() followed by
  // getByteFrequencyData() does on a twin analyser -- for blocks the native side runs as one launch (even, up to one window)
  // and for those it splits (odd lengths, longer than the window) -- and leave the same samples behind
  {
    const one = new S.Analyser({ fftSize: 1024 }), two = new S.Analyser({ fftSize: 1024 });
    one.smoothingTimeConstant = 0.5; two.smoothingTimeConstant = 0.5;
    const b1 = new Uint8Array(512), b2 = new Uint8Array(512), f1 = new Float32Array(512), f2 = new Float32Array(512);
    let same = true, pos = 0, i = 0;
    for (const blk of [736, 128, 735, 1024, 2, 1030, 512]) {
      const block = pcm.subarray(pos, pos + blk);
      pos += blk;
      if (i++ % 2 === 0) {
        one.pushAndGetByteFrequencyData(block, b1);
        two.push(block); two.getByteFrequencyData(b2);
        same = same && b1.every((v, k) => v === b2[k]);
      } else {
        one.pushAndGetFloatFrequencyData(block, f1);
        two.push(block); two.getFloatFrequencyData(f2);
        same = same && f1.every((v, k) => v === f2[k] || (v !== v && f2[k] !== f2[k]));
      }
    }
    const t1 = new Float32Array(1024), t2 = new Float32Array(1024);
    one.getFloatTimeDomainData(t1); two.getFloatTimeDomainData(t2);
    res.fused_frame_equals_two_calls = same && t1.every((v, k) => v === t2[k]) && t1.every((v, k) => v === pcm[pos - 1024 + k]);
    let threw = '';
    try { one.pushAndGetByteFrequencyData(block0(), new Float32Array(4)); } catch (e) { threw = e.name; }
    function block0() { return pcm.subarray(0, 64); }
    res.fused_type_error = threw;
    one.close(); two.close();
  }
  // a push that is not a whole number of rows is refused, not shifted silently
  try { lf.push(new Float32Array(2049)); res.ragged_push = 'accepted'; } catch (e) { res.ragged_push = e.name; }
  lf.startHistory(2);
  lf.getByteFrequencyData(c1);
  res.lf_ring_len = lf.readHistory().length;
  lf.close();
  mono.close();
  console.log(JSON.stringify(res));
}
