'use strict';
/*
 * js_baseline.js -- times the plain-JS STFT (oracle/stft_oracle.js, Float32Array) on the host
 * cores with worker_threads: the "reference JS path" CPU baseline of BASELINE.md section 4 (B1).
 * Prints one JSON line.  Usage: node js_baseline.js <threads> <seconds> [fft hop samples]
 */
const { Worker, isMainThread, parentPort, workerData } = require('worker_threads');
const O = require('./stft_oracle.js');

function synth(n, seed) {  // uniform(-1,1), xorshift32
  const x = new Float32Array(n);
  let s = seed >>> 0 || 1;
  for (let i = 0; i < n; i++) { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; x[i] = s / 2147483648 - 1; }
  return x;
}

if (isMainThread) {
  const threads = parseInt(process.argv[2] || '1', 10);
  const seconds = parseFloat(process.argv[3] || '10');
  const fft = parseInt(process.argv[4] || '1024', 10), hop = parseInt(process.argv[5] || '256', 10);
  const samples = parseInt(process.argv[6] || '28800000', 10);
  const frames = Math.floor((samples - fft) / hop) + 1;
  const per = Math.ceil(frames / threads);
  let done = 0, total = 0;
  const t0 = process.hrtime.bigint();
  for (let i = 0; i < threads; i++) {
    const f0 = i * per, nf = Math.min(per, frames - f0);
    if (nf <= 0) { done++; continue; }
    const w = new Worker(__filename, { workerData: { fft, hop, nf, seconds, seed: 0x5eed0002 + i } });
    w.on('message', (m) => { total += m; });
    w.on('exit', () => {
      if (++done === threads) {
        const dt = Number(process.hrtime.bigint() - t0) / 1e9;
        console.log(JSON.stringify({ frames_per_s: total / dt, threads, frames: total, seconds: dt, fft, hop }));
      }
    });
  }
} else {
  const { fft, hop, nf, seconds, seed } = workerData;
  const seg = synth((nf - 1) * hop + fft, seed);   // this worker's frame range of the stream (with halo)
  const out = new Float32Array(nf * fft / 2);
  const stop = Date.now() + seconds * 1000;
  let n = 0;
  do { O.stftPower(seg, fft, hop, { arrayType: Float32Array, out }); n += nf; } while (Date.now() < stop);
  parentPort.postMessage(n);
}
