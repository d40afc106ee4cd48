'use strict';
/*
 * draw_loop.js -- the reference's per-animation-frame loop (UI/spectrogram.js:153-161 -> 3D/visualizer.js:346-417) over a
 * recorded signal, in the reference's own language, on the MI355X engine:
 *
 *   every frame:  the audio that arrived since the last frame reaches the analyser and
 *                 analyser.getByteFrequencyData(freqByteData)            (3D/visualizer.js:358)
 *                 -> here ONE call, analyser.pushAndGetByteFrequencyData(block, freqByteData): one kernel launch
 *                 the row goes into the 256-row history texture at yoffset  (3D/visualizer.js:399-417)
 *                 -> kept on the device by the analyser (startHistory), rows of the 256 log-frequency columns the
 *                    vertex shader samples (sonogram-vertex.shader:51-52)
 *   at the end:   the history, oldest row first, as a PGM image -- the picture the reference draws.
 *
 *   usage: node examples/draw_loop.js <pcm.f32 (mono float32, 44.1 kHz)> <out.pgm> [frames]
 */
const fs = require('fs');
const path = require('path');
const { Analyser } = require(path.join(__dirname, '..', 'spectrogram_amd', 'js', 'analyser.js'));

const buf = fs.readFileSync(process.argv[2]);
const pcm = new Float32Array(buf.buffer, buf.byteOffset, buf.length >> 2);
const SAMPLES_PER_FRAME = 736;                       // 44.1 kHz / 60 fps, rounded to an even block (one launch per frame)
const HEIGHT = 256, COLS = 256;                      // TEXTURE_HEIGHT, sonogram3DWidth (3D/visualizer.js:54,60)
const frames = Math.min(parseInt(process.argv[4] || '0', 10) || 1e9, Math.floor(pcm.length / SAMPLES_PER_FRAME));

const analyser = new Analyser({ logfreqCols: COLS });  // columns on the 3-D sonogram's mesh (the default)
analyser.fftSize = 2048;                             // UI/player.js:10
analyser.smoothingTimeConstant = 0;                  // UI/player.js:11
analyser.startHistory(HEIGHT);
const freqByteData = new Uint8Array(analyser.rowLength);   // 3D/visualizer.js:301, here 256 columns

const t0 = process.hrtime.bigint();
for (let f = 0; f < frames; f++) {
  analyser.pushAndGetByteFrequencyData(pcm.subarray(f * SAMPLES_PER_FRAME, (f + 1) * SAMPLES_PER_FRAME), freqByteData);
}
const ms = Number(process.hrtime.bigint() - t0) / 1e6;
const image = analyser.readHistory();                // oldest row first
fs.writeFileSync(process.argv[3], Buffer.concat([Buffer.from(`P5\n${COLS} ${HEIGHT}\n255\n`), Buffer.from(image.buffer, image.byteOffset, image.length)]));
console.log(JSON.stringify({ frames, ms_per_frame: ms / frames, last_row_sum: freqByteData.reduce((a, b) => a + b, 0), history_offset: analyser.historyOffset }));
analyser.close();
