#!/usr/bin/env node
/*
 * make_reference_fixtures.js -- the one piece of reference arithmetic that can run here, run as-is.
 *
 * Build-container only (it reads /root/reference, which does not exist on the GPU box; the tests read only the
 * JSON it writes).  The reference has no FFT code to pin (SURVEY.md section 0), but its log-scale legend math
 * is plain JavaScript: util/util.js:32-43 (setLogScale / lin2log / log2lin), configured by
 * UI/spectrogram.js:77 as setLogScale(20, 20, 20000, 20000).  This script loads THAT FILE's text at run time
 * (nothing of it is copied into this repository), evaluates it under node, and records inputs -> outputs in
 * tests/golden/ref_util_logscale.json.  spectrogram_amd's axis helpers (include/spectro.h sg_axis_*) are tested
 * against these values.
 *
 *   node oracle/make_reference_fixtures.js [/root/reference]
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');

const refRoot = process.argv[2] || '/root/reference';
const file = path.join(refRoot, 'src', 'javascripts', 'util', 'util.js');
const text = fs.readFileSync(file, 'utf8');
// the file is an ES module with a single default export; node 12 loads .js as CommonJS, so only the export
// keyword is rewritten in memory -- the object literal that follows is evaluated exactly as written
const sandbox = { module: { exports: null }, Math, console };
vm.runInNewContext(text.replace(/^\s*export\s+default\s*/m, 'module.exports = '), sandbox, { filename: file });
const Util = sandbox.module.exports;
if (!Util || typeof Util.setLogScale !== 'function') throw new Error('unexpected shape of ' + file);

Util.setLogScale(20, 20, 20000, 20000); // UI/spectrogram.js:77
const xs = [];
for (let i = 0; i <= 64; i++) xs.push(20 + (20000 - 20) * i / 64);
xs.push(20.5, 440, 1000, 12345.678, 19999.5);
const ys = [20, 27.5, 55, 110, 220, 440, 880, 1000, 1760, 3520, 7040, 14080, 20000, 31.4159, 9999.99];
// what UI/spectrogram.js:191-221 computes around these calls for a canvas of a given height (the jQuery height
// lookup replaced by a number; the arithmetic is the reference's: padding 30, freqStart 20, freqEnd 20000)
const legend = [];
for (const height of [480, 1080]) {
  for (const y of [30, 31, 100, height / 2, height - 31, height - 30]) {
    const percentFromBottom = 1 - (y - 30) / (height - 30);
    legend.push({ height, y, freq: Util.lin2log(20 + (20000 - 20) * percentFromBottom) });
  }
}
const out = {
  source: 'src/javascripts/util/util.js:32-43 evaluated under node ' + process.version + ' (reference text loaded at run time, not copied)',
  setLogScale: [20, 20, 20000, 20000],
  a: Util.a,
  b: Util.b,
  x: xs,
  lin2log: xs.map((x) => Util.lin2log(x)),
  y: ys,
  log2lin: ys.map((y) => Util.log2lin(y)),
  legend,
};
const dst = path.join(__dirname, '..', 'tests', 'golden', 'ref_util_logscale.json');
fs.writeFileSync(dst, JSON.stringify(out, null, 1) + '\n');
console.log('wrote', dst, 'a =', Util.a, 'b =', Util.b);

/*
 * Second sliver (round 4): where the reference's default display samples the analyser row.  The 3-D sonogram looks the
 * row up in its VERTEX shader at x = pow(256, gTexCoord0.x - 1) (sonogram-vertex.shader:51-52, GLSL: cannot run here),
 * and gTexCoord0 comes from a mesh that plain JavaScript builds: 3D/visualizer.js:203-218, inside initGL() between the
 * WebGL calls.  The loop's text is cut out of THAT FILE at run time (from "const numVertices" to the line before
 * "const vbo3DTexCoordOffset"; nothing is copied into this repository) and evaluated with the three sizes its
 * constructor assigns (:54-56, read from the same text).  Recorded: the texture coordinate u of every mesh column --
 * what sg_config.logfreq_align = SG_LOGFREQ_VERTEX must reproduce (tests/test_axis.py).
 */
const vfile = path.join(refRoot, 'src', 'javascripts', '3D', 'visualizer.js');
const vtext = fs.readFileSync(vfile, 'utf8');
const num = (name) => {
  const m = vtext.match(new RegExp('this\\.' + name + '\\s*=\\s*([0-9.]+)\\s*;'));
  if (!m) throw new Error('no assignment of ' + name + ' in ' + vfile);
  return Number(m[1]);
};
const a0 = vtext.indexOf('const numVertices = sonogram3DWidth * sonogram3DHeight;');
const a1 = vtext.indexOf('const vbo3DTexCoordOffset');
if (a0 < 0 || a1 < a0) throw new Error('mesh loop not found in ' + vfile);
const meshBox = {
  sonogram3DWidth: num('sonogram3DWidth'), sonogram3DHeight: num('sonogram3DHeight'),
  sonogram3DGeometrySize: num('sonogram3DGeometrySize'), Float32Array, vertices: null, texCoords: null,
};
vm.runInNewContext('(function () {' + vtext.slice(a0, a1) + '; return 0; })()', meshBox, { filename: vfile });
const W = meshBox.sonogram3DWidth, H = meshBox.sonogram3DHeight;
const u = [], vrow = [];
for (let x = 0; x < W; x++) u.push(meshBox.texCoords[2 * x]);            // row z = 0: texCoord.x of every column
for (let z = 0; z < H; z++) vrow.push(meshBox.texCoords[2 * (W * z) + 1]);
let same = true;                                                           // every row of the mesh carries the same u
for (let z = 1; z < H; z++) for (let x = 0; x < W; x++) same = same && meshBox.texCoords[2 * (W * z + x)] === u[x];
const mesh = {
  source: 'src/javascripts/3D/visualizer.js:203-218 (mesh loop) with :54-56 (sizes), evaluated under node ' + process.version +
          ' (reference text loaded at run time, not copied)',
  sonogram3DWidth: W, sonogram3DHeight: H, sonogram3DGeometrySize: meshBox.sonogram3DGeometrySize,
  texcoord_x: u, texcoord_y: vrow, texcoord_x_same_in_every_row: same,
  vertex_x_first_last: [meshBox.vertices[0], meshBox.vertices[3 * (W - 1)]],
};
const dst2 = path.join(__dirname, '..', 'tests', 'golden', 'ref_mesh_texcoords.json');
fs.writeFileSync(dst2, JSON.stringify(mesh) + '\n');
console.log('wrote', dst2, 'u[0] =', u[0], 'u[1] =', u[1], 'u[last] =', u[W - 1]);
