// gen_libtwin.mjs - golden vectors for the LIBRARY twin of the limiter, web/lib/dsp/limiter.js:155-362 (applyLookaheadLimiter),
// and for normalizeToLUFS with its limiter branch (web/lib/dsp/normalizer.js:38-94), produced by running the reference's own
// modules under Node (they load unmodified: no newer syntax).  The worker twin's goldens are in golden.json.
//   node tests/golden/gen_libtwin.mjs          (needs /root/reference; nothing of it is copied into the repo)
import fs from 'fs';
import os from 'os';
import path from 'path';
import { fileURLToPath, pathToFileURL } from 'url';
import { synthTrack } from './synth.mjs';

const here = path.dirname(fileURLToPath(import.meta.url));
const REF = process.env.WAM_REF || '/root/reference';
const OUT = path.join(here, 'data');
const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'wam-libtwin-'));
fs.writeFileSync(path.join(dir, 'package.json'), '{"type":"module"}');
for (const f of ['constants.js', 'utils.js', 'lufs.js', 'true-peak.js', 'limiter.js', 'normalizer.js']) fs.copyFileSync(path.join(REF, 'web/lib/dsp', f), path.join(dir, f));

class AB {
  constructor({ numberOfChannels, length, sampleRate }) { this.numberOfChannels = numberOfChannels; this.length = length; this.sampleRate = sampleRate; this.duration = length / sampleRate; this._c = []; for (let i = 0; i < numberOfChannels; i++) this._c.push(new Float32Array(length)); }
  getChannelData(c) { return this._c[c]; }
}
globalThis.AudioBuffer = AB;
function toAB(chans, sr) { const ab = new AB({ numberOfChannels: chans.length, length: chans[0].length, sampleRate: sr }); chans.forEach((c, i) => ab._c[i].set(c)); return ab; }
function chansOf(ab) { const r = []; for (let c = 0; c < ab.numberOfChannels; c++) r.push(ab.getChannelData(c)); return r; }
function saveF32(name, a) { fs.writeFileSync(path.join(OUT, name + '.f32'), Buffer.from(a.buffer, a.byteOffset, a.byteLength)); return name + '.f32'; }
const saveChans = (name, chans) => chans.map((c, i) => saveF32(`${name}.ch${i}`, c));

async function main() {
  const log = console.log; console.log = () => {}; console.warn = () => {};
  const lim = await import(pathToFileURL(path.join(dir, 'limiter.js')).href);
  const nrm = await import(pathToFileURL(path.join(dir, 'normalizer.js')).href);
  const cases = {};
  const sr = 48000, n = 24000;
  const ceil = Math.pow(10, -1 / 20);
  // hot input: the limiter's gain stage leaves inter-sample overshoots for the soft-knee pass, where the twins differ
  const opts = { amp: 2.4, clickAmp: 0.9 };
  const hot = synthTrack(31, 2, n, sr, opts);
  cases.lib_limiter_hot = { idx: 31, nch: 2, n, sr, opts, ceiling_lin: ceil, out: saveChans('libtwin.hot', chansOf(lim.applyLookaheadLimiter(toAB(hot, sr), ceil))) };
  cases.lib_limiter_hot_np = { idx: 31, nch: 2, n, sr, opts, ceiling_lin: ceil, preserve: false,
    out: saveChans('libtwin.hot_np', chansOf(lim.applyLookaheadLimiter(toAB(hot, sr), ceil, 3, 100, 3, false))) };
  const mono = synthTrack(32, 1, n, sr, opts);
  cases.lib_limiter_mono = { idx: 32, nch: 1, n, sr, opts, ceiling_lin: ceil, out: saveChans('libtwin.mono', chansOf(lim.applyLookaheadLimiter(toAB(mono, sr), ceil))) };
  // normalizeToLUFS with the limiter branch taken (target -9: projected peak above -1 dBTP) and not taken (-30)
  const o2 = { clickAmp: 0.5 };
  const x = synthTrack(33, 2, n, sr, o2);
  cases.lib_normalize_limited = { idx: 33, nch: 2, n, sr, opts: o2, target: -9, ceiling_db: -1, out: saveChans('libtwin.norm9', chansOf(nrm.normalizeToLUFS(toAB(x, sr), -9, -1))) };
  cases.lib_normalize_plain = { idx: 33, nch: 2, n, sr, opts: o2, target: -30, ceiling_db: -1, out: saveChans('libtwin.norm30', chansOf(nrm.normalizeToLUFS(toAB(x, sr), -30, -1))) };
  console.log = log;
  fs.writeFileSync(path.join(OUT, 'libtwin.json'), JSON.stringify({ generator: 'tests/golden/gen_libtwin.mjs', node: process.version, cases }, null, 1));
  try { fs.rmdirSync(dir, { recursive: true }); } catch (e) { /* ignore */ }
  console.log('wrote', Object.keys(cases).length, 'cases');
}
main().catch((e) => { console.error(e); process.exit(1); });
