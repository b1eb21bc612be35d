// Golden-vector generator: runs the REFERENCE's own JavaScript under Node and
// records its outputs on seeded synthetic inputs (tests/golden/synth.mjs).
//
//   node tests/golden/gen_golden.mjs [/root/reference] [tests/golden/data]
//
// Runs only in the build container (needs /root/reference). Nothing of the
// reference is written into the repo: its modules are copied to a scratch
// directory under os.tmpdir(), where three mechanical edits make them loadable
// on Node 12 (SURVEY.md 8c):
//   1. `a ?? b` and `a?.b` (Node >= 14 syntax) are lowered to helper calls;
//   2. worker-local functions of dsp-worker.js get an `export {...}` line and
//      `self` is shimmed so the real `self.onmessage` handler can be driven;
//   3. the un-vendored npm dependency `fft.js@4.0.4` is absent, so a
//      double-precision substitute with the same five methods is placed at
//      node_modules/fft.js. It only EXECUTES for de-harsh (a17); every golden
//      that depends on it is tagged `"dft": "substitute"` = parity unpinned.
// The scratch copy is deleted at exit.

import fs from 'fs';
import os from 'os';
import path from 'path';
import { pathToFileURL } from 'url';
import { synthTrack } from './synth.mjs';

const REF = process.argv[2] || '/root/reference';
const OUT = process.argv[3] || path.join(path.dirname(new URL(import.meta.url).pathname), 'data');

// ---------------------------------------------------------------- scratch copy
const HELPERS =
  'const __oc=(o,p)=>(o===undefined||o===null?undefined:o[p]);' +
  'const __nc=(...a)=>{for(const v of a){if(v!==undefined&&v!==null)return v;}return a[a.length-1];};\n';

function lowerLine(line) {
  if (/^\s*(\/\/|\*|\/\*)/.test(line)) return line;
  let s = line;
  const oc = /([A-Za-z_$][\w$]*(?:\.[\w$]+)*|__oc\((?:[^()]|\([^()]*\))*\))\?\.([A-Za-z_$][\w$]*)/;
  while (oc.test(s)) s = s.replace(oc, '__oc($1,"$2")');
  if (s.includes('??')) {
    // <prefix (= or :)> e1 ?? e2 ?? e3 <; or , or nothing>
    s = s.replace(/\s*\/\/.*$/, '');     // trailing line comment
    const m = /^(.*?(?:=|:)\s*)(.+?)\s*([;,]?)\s*$/.exec(s);
    if (!m) throw new Error('cannot lower: ' + line);
    const parts = m[2].split('??').map(x => x.trim());
    s = `${m[1]}__nc(${parts.join(', ')})${m[3]}`;
  }
  return s;
}

function lowerSource(src) {
  return HELPERS + src.split('\n').map(lowerLine).join('\n');
}

const SUBSTITUTE_FFT = `
// Substitute for npm fft.js@4.0.4 (absent offline): same five methods, radix-2
// double-precision DFT, unnormalised forward, 1/N inverse, interleaved complex.
class FFT {
  constructor(size) {
    this.size = size; this._csize = size << 1;
    let bits = 0; while ((1 << bits) < size) bits++;
    this._bits = bits;
    this._rev = new Uint32Array(size);
    for (let i = 0; i < size; i++) { let r = 0, x = i; for (let b = 0; b < bits; b++) { r = (r << 1) | (x & 1); x >>= 1; } this._rev[i] = r; }
    this._cos = new Float64Array(size / 2); this._sin = new Float64Array(size / 2);
    for (let i = 0; i < size / 2; i++) { const a = -2 * Math.PI * i / size; this._cos[i] = Math.cos(a); this._sin[i] = Math.sin(a); }
  }
  createComplexArray() { const r = new Array(this._csize); for (let i = 0; i < r.length; i++) r[i] = 0; return r; }
  _tr(out, data, inv) {
    const n = this.size;
    for (let i = 0; i < n; i++) { const j = this._rev[i]; out[2 * j] = data[2 * i]; out[2 * j + 1] = data[2 * i + 1]; }
    for (let size = 2; size <= n; size <<= 1) {
      const half = size >> 1, step = n / size;
      for (let i = 0; i < n; i += size) {
        for (let j = i, k = 0; j < i + half; j++, k += step) {
          const c = this._cos[k], s = inv ? -this._sin[k] : this._sin[k];
          const xr = out[2 * (j + half)], xi = out[2 * (j + half) + 1];
          const tr = xr * c - xi * s, ti = xr * s + xi * c;
          out[2 * (j + half)] = out[2 * j] - tr; out[2 * (j + half) + 1] = out[2 * j + 1] - ti;
          out[2 * j] += tr; out[2 * j + 1] += ti;
        }
      }
    }
    if (inv) for (let i = 0; i < 2 * n; i++) out[i] /= n;
  }
  transform(out, data) { this._tr(out, data, false); }
  inverseTransform(out, data) { this._tr(out, data, true); }
  realTransform(out, data) {
    const tmp = new Array(this._csize);
    for (let i = 0; i < this.size; i++) { tmp[2 * i] = data[i]; tmp[2 * i + 1] = 0; }
    this._tr(out, tmp, false);
  }
  completeSpectrum(spectrum) {
    const n = this.size;
    for (let i = 1; i < n / 2; i++) { spectrum[2 * (n - i)] = spectrum[2 * i]; spectrum[2 * (n - i) + 1] = -spectrum[2 * i + 1]; }
  }
}
export default FFT;
`;

const WORKER_EXPORTS = `
export { WorkerAudioBuffer, measureLUFSFromChannels, findTruePeakFromChannels, applySoftKneeCurve,
  applySoftKneeOversampled, applyLookaheadLimiterToChannels, normalizeChannelsToLUFS,
  applyStereoProcessingToChannels, applyParametricEQ, applyGlueCompressor };
`;

function prepareScratch() {
  const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'wam-ref-'));
  fs.writeFileSync(path.join(dir, 'package.json'), '{"type":"module"}');
  for (const sub of ['lib/dsp', 'workers', 'ui']) fs.mkdirSync(path.join(dir, sub), { recursive: true });
  for (const f of fs.readdirSync(path.join(REF, 'web/lib/dsp'))) {
    fs.writeFileSync(path.join(dir, 'lib/dsp', f), lowerSource(fs.readFileSync(path.join(REF, 'web/lib/dsp', f), 'utf8')));
  }
  fs.writeFileSync(path.join(dir, 'workers/dsp-worker.js'),
    lowerSource(fs.readFileSync(path.join(REF, 'web/workers/dsp-worker.js'), 'utf8')) + WORKER_EXPORTS);
  for (const f of ['encoder.js', 'spectrogram.js']) {
    fs.writeFileSync(path.join(dir, 'ui', f), lowerSource(fs.readFileSync(path.join(REF, 'web/ui', f), 'utf8')));
  }
  fs.mkdirSync(path.join(dir, 'node_modules/fft.js'), { recursive: true });
  fs.writeFileSync(path.join(dir, 'node_modules/fft.js/package.json'), '{"name":"fft.js","type":"module","main":"index.js"}');
  fs.writeFileSync(path.join(dir, 'node_modules/fft.js/index.js'), SUBSTITUTE_FFT);
  return dir;
}

// ---------------------------------------------------------------- fixtures out
const golden = { generator: 'tests/golden/gen_golden.mjs', node: process.version, cases: {} };
fs.mkdirSync(OUT, { recursive: true });

function num(v) { return Number.isFinite(v) ? v : (Number.isNaN(v) ? 'nan' : (v > 0 ? 'inf' : '-inf')); }
function saveF32(name, arr) {
  const a = arr instanceof Float32Array ? arr : Float32Array.from(arr);
  fs.writeFileSync(path.join(OUT, name + '.f32'), Buffer.from(a.buffer, a.byteOffset, a.byteLength));
  return name + '.f32';
}
function saveF64(name, arr) {
  const a = arr instanceof Float64Array ? arr : Float64Array.from(arr);
  fs.writeFileSync(path.join(OUT, name + '.f64'), Buffer.from(a.buffer, a.byteOffset, a.byteLength));
  return name + '.f64';
}
function saveU8(name, arr) {
  fs.writeFileSync(path.join(OUT, name + '.u8'), Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength));
  return name + '.u8';
}
function saveChans(name, chans) { return chans.map((c, i) => saveF32(`${name}.ch${i}`, c)); }
function stats(chans) {           // order-defined checksums (sequential double sums)
  return chans.map(c => {
    let s = 0, s2 = 0, m = 0, am = 0;
    for (let i = 0; i < c.length; i++) { const v = c[i]; s += v; s2 += v * v; const a = Math.abs(v); if (a > m) { m = a; am = i; } }
    return { sum: num(s), sumsq: num(s2), maxabs: num(m), argmax: am };
  });
}
function windows(chans, nwin, wlen) { // strided windows of a long signal
  const n = chans[0].length;
  const out = new Float32Array(chans.length * nwin * wlen);
  const starts = [];
  for (let w = 0; w < nwin; w++) starts.push(Math.floor((n - wlen) * w / (nwin - 1)));
  let p = 0;
  for (const c of chans) for (const s of starts) for (let i = 0; i < wlen; i++) out[p++] = c[s + i];
  return { starts, data: out };
}

class AB {                         // duck-typed AudioBuffer, same shape as dsp-worker.js:36-61
  constructor({ numberOfChannels, length, sampleRate }) {
    this.numberOfChannels = numberOfChannels; this.length = length; this.sampleRate = sampleRate;
    this.duration = length / sampleRate; this._channels = [];
    for (let i = 0; i < numberOfChannels; i++) this._channels.push(new Float32Array(length));
  }
  getChannelData(c) { return this._channels[c]; }
  copyToChannel(src, c, start = 0) { this._channels[c].set(src, start); }
  copyFromChannel(dest, c, start = 0) { dest.set(this._channels[c].subarray(start, start + dest.length)); }
}
function toAB(chans, sr) {
  const ab = new AB({ numberOfChannels: chans.length, length: chans[0].length, sampleRate: sr });
  chans.forEach((c, i) => ab.copyToChannel(c, i));
  return ab;
}
function chansOf(ab) { const r = []; for (let c = 0; c < ab.numberOfChannels; c++) r.push(ab.getChannelData(c)); return r; }
function cloneChans(ch) { return ch.map(c => c.slice()); }

// mirror of lufs.js gating, used ONLY to record block energies / gate masks
// next to the value measureLUFS itself returned (which is what is pinned)
function lufsDetail(lib, ab) {
  const sr = ab.sampleRate, nch = ab.numberOfChannels, n = ab.length;
  const hs = lib.calcHighShelfCoeffs(sr, lib.K_WEIGHTING.HIGH_SHELF_FREQ, lib.K_WEIGHTING.HIGH_SHELF_GAIN, lib.K_WEIGHTING.HIGH_SHELF_Q);
  const hp = lib.calcHighPassCoeffs(sr, lib.K_WEIGHTING.HIGH_PASS_FREQ, lib.K_WEIGHTING.HIGH_PASS_Q);
  const f = chansOf(ab).map(c => lib.applyBiquadFilter(lib.applyBiquadFilter(c, hs), hp));
  const bs = Math.floor(sr * 0.4), hop = Math.floor(sr * 0.4 * (1 - 0.75));
  const blocks = [];
  for (let st = 0; st + bs <= n; st += hop) {
    let ss = 0;
    for (let c = 0; c < nch; c++) for (let i = st; i < st + bs; i++) ss += f[c][i] * f[c][i];
    blocks.push(ss / (bs * nch));
  }
  const g1 = blocks.map(m => m > 1e-7);
  const k1 = blocks.filter(m => m > 1e-7);
  const mean1 = k1.length ? k1.reduce((a, b) => a + b, 0) / k1.length : 0;
  const mask = blocks.map((m, i) => (g1[i] && m > mean1 * 0.1) ? 2 : (g1[i] ? 1 : 0));
  return { blocks, mask, coeffs: { hs, hp }, filtered: f };
}

async function main() {
  const dir = prepareScratch();
  const cleanup = () => { if (process.env.WAM_KEEP) { quiet.log('scratch kept at', dir); return; } try { fs.rmdirSync(dir, { recursive: true }); } catch (e) { /* ignore */ } };
  process.on('exit', cleanup);
  const quiet = { log: console.log, warn: console.warn, error: console.error };
  console.log = () => { }; console.warn = () => { };
  globalThis.AudioBuffer = AB;
  const posted = [];
  globalThis.self = { postMessage: (m) => posted.push(m) };
  const imp = (p) => import(pathToFileURL(path.join(dir, p)).href);
  const lib = await imp('lib/dsp/index.js');
  const worker = await imp('workers/dsp-worker.js');
  const enc = await imp('ui/encoder.js');
  const spec = await imp('ui/spectrogram.js');
  const onmessage = globalThis.self.onmessage;
  globalThis.AudioBuffer = AB;     // dsp-worker.js replaced it with its own class; same shape
  const say = (...a) => quiet.log(...a);

  async function runChain(chans, sr, settings, mode = 'export') {
    posted.length = 0;
    await onmessage({ data: { type: 'RENDER_FULL_CHAIN', id: 1, data: { channels: cloneChans(chans), sampleRate: sr, settings, mode } } });
    const fin = posted.filter(m => m.type !== 'PROGRESS').pop();
    const progress = posted.filter(m => m.type === 'PROGRESS').map(m => [m.progress, m.status]);
    return { fin, progress };
  }

  // ------------------------------------------------------------ A: metering
  const meterCases = [
    { name: 'meter_10s_48k', idx: 0, nch: 2, sr: 48000, sec: 10, opts: {} , blocks: true },
    { name: 'meter_gated_48k', idx: 1, nch: 2, sr: 48000, sec: 10, blocks: true,
      opts: { quietFrom: 96000, quietTo: 192000, quietGain: 0.0, quiet2From: 240000, quiet2To: 384000, quiet2Gain: 0.02, clickAmp: 0.3 } },
    { name: 'meter_mono_44k', idx: 2, nch: 1, sr: 44100, sec: 5, opts: { amp: 0.5 }, blocks: true },
    { name: 'meter_96k', idx: 3, nch: 2, sr: 96000, sec: 3, opts: { dc: 0.002 }, blocks: true },
    { name: 'meter_short', idx: 4, nch: 2, sr: 48000, n: 19199, opts: {} },
    { name: 'meter_exact_block', idx: 4, nch: 2, sr: 48000, n: 19200, opts: {}, blocks: true },
    { name: 'meter_silence', idx: 5, nch: 2, sr: 48000, sec: 1, opts: { amp: 0.0 } },
    { name: 'meter_180s_48k', idx: 0, nch: 2, sr: 48000, sec: 180, opts: { clickAmp: 0.25 } }
  ];
  for (const c of meterCases) {
    const n = c.n || Math.round(c.sec * c.sr);
    const ch = synthTrack(c.idx, c.nch, n, c.sr, c.opts);
    const ab = toAB(ch, c.sr);
    const t0 = Date.now();
    const lufs = lib.measureLUFS(ab);
    const t1 = Date.now();
    const tp = lib.findTruePeak(ab);
    const t2 = Date.now();
    const lufsW = worker.measureLUFSFromChannels(ch, c.sr);
    const tpW = worker.findTruePeakFromChannels(ch);
    const rec = { kind: 'meter', idx: c.idx, nch: c.nch, sr: c.sr, n, opts: c.opts, lufs: num(lufs), tp_db: num(tp),
      lufs_worker: num(lufsW), tp_db_worker: num(tpW), lufs_fallback_m20: num(lib.measureLUFS(ab, -20)),
      input_stats: stats(ch), ms_lufs: t1 - t0, ms_tp: t2 - t1 };
    if (ab.duration >= 0.4) {
      const d = lufsDetail(lib, ab);
      rec.nblocks = d.blocks.length;
      rec.ngate_abs = d.mask.filter(m => m >= 1).length;
      rec.ngate_rel = d.mask.filter(m => m === 2).length;
      rec.kcoeffs = d.coeffs;
      if (c.blocks) { rec.block_ms = saveF64(c.name + '.block_ms', d.blocks); rec.gate_mask = saveU8(c.name + '.gate_mask', Uint8Array.from(d.mask)); }
      rec.kweighted_stats = stats(d.filtered);
    }
    rec.dc = ch.map(x => num(lib.detectDCOffset(x).linear));
    golden.cases[c.name] = rec;
    say(c.name, 'lufs', lufs, 'tp', tp, `(${t1 - t0} ms, ${t2 - t1} ms)`);
  }
  // short-term / momentary on the 10 s track (a4)
  {
    const ch = synthTrack(0, 2, 480000, 48000, {});
    const ab = toAB(ch, 48000);
    golden.cases.meter_10s_48k.short_term = [0, 2.5, 8.6].map(p => [p, num(lib.measureShortTermLUFS(ab, p))]);
    golden.cases.meter_10s_48k.momentary = [0, 2.5, 9.7, 9.9].map(p => [p, num(lib.measureMomentaryLUFS(ab, p))]);
  }
  // true-peak special vectors (inputs stored: they use Math.sin)
  {
    const n = 64;
    const imp1 = new Float32Array(n); imp1[10] = 1;
    const alt = new Float32Array(n); for (let i = 0; i < n; i++) alt[i] = (i & 1) ? -1 : 1;
    const q = new Float32Array(n); for (let i = 0; i < n; i++) q[i] = Math.sin(2 * Math.PI * 0.25 * i + Math.PI / 4);
    const edge = new Float32Array(n); edge[0] = 0.9; edge[1] = -0.8; edge[n - 1] = 0.95; edge[20] = 0.1;
    const all = { impulse: imp1, alternating: alt, fs4_45deg: q, edges: edge };
    const rec = { kind: 'tp_special', vectors: {} };
    for (const k of Object.keys(all)) {
      rec.vectors[k] = { input: saveF32('tp_' + k, all[k]), tp_db: num(lib.findTruePeak(toAB([all[k]], 48000))),
        tp_lin_channel: num(lib.findChannelTruePeak(all[k])) };
    }
    rec.sample4 = [[0.1, 0.9, -0.7, 0.3], [1, 1, 1, 1], [0, 1, 0, -1]].map(v => [v, lib.calculateTruePeakSample(v)]);
    golden.cases.tp_special = rec;
  }

  // ------------------------------------------------------------ B: stages, 1 s hot stereo clip
  {
    const sr = 48000, n = 24000, idx = 3;
    const opts = { amp: 2.6, dc: 0.004, clickAmp: 0.9 };
    const x = synthTrack(idx, 2, n, sr, opts);
    const rec = { kind: 'stages', idx, nch: 2, sr, n, opts, input_stats: stats(x), stages: {} };
    const S = rec.stages;
    // a7 DC
    const dcin = cloneChans(x);
    S.dc = { detect: dcin.map(c => num(lib.detectDCOffset(c).linear)),
      removed: dcin.map(c => num(lib.removeDCOffset(c))), out: saveChans('stages.dc', dcin) };
    // a8 input gain (worker-inline): reproduce through the real handler with everything else off
    const off = { normalizeLoudness: false, truePeakLimit: false, cleanLowEnd: false, glueCompression: false, deharsh: false,
      stereoWidth: 100, centerBass: false, cutMud: false, addAir: false, tapeWarmth: false, addPunch: false, inputGain: 0 };
    // a9 final filters
    const ff = lib.applyFinalFilters(toAB(dcin, sr), { highpass: true, lowpass: true });
    S.final_filters = { in: 'dc', out: saveChans('stages.ff', chansOf(ff)) };
    const ffNoHp = lib.applyFinalFilters(toAB(dcin, sr), { highpass: false, lowpass: true });
    S.final_filters_lp_only = { in: 'dc', out: saveChans('stages.ff_lp', chansOf(ffNoHp)) };
    // a11 glue
    const glue = worker.applyGlueCompressor(toAB(chansOf(ff), sr), { threshold: -18, ratio: 3, attack: 0.02, release: 0.25, knee: 10 });
    S.glue = { in: 'ff', out: saveChans('stages.glue', chansOf(glue)) };
    // a12 stereo
    const st1 = worker.applyStereoProcessingToChannels(chansOf(glue), sr, null, { width: 1.0, bassMono: true, bassFreq: 200 });
    S.stereo_bassmono = { in: 'glue', width: 1.0, bassMono: true, out: saveChans('stages.stereo_bm', st1) };
    const st2 = worker.applyStereoProcessingToChannels(chansOf(glue), sr, null, { width: 1.3, bassMono: false, bassFreq: 200 });
    S.stereo_wide = { in: 'glue', width: 1.3, bassMono: false, out: saveChans('stages.stereo_wide', st2) };
    // a13 normalize (skipLimiter)
    const stAB = toAB(st1, sr);
    const pre_lufs = lib.measureLUFS(stAB, -14), pre_tp = lib.findTruePeak(stAB);
    const norm = lib.normalizeToLUFS(stAB, -14, 0, { skipLimiter: true });
    S.normalize = { in: 'stereo_bm', target: -14, lufs_in: num(pre_lufs), tp_in: num(pre_tp), out: saveChans('stages.norm', chansOf(norm)) };
    // hot version so the clipper and limiter are well engaged
    const norm9 = lib.normalizeToLUFS(stAB, -9, 0, { skipLimiter: true });
    S.normalize_m9 = { in: 'stereo_bm', target: -9, out: saveChans('stages.norm9', chansOf(norm9)) };
    // a14 soft clip
    const sc = lib.applyMasteringSoftClip(norm9, { ceiling: -1, lookaheadMs: 0.5, releaseMs: 10, drive: 1.5 });
    S.softclip = { in: 'norm9', ceiling: -1, out: saveChans('stages.softclip', chansOf(sc)) };
    // a15/a16 worker limiter
    const ceil = Math.pow(10, -1 / 20);
    const lim = worker.applyLookaheadLimiterToChannels(chansOf(sc), sr, 1, ceil, 3, 100, 3, true);
    S.limiter = { in: 'softclip', ceiling_lin: ceil, out: saveChans('stages.limiter', lim) };
    // limiter straight on the hot signal (dense events, i < L peaks)
    const lim2 = worker.applyLookaheadLimiterToChannels(chansOf(norm9), sr, 1, ceil, 3, 100, 3, true);
    S.limiter_hot = { in: 'norm9', ceiling_lin: ceil, out: saveChans('stages.limiter_hot', lim2) };
    const lim3 = worker.applyLookaheadLimiterToChannels(chansOf(norm9), sr, 1, ceil, 3, 100, 3, false);
    S.limiter_hot_nopreserve = { in: 'norm9', ceiling_lin: ceil, out: saveChans('stages.limiter_hot_np', lim3) };
    S.softknee_os = { in: 'norm9', out: saveChans('stages.softknee', chansOf(norm9).map(c => worker.applySoftKneeOversampled(c, ceil, 3))) };
    // final meter
    S.final = { lufs: num(lib.measureLUFS(toAB(lim, sr))), tp_db: num(lib.findTruePeak(toAB(lim, sr))) };
    // a18 exciter, a19 warmth, a20 punch on the DC-removed clip
    const exc = lib.applyExciter(toAB(dcin, sr));
    S.exciter = { in: 'dc', out: saveChans('stages.exciter', chansOf(exc)) };
    const warm = lib.applyTapeWarmth(toAB(dcin, sr));
    S.warmth = { in: 'dc', out: saveChans('stages.warmth', chansOf(warm)) };
    const punch = lib.applyMultibandTransient(toAB(dcin, sr));
    S.punch = { in: 'dc', out: saveChans('stages.punch', chansOf(punch)) };
    // a17 de-harsh (substitute DFT -> unpinned)
    const dh = lib.processHybridDynamic(toAB(dcin, sr), 'mastering');
    S.deharsh = { in: 'dc', dft: 'substitute', out: saveChans('stages.deharsh', chansOf(dh)) };
    // a21 24-bit quantiser on the limiter output
    const wav = enc.encodeWAV(toAB(lim, sr), sr, 24, { ditherMode: 'none' });
    S.pcm24 = { in: 'limiter', header_hex: Buffer.from(wav.slice(0, 44)).toString('hex'), out: saveU8('stages.pcm24', wav.slice(44)) };
    const wav16 = enc.encodeWAV(toAB(lim, sr), sr, 16, { ditherMode: 'none' });
    S.pcm16_none = { in: 'limiter', header_hex: Buffer.from(wav16.slice(0, 44)).toString('hex'), out: saveU8('stages.pcm16', wav16.slice(44)) };
    golden.cases.stages_hot = rec;
    say('stages_hot final', S.final);
  }
  // a21 rounding edge cases
  {
    const n = 4096, M = 8388607;
    const a = new Float32Array(n);
    for (let i = 0; i < n; i++) a[i] = -1.0005 + 2.001 * i / (n - 1);
    a[0] = -1; a[1] = 1; a[2] = -1.5; a[3] = 1.5; a[4] = 0; a[5] = -0;
    for (let k = 0; k < 64; k++) { a[16 + 2 * k] = (k + 0.5) / M; a[17 + 2 * k] = -(k + 0.5) / M; }
    const b = new Float32Array(n); for (let i = 0; i < n; i++) b[i] = -a[n - 1 - i] * 0.37;
    const wav = enc.encodeWAV(toAB([a, b], 48000), 48000, 24, {});
    golden.cases.pcm24_edges = { kind: 'pcm24', n, in: [saveF32('pcm24_edges.ch0', a), saveF32('pcm24_edges.ch1', b)], out: saveU8('pcm24_edges.pcm', wav.slice(44)),
      header_hex: Buffer.from(wav.slice(0, 44)).toString('hex') };
  }
  // limiter with peaks inside the first L samples, quiet elsewhere
  {
    const sr = 48000, n = 4096;
    const x = synthTrack(6, 2, n, sr, { amp: 0.8 });
    x[0][5] = 1.4; x[1][60] = -1.3; x[0][143] = 1.2; x[0][144] = 1.25; x[1][300] = 1.5; x[0][n - 1] = 1.3; x[1][n - 2] = -1.2;
    const ceil = Math.pow(10, -1 / 20);
    const lim = worker.applyLookaheadLimiterToChannels(cloneChans(x), sr, 1, ceil, 3, 100, 3, true);
    const sc = lib.applyMasteringSoftClip(toAB(x, sr), { ceiling: -1, lookaheadMs: 0.5, releaseMs: 10, drive: 1.5 });
    golden.cases.limiter_edges = { kind: 'edges', sr, n, nch: 2, in: saveChans('lim_edges.in', x), limiter: saveChans('lim_edges.lim', lim),
      softclip: saveChans('lim_edges.sc', chansOf(sc)), ceiling_lin: ceil };
  }

  // ------------------------------------------------------------ C: full chain through the real worker handler
  const base = { normalizeLoudness: true, targetLufs: -14, truePeakLimit: true, truePeakCeiling: -1, cleanLowEnd: true,
    glueCompression: false, deharsh: true, stereoWidth: 100, centerBass: true, cutMud: false, addAir: true, tapeWarmth: true,
    autoLevel: false, addPunch: true, inputGain: 0, eqLow: 0, eqLowMid: 0, eqMid: 0, eqHighMid: 0, eqHigh: 0 };
  const chainCases = [
    { name: 'chain_defaults_1s', sec: 1, idx: 7, settings: base, dft: 'substitute', opts: { amp: 1.6, clickAmp: 0.6 } },
    { name: 'chain_nodeharsh_1s', sec: 1, idx: 7, settings: Object.assign({}, base, { deharsh: false }), opts: { amp: 1.6, clickAmp: 0.6 } },
    { name: 'chain_core_1s', sec: 1, idx: 8, opts: { amp: 1.6, clickAmp: 0.6, dc: 0.003 },
      settings: Object.assign({}, base, { deharsh: false, addAir: false, tapeWarmth: false, addPunch: false, glueCompression: true, stereoWidth: 120, inputGain: 1.5 }) },
    { name: 'chain_c2flat_1s', sec: 1, idx: 9, dft: 'substitute', opts: { amp: 1.6, clickAmp: 0.6 },
      settings: Object.assign({}, base, { glueCompression: true, tapeWarmth: false, addPunch: false }) },
    { name: 'chain_ceiling0_1s', sec: 1, idx: 10, opts: { amp: 1.6, clickAmp: 0.6 },
      settings: Object.assign({}, base, { deharsh: false, addAir: false, tapeWarmth: false, addPunch: false, truePeakCeiling: 0, targetLufs: -9 }) },
    { name: 'chain_preview_1s', sec: 1, idx: 11, mode: 'preview', opts: { amp: 1.0 },
      settings: Object.assign({}, base, { deharsh: false, inputGain: -3 }) },
    { name: 'chain_mono_1s', sec: 1, idx: 12, nch: 1, opts: { amp: 1.6, clickAmp: 0.6 },
      settings: Object.assign({}, base, { deharsh: false, glueCompression: true }) }
  ];
  for (const c of chainCases) {
    const sr = 48000, n = c.sec * sr, nch = c.nch || 2;
    const x = synthTrack(c.idx, nch, n, sr, c.opts);
    const { fin, progress } = await runChain(x, sr, c.settings, c.mode || 'export');
    if (!fin || !fin.success) throw new Error(c.name + ' failed: ' + (fin && fin.error));
    golden.cases[c.name] = { kind: 'chain', idx: c.idx, nch, sr, n, opts: c.opts, settings: c.settings, mode: c.mode || 'export',
      dft: c.dft || 'unused', lufs: num(fin.result.lufs), measured_lufs: num(fin.result.measuredLufs),
      out: saveChans(c.name, fin.result.channels), out_stats: stats(fin.result.channels), progress };
    say(c.name, 'lufs', fin.result.lufs);
  }
  // broken-EQ behaviour (SURVEY 0.1): record that the reference worker fails with non-flat EQ
  {
    const x = synthTrack(7, 2, 24000, 48000, {});
    const { fin } = await runChain(x, 48000, Object.assign({}, base, { deharsh: false, eqLow: 1 }));
    golden.cases.chain_eq_broken = { kind: 'note', success: !!(fin && fin.success), error: fin ? (fin.error || null) : 'no reply' };
  }
  // 180 s chain: checksums + strided windows (deharsh off = pinned; on = substitute DFT)
  for (const c of [{ name: 'chain_180s_nodeharsh', deharsh: false }, { name: 'chain_180s_defaults', deharsh: true }]) {
    const sr = 48000, n = 180 * sr;
    const opts = { amp: 1.6, clickAmp: 0.6, dc: 0.002 };
    const x = synthTrack(1, 2, n, sr, opts);
    const settings = Object.assign({}, base, { deharsh: c.deharsh, glueCompression: true });
    const t0 = Date.now();
    const { fin } = await runChain(x, sr, settings);
    const ms = Date.now() - t0;
    if (!fin || !fin.success) throw new Error(c.name + ' failed: ' + (fin && fin.error));
    const w = windows(fin.result.channels, 64, 256);
    golden.cases[c.name] = { kind: 'chain_long', idx: 1, nch: 2, sr, n, opts, settings, dft: c.deharsh ? 'substitute' : 'unused',
      lufs: num(fin.result.lufs), out_stats: stats(fin.result.channels), tp_db: num(lib.findTruePeak(toAB(fin.result.channels, sr))),
      win_starts: w.starts, win_len: 256, windows: saveF32(c.name + '.windows', w.data), node_ms: ms };
    say(c.name, 'lufs', fin.result.lufs, ms, 'ms');
  }

  // ------------------------------------------------------------ D: spectrogram frame (a22)
  for (const pt of [11, 12]) {
    const s = new spec.Spectrogram();
    s.pt = pt; s.fftSize = 1 << pt; s.sampleRate = 48000;
    s.timeDomainBuffer = new Float32Array(s.fftSize);
    s.spectrum = new Float32Array(s.fftSize >> 1).fill(-144);
    s.real = new Float32Array(s.fftSize); s.imag = new Float32Array(s.fftSize); s.window = new Float32Array(s.fftSize);
    s.sinTable = new Float32Array(s.fftSize); s.cosTable = new Float32Array(s.fftSize);
    s.initTables();
    s.imageDataCache = { data: new Uint8Array(256 * 1024 * 4) };
    const x = synthTrack(13, 2, s.fftSize + 3 * 1024, 48000, { amp: 1.0, clickAmp: 0.3 });
    const frames = [];
    const cols = [];
    for (let f = 0; f < 4; f++) {
      const mono = new Float32Array(s.fftSize);
      for (let i = 0; i < s.fftSize; i++) mono[i] = 0.5 * (x[0][f * 1024 + i] + x[1][f * 1024 + i]);
      s.analyser = { getFloatTimeDomainData: (buf) => buf.set(mono) };
      s.process();
      frames.push(s.spectrum.slice());
      const col = new Float32Array(256);
      for (let y = 0; y < 256; y++) col[y] = s.spectrogramBuffer[y * 1024 + 1023];
      cols.push(col);
    }
    const flat = new Float32Array(4 * (s.fftSize >> 1)); frames.forEach((f, i) => flat.set(f, i * (s.fftSize >> 1)));
    const cflat = new Float32Array(4 * 256); cols.forEach((f, i) => cflat.set(f, i * 256));
    golden.cases['spectrogram_' + s.fftSize] = { kind: 'stft', nfft: s.fftSize, hop: 1024, idx: 13, sr: 48000, n: s.fftSize + 3 * 1024,
      opts: { amp: 1.0, clickAmp: 0.3 }, nframes: 4, spectrum_db: saveF32('spec' + s.fftSize + '.db', flat), columns: saveF32('spec' + s.fftSize + '.cols', cflat) };
  }

  fs.writeFileSync(path.join(OUT, 'golden.json'), JSON.stringify(golden, null, 1));
  say('wrote', path.join(OUT, 'golden.json'));
}

// (tests/golden/gen_shapes.mjs reuses the scratch copy and the helpers)
export { prepareScratch, AB, toAB, chansOf, cloneChans, lufsDetail, num, saveF32, saveF64, saveU8, saveChans, stats };
if (process.argv[1] && import.meta.url === pathToFileURL(process.argv[1]).href) main().catch(e => { console.error(e); process.exit(1); });
