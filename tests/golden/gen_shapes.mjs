// gen_shapes.mjs - golden vectors for the shapes and message streams round 1 refused or dropped, produced by running the
// REFERENCE's own JavaScript under Node (same scratch copy and three mechanical edits as gen_golden.mjs):
//   * meters with more than two channels (web/lib/dsp/lufs.js:28-65, true-peak.js:46-71 loop over numberOfChannels);
//   * a sample rate whose 400 ms block is not four 100 ms hops (11,025 Hz: block 4410, hop 1102; lufs.js:53-57);
//   * the look-ahead limiter, the soft clipper and the chain at 176.4 / 192 kHz (dsp-worker.js:299-309: 3 ms look-ahead
//     = 576 samples, 10 ms RMS window = 1920 samples at 192 kHz);
//   * the PROGRESS streams of the NORMALIZE and APPLY_LIMITER messages (dsp-worker.js:458,471,491 and :302,364,427,448).
//   * processHybridDynamic's 'gentle' and 'aggressive' presets (dynamic-processor.js:538-566; substitute DFT as in gen_golden.mjs);
//   node tests/golden/gen_shapes.mjs      (needs /root/reference; nothing of it is written into the repo)
import fs from 'fs';
import path from 'path';
import { fileURLToPath, pathToFileURL } from 'url';
import { synthTrack } from './synth.mjs';
import { prepareScratch, AB, toAB, chansOf, cloneChans, lufsDetail, num, saveF32, saveF64, saveU8, saveChans, stats } from './gen_golden.mjs';

const here = path.dirname(fileURLToPath(import.meta.url));
const OUT = path.join(here, 'data');

async function main() {
  const dir = prepareScratch();
  process.on('exit', () => { try { fs.rmdirSync(dir, { recursive: true }); } catch (e) { /* ignore */ } });
  const log = console.log; console.log = () => {}; console.warn = () => {};
  globalThis.AudioBuffer = AB;
  const posted = [];
  globalThis.self = { postMessage: (m) => posted.push(m) };
  const imp = (p) => import(pathToFileURL(path.join(dir, p)).href);
  const lib = await imp('lib/dsp/index.js');
  const worker = await imp('workers/dsp-worker.js');
  const onmessage = globalThis.self.onmessage;
  globalThis.AudioBuffer = AB;
  const cases = {};

  async function send(type, data) {
    posted.length = 0;
    await onmessage({ data: { type, id: 1, data } });
    const fin = posted.filter(m => m.type !== 'PROGRESS').pop();
    const progress = posted.filter(m => m.type === 'PROGRESS').map(m => [m.progress, m.status]);
    if (!fin || !fin.success) throw new Error(type + ' failed: ' + (fin && fin.error));
    return { result: fin.result, progress };
  }

  // ---- meters: channel counts and sample rates
  const meterCases = [
    { name: 'meter_3ch_48k', idx: 40, nch: 3, sr: 48000, sec: 4, opts: { clickAmp: 0.4 } },
    { name: 'meter_6ch_44k', idx: 41, nch: 6, sr: 44100, sec: 2, opts: { amp: 0.7 } },
    { name: 'meter_11025', idx: 42, nch: 2, sr: 11025, sec: 6, opts: { clickAmp: 0.3, quietFrom: 22050, quietTo: 33075, quietGain: 0.01 } },
    { name: 'meter_11025_mono', idx: 43, nch: 1, sr: 11025, sec: 3, opts: { amp: 0.5 } },
    { name: 'meter_44056', idx: 44, nch: 2, sr: 44056, sec: 3, opts: {} },              // block 17622, hop 4405: 2 samples over
    { name: 'meter_192k', idx: 45, nch: 2, sr: 192000, sec: 1.5, opts: { clickAmp: 0.5 } }
  ];
  for (const c of meterCases) {
    const n = Math.round(c.sec * c.sr);
    const ch = synthTrack(c.idx, c.nch, n, c.sr, c.opts);
    const ab = toAB(ch, c.sr);
    const d = lufsDetail(lib, ab);
    const rec = { kind: 'meter', idx: c.idx, nch: c.nch, sr: c.sr, n, opts: c.opts, lufs: num(lib.measureLUFS(ab)), tp_db: num(lib.findTruePeak(ab)),
      lufs_worker: num(worker.measureLUFSFromChannels(ch, c.sr)), tp_db_worker: num(worker.findTruePeakFromChannels(ch)),
      nblocks: d.blocks.length, block_ms: saveF64('shapes.' + c.name + '.block_ms', d.blocks), gate_mask: saveU8('shapes.' + c.name + '.gate_mask', Uint8Array.from(d.mask)),
      short_term: [0, 1.2].map(p => [p, num(lib.measureShortTermLUFS(ab, p))]), momentary: [0, 0.9, c.sec - 0.3].map(p => [p, num(lib.measureMomentaryLUFS(ab, p))]) };
    cases[c.name] = rec;
    log(c.name, rec.lufs, rec.tp_db, rec.nblocks);
  }

  // ---- limiter / soft clip / chain at high sample rates
  const base = { normalizeLoudness: true, targetLufs: -14, truePeakLimit: true, truePeakCeiling: -1, cleanLowEnd: true,
    glueCompression: true, deharsh: false, stereoWidth: 100, centerBass: true, cutMud: false, addAir: true, tapeWarmth: true,
    autoLevel: false, addPunch: true, inputGain: 0, eqLow: 0, eqLowMid: 0, eqMid: 0, eqHighMid: 0, eqHigh: 0 };
  const ceil = Math.pow(10, -1 / 20);
  for (const sr of [176400, 192000]) {
    const n = Math.round(0.3 * sr), tag = 'hi' + sr;
    const opts = { amp: 2.4, clickAmp: 0.9 };
    const x = synthTrack(46, 2, n, sr, opts);
    const r = await send('APPLY_LIMITER', { channels: cloneChans(x), sampleRate: sr, ceilingLinear: ceil });
    cases[tag + '_limiter'] = { kind: 'limiter', idx: 46, nch: 2, sr, n, opts, ceiling_lin: ceil, out: saveChans('shapes.' + tag + '.lim', r.result.channels), progress: r.progress };
    const np = worker.applyLookaheadLimiterToChannels(cloneChans(x), sr, 1, ceil, 3, 100, 3, false);
    cases[tag + '_limiter_np'] = { kind: 'limiter', idx: 46, nch: 2, sr, n, opts, ceiling_lin: ceil, preserve: false, out: saveChans('shapes.' + tag + '.lim_np', np) };
    const lt = lib.applyLookaheadLimiter(toAB(x, sr), ceil);
    cases[tag + '_limiter_lib'] = { kind: 'limiter_lib', idx: 46, nch: 2, sr, n, opts, ceiling_lin: ceil, out: saveChans('shapes.' + tag + '.lim_lib', chansOf(lt)) };
    const sc = lib.applyMasteringSoftClip(toAB(x, sr), { ceiling: -1, lookaheadMs: 0.5, releaseMs: 10, drive: 1.5 });
    cases[tag + '_softclip'] = { kind: 'softclip', idx: 46, nch: 2, sr, n, opts, out: saveChans('shapes.' + tag + '.sc', chansOf(sc)) };
    const copts = { amp: 1.6, clickAmp: 0.6 };
    const nc = Math.round(0.6 * sr);                 // longer than one 400 ms block, so that the chain's loudness stages measure
    const y = synthTrack(47, 2, nc, sr, copts);
    const ch = await send('RENDER_FULL_CHAIN', { channels: cloneChans(y), sampleRate: sr, settings: base, mode: 'export' });
    cases[tag + '_chain'] = { kind: 'chain', idx: 47, nch: 2, sr, n: nc, opts: copts, settings: base, mode: 'export', dft: 'unused', lufs: num(ch.result.lufs),
      out: saveChans('shapes.' + tag + '.chain', ch.result.channels), progress: ch.progress };
    log(tag, 'chain lufs', ch.result.lufs);
  }
  // the chain at 11,025 Hz (stages whose corner lies above Nyquist off; normalisation = the non-4-hop meter inside the chain)
  {
    const sr = 11025, n = sr * 2, copts = { amp: 1.6, clickAmp: 0.6 };
    const settings = Object.assign({}, base, { addAir: false, tapeWarmth: false, addPunch: false });
    const y = synthTrack(48, 2, n, sr, copts);
    const ch = await send('RENDER_FULL_CHAIN', { channels: cloneChans(y), sampleRate: sr, settings, mode: 'export' });
    cases.chain_11025 = { kind: 'chain', idx: 48, nch: 2, sr, n, opts: copts, settings, mode: 'export', dft: 'unused', lufs: num(ch.result.lufs),
      out: saveChans('shapes.chain_11025', ch.result.channels), progress: ch.progress };
  }

  // ---- NORMALIZE / APPLY_LIMITER message streams at 48 kHz (progress lists + results)
  {
    const sr = 48000, n = 24000;
    const o2 = { clickAmp: 0.5 };
    const x = synthTrack(33, 2, n, sr, o2);
    for (const [name, target] of [['msg_normalize_limited', -9], ['msg_normalize_plain', -30]]) {
      const r = await send('NORMALIZE', { channels: cloneChans(x), sampleRate: sr, targetLUFS: target, ceilingDB: -1 });
      cases[name] = { kind: 'normalize', idx: 33, nch: 2, sr, n, opts: o2, target, ceiling_db: -1, progress: r.progress, out: saveChans('shapes.' + name, r.result.channels),
        currentLUFS: num(r.result.currentLUFS), finalLUFS: num(r.result.finalLUFS), peakDB: num(r.result.peakDB), gainApplied: num(r.result.gainApplied),
        limiterApplied: !!r.result.limiterApplied };
    }
    const z = [new Float32Array(n), new Float32Array(n)];
    const rs = await send('NORMALIZE', { channels: z, sampleRate: sr, targetLUFS: -14, ceilingDB: -1 });
    cases.msg_normalize_silent = { kind: 'normalize', nch: 2, sr, n, target: -14, ceiling_db: -1, progress: rs.progress, currentLUFS: num(rs.result.currentLUFS),
      finalLUFS: num(rs.result.finalLUFS), peakDB: num(rs.result.peakDB), has_gain: 'gainApplied' in rs.result };
    const hot = synthTrack(31, 2, n, sr, { amp: 2.4, clickAmp: 0.9 });
    const rl = await send('APPLY_LIMITER', { channels: cloneChans(hot), sampleRate: sr, ceilingLinear: ceil });
    cases.msg_limiter = { kind: 'limiter', idx: 31, nch: 2, sr, n, opts: { amp: 2.4, clickAmp: 0.9 }, ceiling_lin: ceil, progress: rl.progress, out: saveChans('shapes.msg_limiter', rl.result.channels) };
    const mono = synthTrack(32, 1, n, sr, { amp: 2.4, clickAmp: 0.9 });
    const rm = await send('APPLY_LIMITER', { channels: cloneChans(mono), sampleRate: sr });
    cases.msg_limiter_mono_defaults = { kind: 'limiter', idx: 32, nch: 1, sr, n, opts: { amp: 2.4, clickAmp: 0.9 }, ceiling_lin: 0.891, progress: rm.progress, out: saveChans('shapes.msg_limiter_mono', rm.result.channels) };
  }
  // ---- processHybridDynamic presets other than the chain's 'mastering' (fully wet: the tail after the last frame fades to zero)
  for (const c of [{ name: 'deharsh_gentle', preset: 'gentle', idx: 50, nch: 2, sr: 48000, n: 30000, opts: { amp: 1.8, clickAmp: 0.7 } },
    { name: 'deharsh_aggressive', preset: 'aggressive', idx: 50, nch: 2, sr: 48000, n: 30000, opts: { amp: 1.8, clickAmp: 0.7 } },
    { name: 'deharsh_gentle_mono44', preset: 'gentle', idx: 51, nch: 1, sr: 44100, n: 20000, opts: { amp: 0.6 } },
    { name: 'deharsh_aggressive_mono44', preset: 'aggressive', idx: 51, nch: 1, sr: 44100, n: 20000, opts: { amp: 0.6 } },
    { name: 'deharsh_default_short', preset: undefined, idx: 52, nch: 2, sr: 48000, n: 2500, opts: {} }]) {     // one frame... none: floor((n - 2048) / 512) = 0
    const x = synthTrack(c.idx, c.nch, c.n, c.sr, c.opts);
    const y = c.preset === undefined ? lib.processHybridDynamic(toAB(x, c.sr)) : lib.processHybridDynamic(toAB(x, c.sr), c.preset);
    cases[c.name] = { kind: 'deharsh', preset: c.preset === undefined ? null : c.preset, idx: c.idx, nch: c.nch, sr: c.sr, n: c.n, opts: c.opts, dft: 'substitute',
      out: saveChans('shapes.' + c.name, chansOf(y)) };
  }
  console.log = log;
  fs.writeFileSync(path.join(OUT, 'shapes.json'), JSON.stringify({ generator: 'tests/golden/gen_shapes.mjs', node: process.version, cases }, null, 1));
  console.log('wrote', Object.keys(cases).length, 'cases');
}
main().catch((e) => { console.error(e); process.exit(1); });
