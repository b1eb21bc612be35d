// host_test.mjs - checks the JavaScript host layer (web-audio-mastering_amd/js/index.mjs) over the N-API addon.
//   node tests/js/host_test.mjs cpu   : addon loads, surface is complete, error behaviour without a device
//   node tests/js/host_test.mjs gpu   : results through the JS surface equal the reference goldens (tests/golden/data)
// Driven by tests/test_js_host.py; prints one JSON line {"ok": true, "checks": N} or throws.
import fs from 'fs';
import path from 'path';
import { fileURLToPath } from 'url';
import * as wam from '../../web-audio-mastering_amd/js/index.mjs';
import { synthTrack } from '../golden/synth.mjs';

const here = path.dirname(fileURLToPath(import.meta.url));
const DATA = path.join(here, '..', 'golden', 'data');
const golden = JSON.parse(fs.readFileSync(path.join(DATA, 'golden.json'), 'utf8')).cases;
const f32 = (name) => { const b = fs.readFileSync(path.join(DATA, name)); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.length)); };
const u8 = (name) => new Uint8Array(fs.readFileSync(path.join(DATA, name)));
let checks = 0;
function ok(cond, msg) { checks++; if (!cond) throw new Error('FAILED: ' + msg); }
function maxAbsDiff(a, b) { let m = 0; for (let i = 0; i < a.length; i++) { const d = Math.abs(a[i] - b[i]); if (!(d <= m)) m = d; } return m; }
function track(c) {
  const chans = synthTrack(c.idx, c.nch, c.n, c.sr, c.opts || {});
  const ab = new wam.WamAudioBuffer({ numberOfChannels: c.nch, length: c.n, sampleRate: c.sr });
  chans.forEach((x, i) => ab.copyToChannel(x, i));
  return ab;
}

async function cpu() {
  for (const name of ['measureLUFS', 'measureShortTermLUFS', 'measureMomentaryLUFS', 'measureLUFSSeries', 'extractPeaks', 'resampleAudioBuffer', 'analyzeOnLoad', 'findTruePeak', 'normalizeToLUFS', 'applyGain', 'applyLookaheadLimiter', 'applyMasteringSoftClip',
    'applyFinalFilters', 'processHybridDynamic', 'applyExciter', 'applyTapeWarmth', 'applyMultibandTransient', 'detectDCOffsetBuffer',
    'removeDCOffset', 'encodeWAV', 'encodeWAVAsync', 'renderBatch', 'WamWorker', 'DSPWorkerInterface', 'getDSPWorker', 'initDSPWorker'])
    ok(typeof wam[name] === 'function', `export ${name}`);
  const w = new wam.DSPWorkerInterface();
  ok(w.isReady === false, 'not ready before init');
  const ab = new wam.WamAudioBuffer({ numberOfChannels: 2, length: 4800, sampleRate: 48000 });
  let msg = '';
  try { await w.measureLUFS(ab); } catch (e) { msg = e.message; }
  ok(msg === 'Worker not initialized. Call init() first.', `uninitialised message, got '${msg}'`);   // worker-interface.js:95
  ok(wam.getDSPWorker() === wam.getDSPWorker(), 'singleton');
  // no device in this container: init must fail loudly, nothing falls back to JavaScript
  const { createRequire } = await import('module');
  const native = createRequire(import.meta.url)(path.join(here, '..', '..', 'web-audio-mastering_amd', 'napi', 'wam.node'));
  if (native.deviceCount() === 0) {
    let threw = false;
    try { wam.measureLUFS(ab); } catch (e) { threw = true; }
    ok(threw, 'measureLUFS throws without a device');
    let rejected = false;
    try { await w.init(); } catch (e) { rejected = true; }
    ok(rejected && !w.isReady, 'init rejects without a device');
  }
}

async function gpu() {
  const w = await wam.initDSPWorker();
  ok(w.isReady, 'ready');
  // meters (lufs.js / true-peak.js goldens)
  for (const name of ['meter_10s_48k', 'meter_gated_48k', 'meter_mono_44k', 'meter_short', 'meter_silence']) {
    const c = golden[name]; const ab = track(c);
    const num = (v) => typeof v === 'string' ? ({ inf: Infinity, '-inf': -Infinity, nan: NaN })[v] : v;
    const l = wam.measureLUFS(ab), p = wam.findTruePeak(ab);
    const gl = num(c.lufs), gp = num(c.tp_db);
    ok(Number.isFinite(gl) ? Math.abs(l - gl) <= 0.01 : l === gl, `${name} lufs ${l} vs ${gl}`);
    ok(Number.isFinite(gp) ? Math.abs(p - gp) <= 0.01 : p === gp, `${name} tp ${p} vs ${gp}`);
    const r = await w.measureLUFS(ab);
    ok(Object.is(r.lufs, l), `${name} async lufs equals sync`);
    for (const [pos, v] of (c.short_term || [])) { const g = wam.measureShortTermLUFS(ab, pos), want = num(v); ok(Number.isFinite(want) ? Math.abs(g - want) <= 0.01 : g === want, `${name} short-term @${pos}`); }
    for (const [pos, v] of (c.momentary || [])) { const g = wam.measureMomentaryLUFS(ab, pos), want = num(v); ok(Number.isFinite(want) ? Math.abs(g - want) <= 0.01 : g === want, `${name} momentary @${pos}`); }
  }
  // 8f rank 4: loudness series and waveform peaks against the reference run (tests/golden/gen_series.mjs)
  { const S = JSON.parse(fs.readFileSync(path.join(DATA, 'series.json'), 'utf8')).cases;
    const num = (v) => typeof v === 'string' ? ({ inf: Infinity, '-inf': -Infinity, nan: NaN })[v] : v;
    const c = S['series_12s_48k']; const ab = track(c);
    for (const [kind, key, tol] of [['short-term', 'short_term', 1e-6], ['momentary', 'momentary', 1e-9]]) {
      const got = wam.measureLUFSSeries(ab, c.positions, kind);
      let good = got.length === c.positions.length;
      for (let i = 0; good && i < got.length; i++) { const w = num(c[key][i]); good = Number.isFinite(w) ? Math.abs(got[i] - w) <= tol : got[i] === w; }
      ok(good, `series ${kind}`);
    }
    const p = S['peaks_full']; const pab = track({ idx: p.idx, nch: p.nch, n: p.n_synth, sr: p.sr, opts: p.opts });
    const pk = wam.extractPeaks(pab, p.num_peaks), want = f32(p.peaks);
    ok(Array.isArray(pk) && pk.length === want.length && pk.every((v, i) => v === want[i]), 'extractPeaks identical'); }
  // the load-time bundle equals the separate calls the app makes (app.js:823-872)
  { const c = golden['meter_10s_48k']; const ab = track(c);
    const a = wam.analyzeOnLoad(ab, -14);
    const dcInfo = wam.detectDCOffsetBuffer(ab);
    ok(a.dcOffset.significant === dcInfo.significant && Math.abs(a.dcOffset.percent - dcInfo.average.percent) < 1e-9, 'analyzeOnLoad dc');
    const base = a.dcOffset.removed ? a.originalBuffer : ab;
    ok(Math.abs(a.originalLufs - wam.measureLUFS(base)) <= 1e-9 && Math.abs(a.originalTruePeak - wam.findTruePeak(base)) <= 1e-9, 'analyzeOnLoad meters');
    const nrm = wam.normalizeToLUFS(base, -14);
    ok(maxAbsDiff(a.normalizedBuffer.getChannelData(0), nrm.getChannelData(0)) === 0, 'analyzeOnLoad normalised buffer equals normalizeToLUFS'); }
  // sample-rate conversion (self-defined, unpinned): length rule of renderer.js:178, early return, a sine stays a sine
  { const sr = 48000, n = 24000; const ab = new wam.WamAudioBuffer({ numberOfChannels: 1, length: n, sampleRate: sr });
    const x = ab.getChannelData(0); for (let i = 0; i < n; i++) x[i] = 0.5 * Math.sin(2 * Math.PI * 1000 * i / sr);
    ok(await wam.resampleAudioBuffer(ab, sr) === ab && await wam.resampleAudioBuffer(ab, 0) === ab, 'resample early return');
    const r = await wam.resampleAudioBuffer(ab, 44100);
    ok(r.sampleRate === 44100 && r.length === Math.ceil(ab.duration * 44100), 'resample length');
    let e = 0; const y = r.getChannelData(0); for (let i = 300; i < r.length - 300; i++) e = Math.max(e, Math.abs(y[i] - 0.5 * Math.sin(2 * Math.PI * 1000 * i / 44100)));
    ok(e <= 2e-5, `resampled sine error ${e}`); }
  // the library twins: applyLookaheadLimiter (limiter.js:155) and normalizeToLUFS with / without its limiter branch (normalizer.js:38)
  { const L = JSON.parse(fs.readFileSync(path.join(DATA, 'libtwin.json'), 'utf8')).cases;
    for (const name of ['lib_limiter_hot', 'lib_limiter_mono']) {
      const c = L[name]; const r = wam.applyLookaheadLimiter(track(c), c.ceiling_lin);
      for (let ch = 0; ch < c.nch; ch++) ok(maxAbsDiff(r.getChannelData(ch), f32(c.out[ch])) <= 1e-5, `${name} ch${ch}`);
    }
    for (const name of ['lib_normalize_limited', 'lib_normalize_plain']) {
      const c = L[name]; const r = wam.normalizeToLUFS(track(c), c.target, c.ceiling_db);
      for (let ch = 0; ch < c.nch; ch++) { const d = maxAbsDiff(r.getChannelData(ch), f32(c.out[ch])); ok(d <= 1e-5, `${name} ch${ch} max abs diff ${d}`); }
    } }
  // full chain through the promise API, progress strings in worker order
  for (const name of ['chain_defaults_1s', 'chain_mono_1s', 'chain_preview_1s']) {
    const c = golden[name]; const ab = track(c);
    const seen = [];
    const r = await w.renderFullChain(ab, c.settings, c.mode, (p, s) => seen.push([p, s]));
    for (let ch = 0; ch < c.nch; ch++) {
      const d = maxAbsDiff(r.audioBuffer.getChannelData(ch), f32(c.out[ch]));
      ok(d <= 1e-5, `${name} ch${ch} max abs diff ${d}`);
    }
    ok(Math.abs(r.lufs - c.lufs) <= 0.01, `${name} lufs`);
    // the worker's whole PROGRESS stream, the limiter's own messages included (as in tests/test_gpu_chain.py)
    const want = c.progress;
    for (let k = 0; k < 5 && seen.length < want.length; k++) await new Promise((res) => setTimeout(res, 10));   // let queued progress calls drain
    ok(seen.length === want.length, `${name} progress count ${seen.length} vs ${want.length}`);
    for (let i = 0; i < seen.length; i++) ok(seen[i][1] === want[i][1] && Math.abs(seen[i][0] - want[i][0]) < 1e-12, `${name} progress[${i}] ${seen[i]} vs ${want[i]}`);
  }
  // FIFO: three requests issued together resolve in order
  { const c = golden['chain_core_1s']; const ab = track(c); const order = [];
    await Promise.all([0, 1, 2].map((i) => w.renderFullChain(ab, c.settings, c.mode).then(() => order.push(i))));
    ok(order.join() === '0,1,2', `fifo order ${order}`); }
  // 24-bit WAV
  { const c = golden['pcm24_edges']; const ab = new wam.WamAudioBuffer({ numberOfChannels: 2, length: c.n, sampleRate: 48000 });
    c.in.forEach((nme, i) => ab.copyToChannel(f32(nme), i));
    const wav = wam.encodeWAV(ab, 48000, 24);
    ok(Buffer.from(wav.subarray(0, 44)).toString('hex') === c.header_hex, 'wav header');
    const g = u8(c.out); let same = wav.length === 44 + g.length;
    for (let i = 0; same && i < g.length; i++) same = wav[44 + i] === g[i];
    ok(same, 'pcm24 bytes identical'); }
  // 16-bit WAV: 'none' is deterministic (reference golden); dithered modes with the reference's own draws supplied
  { const c16 = JSON.parse(fs.readFileSync(path.join(DATA, 'pcm16.json'), 'utf8')).cases;
    for (const name of ['hot_none', 'hot_tpdf', 'edge_noise_shaped', 'mono_tpdf']) {
      const c = c16[name];
      const ab = new wam.WamAudioBuffer({ numberOfChannels: c.nch, length: c.n, sampleRate: c.sr });
      c.input.forEach((nme, i) => ab.copyToChannel(f32(nme), i));
      const wav = wam.encodeWAV(ab, c.sr, 16, { ditherMode: c.mode, ditherSource: c.mode === 'none' ? undefined : f32(c.dither) });
      const g = u8(c.out); let same = wav.length === 44 + g.length && Buffer.from(wav.subarray(0, 44)).toString('hex') === c.header_hex;
      for (let i = 0; same && i < g.length; i++) same = wav[44 + i] === g[i];
      ok(same, `pcm16 ${name} bytes identical`);
    }
    const c = c16['mono_tpdf'];
    const ab = new wam.WamAudioBuffer({ numberOfChannels: 1, length: c.n, sampleRate: c.sr }); ab.copyToChannel(f32(c.input[0]), 0);
    const w1 = wam.encodeWAV(ab, c.sr, 16, { ditherSeed: 7 }), w2 = wam.encodeWAV(ab, c.sr, 16, { ditherSeed: 7 }), w3 = wam.encodeWAV(ab, c.sr, 16, { ditherSource: 'math-random' });
    ok(Buffer.compare(Buffer.from(w1), Buffer.from(w2)) === 0 && w3.length === w1.length, 'pcm16 seeded dither is reproducible'); }
  // The reference's own hot-path tests (web/ui/encoder.test.js:35-92), restated against the shim with a duck-typed buffer that,
  // like the reference's fixture (:4-12), has numberOfChannels, sampleRate and getChannelData - and NO `length` field.
  { const duck = (channels, sampleRate = 48000) => ({ numberOfChannels: channels.length, sampleRate, getChannelData(i) { return channels[i]; } });
    const header = (wav) => { const v = new DataView(wav.buffer, wav.byteOffset, wav.byteLength); return { sampleRate: v.getUint32(24, true), bitDepth: v.getUint16(34, true), dataSize: v.getUint32(40, true) }; };
    const pcm16 = (wav) => { const v = new DataView(wav.buffer, wav.byteOffset, wav.byteLength); const o = new Int16Array((wav.byteLength - 44) / 2); for (let i = 0; i < o.length; i++) o[i] = v.getInt16(44 + 2 * i, true); return o; };
    // :36-44 target sample rate and bit depth in the header, from a 48 k buffer
    { const wav = await wam.encodeWAVAsync(duck([new Float32Array([0, 0.25, -0.25, 0.5])], 48000), 44100, 16, { ditherMode: 'none' });
      const h = header(wav);
      ok(h.sampleRate === 44100 && h.bitDepth === 16 && h.dataSize === 8, `encoder.test: header ${JSON.stringify(h)}`);
      const q = pcm16(wav); ok(q[0] === 0 && q[1] === Math.round(0.25 * 32767) && q[2] === Math.round(-0.25 * 32767) && q[3] === Math.round(0.5 * 32767), `encoder.test: 16-bit none samples ${Array.from(q)}`); }
    // :46-54 shouldCancel
    { let msg = ''; try { await wam.encodeWAVAsync(duck([new Float32Array(32768)], 48000), 48000, 16, { shouldCancel: () => true }); } catch (e) { msg = e.message; }
      ok(msg === 'Cancelled', `encoder.test: shouldCancel rejects with 'Cancelled', got '${msg}'`); }
    // :56-65 16-bit 'none' is deterministic
    { const src = new Float32Array(512); src.fill(0.00123); const a = wam.encodeWAV(duck([src]), 48000, 16, { ditherMode: 'none' }), b = wam.encodeWAV(duck([src]), 48000, 16, { ditherMode: 'none' });
      ok(a.length === 44 + 1024 && Buffer.compare(Buffer.from(a), Buffer.from(b)) === 0, 'encoder.test: none is deterministic'); }
    // :67-80 TPDF differs from none - with the device generator, and with the reference's own Math.random draws on a length-less buffer
    { const src = new Float32Array(4096); src.fill(0.00001);
      const none = pcm16(wam.encodeWAV(duck([src]), 48000, 16, { ditherMode: 'none' }));
      for (const opt of [{ ditherMode: 'tpdf' }, { ditherMode: 'tpdf', ditherSource: 'math-random' }]) {
        const t = pcm16(wam.encodeWAV(duck([src]), 48000, 16, opt));
        ok(t.length === none.length && none.some((v, i) => v !== t[i]), `encoder.test: tpdf differs from none (${opt.ditherSource || 'device'})`);
        ok(t.every((v) => v >= -1 && v <= 2), 'encoder.test: tpdf stays within one LSB of the input level');
      } }
    // :82-91 24-bit ignores the dither mode
    { const src = new Float32Array(1024); src.fill(0.03125);
      const a = wam.encodeWAV(duck([src]), 48000, 24, { ditherMode: 'none' }), b = wam.encodeWAV(duck([src]), 48000, 24, { ditherMode: 'noise-shaped' });
      ok(a.length === 44 + 3072 && Buffer.compare(Buffer.from(a), Buffer.from(b)) === 0, 'encoder.test: 24-bit ignores dither mode'); }
    // encodeWAVAsync's chunk loop (encoder.js:189-243): cancel polled before every chunk, progress end / numSamples after every chunk,
    // a yield between chunks, a closing 1, chunkSize floor of 1024; the bytes equal encodeWAV's
    { const n = 5000; const src = new Float32Array(n); for (let i = 0; i < n; i++) src[i] = Math.sin(i * 0.01) * 0.7;
      const prog = []; let polls = 0, yields = 0; const t0 = setInterval(() => {}, 1000);
      const tick = () => { yields++; };
      const pr = wam.encodeWAVAsync(duck([src, src]), 48000, 24, { chunkSize: 100, onProgress: (p) => { prog.push(p); setTimeout(tick, 0); }, shouldCancel: () => { polls++; return false; } });
      const wav = await pr; clearInterval(t0);
      const want = [1024 / n, 2048 / n, 3072 / n, 4096 / n, 1, 1];
      ok(prog.length === want.length && prog.every((v, i) => v === want[i]), `encodeWAVAsync progress ${prog}`);
      ok(polls === 5, `encodeWAVAsync polls shouldCancel before every chunk (${polls})`);
      ok(yields >= 4, `encodeWAVAsync yields between chunks (${yields} timers ran before it resolved)`);
      ok(Buffer.compare(Buffer.from(wav), Buffer.from(wam.encodeWAV(duck([src, src]), 48000, 24))) === 0, 'encodeWAVAsync bytes equal encodeWAV');
      let seen = 0, msg = '';
      try { await wam.encodeWAVAsync(duck([src]), 48000, 16, { chunkSize: 1024, shouldCancel: () => ++seen === 3, onProgress: () => {} }); } catch (e) { msg = e.message; }
      ok(msg === 'Cancelled' && seen === 3, `encodeWAVAsync cancels at the third chunk (${seen}, '${msg}')`);
      const empty = await wam.encodeWAVAsync(duck([new Float32Array(0)]), 48000, 16, {});
      ok(empty.length === 44 && header(empty).dataSize === 0, 'encodeWAVAsync of an empty buffer is a header'); } }
  // submit-time snapshot (worker-interface.js:116-132 slices every channel before posting): a caller that edits its buffer while the
  // request waits in the queue must not change the result
  { const c = golden['chain_core_1s']; const ab = track(c), ab2 = track(c);
    const first = w.renderFullChain(ab, c.settings, c.mode);           // occupies the queue
    const second = w.renderFullChain(ab2, c.settings, c.mode);         // waits behind it
    ab2.getChannelData(0).fill(0); ab2.getChannelData(1).fill(0);      // edited after submission
    const [r1, r2] = await Promise.all([first, second]);
    ok(maxAbsDiff(r2.audioBuffer.getChannelData(0), r1.audioBuffer.getChannelData(0)) === 0 && maxAbsDiff(r2.audioBuffer.getChannelData(1), r1.audioBuffer.getChannelData(1)) === 0,
       'a queued request renders the buffer as it was when it was submitted');
    const wk = new wam.WamWorker(); const ch = [track(c).getChannelData(0), track(c).getChannelData(1)];
    const reply = await new Promise((resolve) => { wk.onmessage = (e) => { if (e.data.type !== 'PROGRESS') resolve(e.data); };
      wk.postMessage({ type: 'RENDER_FULL_CHAIN', id: 1, data: { channels: ch, sampleRate: c.sr, settings: c.settings, mode: c.mode } }); ch[0].fill(0); ch[1].fill(0); });
    ok(reply.success && maxAbsDiff(reply.result.channels[0], r1.audioBuffer.getChannelData(0)) === 0, 'WamWorker.postMessage snapshots the channels when the message is posted');
    wk.terminate(); }
  // the Worker-shaped adapter: the reference's message protocol (dsp-worker.js:1-9)
  { const c = golden['chain_core_1s']; const ab = track(c);
    const wk = new wam.WamWorker();
    const got = await new Promise((resolve) => {
      const msgs = [];
      wk.onmessage = (e) => { msgs.push(e.data); if (e.data.type !== 'PROGRESS' && e.data.id === 8) resolve(msgs); };
      wk.postMessage({ type: 'RENDER_FULL_CHAIN', id: 7, data: { channels: [ab.getChannelData(0), ab.getChannelData(1)], sampleRate: c.sr, settings: c.settings, mode: c.mode } });
      wk.postMessage({ type: 'NO_SUCH_TYPE', id: 8, data: {} });
    });
    const reply = got.find((m) => m.id === 7 && m.type !== 'PROGRESS');
    ok(reply && reply.success === true && Math.abs(reply.result.lufs - c.lufs) <= 0.01, 'worker protocol: chain reply');
    ok(maxAbsDiff(reply.result.channels[0], f32(c.out[0])) <= 1e-5, 'worker protocol: chain audio');
    const prog = got.filter((m) => m.id === 7 && m.type === 'PROGRESS');
    ok(prog.length >= 5 && prog[prog.length - 1].progress === 1.0 && prog[prog.length - 1].status === 'Complete', 'worker protocol: progress ends with Complete');
    ok(got.indexOf(reply) > got.indexOf(prog[prog.length - 1]), 'worker protocol: reply after last progress');
    const bad = got.find((m) => m.id === 8);
    ok(bad.success === false && bad.error === 'Unknown message type: NO_SUCH_TYPE', `worker protocol: unknown type, got '${bad.error}'`);
    wk.terminate(); }
  // NORMALIZE / APPLY_LIMITER: the messages' own PROGRESS streams (dsp-worker.js:458,471,491 and :302,364,427,448), through the
  // Worker-shaped adapter and through the promise interface; meters with more than two channels and a non-4-hop sample rate
  { const SH = JSON.parse(fs.readFileSync(path.join(DATA, 'shapes.json'), 'utf8')).cases;
    const wk = new wam.WamWorker();
    const ask = (type, data) => new Promise((resolve) => {
      const msgs = []; wk.onmessage = (e) => { msgs.push(e.data); if (e.data.type !== 'PROGRESS') resolve(msgs); };
      wk.postMessage({ type, id: 21, data }); });
    const sameProgress = (got, want, what) => {
      ok(got.length === want.length, `${what}: ${got.length} progress messages vs ${want.length}`);
      for (let i = 0; i < Math.min(got.length, want.length); i++) ok(got[i][1] === want[i][1] && Math.abs(got[i][0] - want[i][0]) < 1e-12, `${what} progress[${i}] ${got[i]} vs ${want[i]}`); };
    for (const name of ['msg_normalize_limited', 'msg_normalize_plain']) {
      const c = SH[name]; const ab = track(c);
      const msgs = await ask('NORMALIZE', { channels: [ab.getChannelData(0), ab.getChannelData(1)], sampleRate: c.sr, targetLUFS: c.target, ceilingDB: c.ceiling_db });
      const reply = msgs[msgs.length - 1];
      ok(reply.success === true && reply.result.limiterApplied === c.limiterApplied && Math.abs(reply.result.currentLUFS - c.currentLUFS) <= 0.01, `${name} reply`);
      ok(maxAbsDiff(reply.result.channels[1], f32(c.out[1])) <= 1e-5, `${name} audio`);
      sameProgress(msgs.filter((m) => m.type === 'PROGRESS').map((m) => [m.progress, m.status]), c.progress, `worker ${name}`);
      const seen = []; const r = await w.normalize(ab, c.target, c.ceiling_db, (p, s) => seen.push([p, s]));
      for (let k = 0; k < 5 && seen.length < c.progress.length; k++) await new Promise((res) => setTimeout(res, 10));
      sameProgress(seen, c.progress, `interface ${name}`);
      ok(maxAbsDiff(r.audioBuffer.getChannelData(0), f32(c.out[0])) <= 1e-5, `${name} interface audio`);
    }
    { const c = SH['msg_limiter']; const ab = track(c);
      const msgs = await ask('APPLY_LIMITER', { channels: [ab.getChannelData(0), ab.getChannelData(1)], sampleRate: c.sr, ceilingLinear: c.ceiling_lin });
      sameProgress(msgs.filter((m) => m.type === 'PROGRESS').map((m) => [m.progress, m.status]), c.progress, 'worker APPLY_LIMITER');
      ok(maxAbsDiff(msgs[msgs.length - 1].result.channels[0], f32(c.out[0])) <= 1e-5, 'worker APPLY_LIMITER audio');
      const seen = []; const r = await w.applyLimiter(ab, { ceilingLinear: c.ceiling_lin }, (p, s) => seen.push([p, s]));
      for (let k = 0; k < 5 && seen.length < c.progress.length; k++) await new Promise((res) => setTimeout(res, 10));
      sameProgress(seen, c.progress, 'interface applyLimiter');
      ok(maxAbsDiff(r.audioBuffer.getChannelData(1), f32(c.out[1])) <= 1e-5, 'interface applyLimiter audio'); }
    wk.terminate();
    for (const name of ['meter_3ch_48k', 'meter_11025']) {
      const c = SH[name]; const ab = track(c);
      ok(Math.abs(wam.measureLUFS(ab) - c.lufs) <= 0.01 && Math.abs(wam.findTruePeak(ab) - c.tp_db) <= 0.01, `${name} through the JS surface`);
      ok(Math.abs(wam.measureShortTermLUFS(ab, c.short_term[1][0]) - c.short_term[1][1]) <= 0.01, `${name} short-term`);
    } }
  // processHybridDynamic presets (dynamic-processor.js:595-607): named, unnamed (= 'gentle') and unknown (= 'gentle')
  { const SH = JSON.parse(fs.readFileSync(path.join(DATA, 'shapes.json'), 'utf8')).cases;
    for (const name of ['deharsh_gentle', 'deharsh_aggressive', 'deharsh_aggressive_mono44']) {
      const c = SH[name]; const y = wam.processHybridDynamic(track(c), c.preset);
      ok(maxAbsDiff(y.getChannelData(0), f32(c.out[0])) <= 1e-5, `${name} vs reference golden`);
    }
    const c = SH['deharsh_gentle'];
    ok(maxAbsDiff(wam.processHybridDynamic(track(c)).getChannelData(1), f32(c.out[1])) <= 1e-5, 'no preset = gentle');
    ok(maxAbsDiff(wam.processHybridDynamic(track(c), 'nope').getChannelData(1), f32(c.out[1])) <= 1e-5, 'unknown preset = gentle');
    ok(maxAbsDiff(wam.processHybridDynamic(track(c), 'mastering').getChannelData(1), f32(c.out[1])) > 1e-3, 'mastering differs from gentle'); }
  // multi-device batch: two entries on device 0, three tracks - each track equals its single render bit for bit
  { const c = golden['chain_core_1s'];
    const abs = [0, 1, 2].map((k) => track(Object.assign({}, c, { idx: c.idx + k })));
    const batch = await wam.renderBatch(abs, c.settings, c.mode, { devices: [0, 0] });
    ok(batch.length === 3, 'renderBatch returns one result per track');
    for (let k = 0; k < 3; k++) {
      const one = await w.renderFullChain(abs[k], c.settings, c.mode);
      ok(maxAbsDiff(batch[k].audioBuffer.getChannelData(0), one.audioBuffer.getChannelData(0)) === 0 &&
         maxAbsDiff(batch[k].audioBuffer.getChannelData(1), one.audioBuffer.getChannelData(1)) === 0 && batch[k].lufs === one.lufs, `renderBatch track ${k} equals renderFullChain`);
    }
    ok(maxAbsDiff(batch[0].audioBuffer.getChannelData(0), f32(c.out[0])) <= 1e-5, 'renderBatch track 0 vs reference golden');
    let msg = ''; try { await wam.renderBatch(abs, c.settings, c.mode, { devices: [99] }); } catch (e) { msg = e.message; }
    ok(/out of range/.test(msg), `renderBatch rejects a bad device, got '${msg}'`); }
  // normalize + terminate semantics
  { const c = golden['meter_10s_48k']; const ab = track(c);
    const r = await w.normalize(ab, -14, -1);
    ok(Math.abs(r.currentLUFS - c.lufs) <= 0.01 && typeof r.limiterApplied === 'boolean', 'normalize keys');
    const same = wam.normalizeToLUFS(new wam.WamAudioBuffer({ numberOfChannels: 2, length: 48000, sampleRate: 48000 }), -14, -1);
    ok(same.length === 48000, 'silent normalise returns input');
    const pend = w.measureLUFS(ab); w.terminate();
    let msg = ''; try { await pend; } catch (e) { msg = e.message; }
    ok(msg === 'Worker terminated', `terminate rejects pending, got '${msg}'`); }
}

const mode = process.argv[2] || 'cpu';
(mode === 'gpu' ? gpu() : cpu()).then(() => { console.log(JSON.stringify({ ok: true, mode, checks })); })
  .catch((e) => { console.error(e.stack || String(e)); process.exit(1); });
