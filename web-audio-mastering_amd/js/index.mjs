// index.mjs - drop-in JavaScript surface of the MI355X mastering backend.
//
// Same names, argument meaning and error behaviour as the reference's JavaScript:
//   - the sync library functions of web/lib/dsp/index.js (measureLUFS, findTruePeak,
//     normalizeToLUFS, applyGain, applyLookaheadLimiter, applyMasteringSoftClip,
//     applyFinalFilters, processHybridDynamic, applyExciter, applyTapeWarmth,
//     applyMultibandTransient, detectDCOffsetBuffer, removeDCOffset) and encodeWAV
//     (web/ui/encoder.js:51);
//   - the promise API of web/workers/worker-interface.js (DSPWorkerInterface,
//     getDSPWorker, initDSPWorker).
// Everything below is argument marshalling; the arithmetic runs in libwam_hip.so through
// the N-API addon (../napi/wam.node).  There is no JavaScript fallback: without the addon
// or a HIP device the calls throw / reject.
import { createRequire } from 'module';
import { fileURLToPath } from 'url';
import path from 'path';

const require = createRequire(import.meta.url);
const here = path.dirname(fileURLToPath(import.meta.url));
let native = null;
function addon() {
  if (!native) {
    try { native = require(path.join(here, '..', 'napi', 'wam.node')); }
    catch (e) { throw new Error(`wam: native addon not available (${e.message}); there is no JavaScript fallback`); }
  }
  return native;
}

// AudioBuffer-like container, same shape as WorkerAudioBuffer (web/workers/dsp-worker.js:36-61)
export class WamAudioBuffer {
  constructor({ numberOfChannels, length, sampleRate }) {
    this.numberOfChannels = numberOfChannels;
    this.length = length;
    this.sampleRate = sampleRate;
    this.duration = length / sampleRate;
    this._channels = [];
    for (let i = 0; i < numberOfChannels; i++) this._channels.push(new Float32Array(length));
  }
  getChannelData(c) { return this._channels[c]; }
  copyToChannel(source, c, start = 0) { this._channels[c].set(source, start); }
  copyFromChannel(dest, c, start = 0) { dest.set(this._channels[c].subarray(start, start + dest.length)); }
}

function chansOf(ab) { const r = []; for (let c = 0; c < ab.numberOfChannels; c++) r.push(ab.getChannelData(c)); return r; }
// Copies of the channels, taken when a request is SUBMITTED: the reference slices every channel before it posts the message
// (_extractChannelData, worker-interface.js:116-132), so what a caller does to its buffer while the request waits in the queue
// does not reach the worker.  (The addon copies once more when the request runs; that copy is the one the device reads.)
function snapshotOf(ab) { return chansOf(ab).map((c) => c.slice()); }
function makeAB(channels, sampleRate) {
  const Ctor = (typeof globalThis.AudioBuffer === 'function') ? globalThis.AudioBuffer : WamAudioBuffer;
  const ab = new Ctor({ numberOfChannels: channels.length, length: channels[0].length, sampleRate });
  channels.forEach((c, i) => ab.copyToChannel(c, i));
  return ab;
}
const stage = (name, ab, params = []) => makeAB(addon().stageSync(name, chansOf(ab), ab.sampleRate, params), ab.sampleRate);

// ---- web/lib/dsp/index.js ---------------------------------------------------------------
export function measureLUFS(audioBuffer, fallbackLufs = -14) {                       // lufs.js:17
  return addon().meterSync(chansOf(audioBuffer), audioBuffer.sampleRate, fallbackLufs).lufs;
}
export function measureShortTermLUFS(audioBuffer, position = 0) {                    // lufs.js:91
  return addon().windowLufsSync(chansOf(audioBuffer), audioBuffer.sampleRate, position, false);
}
export function measureMomentaryLUFS(audioBuffer, position = 0) {                    // lufs.js:128
  return addon().windowLufsSync(chansOf(audioBuffer), audioBuffer.sampleRate, position, true);
}
// the two window measures at many positions in one call (one upload); no counterpart by name in the reference, which polls per position
export function measureLUFSSeries(audioBuffer, positions, kind = 'short-term') {
  return Array.from(addon().lufsSeries(chansOf(audioBuffer), audioBuffer.sampleRate, Float64Array.from(positions), kind === 'momentary'));
}
export function extractPeaks(audioBuffer, numPeaks = 800) {                          // ui/waveform.js:164 (channel 0, plain number[])
  return Array.from(addon().extractPeaks([audioBuffer.getChannelData(0)], numPeaks));
}
// ui/renderer.js:172-187.  The reference renders through Chromium's OfflineAudioContext; here: a self-defined windowed-sinc
// (options.quality 'sinc', default) or linear ('linear') conversion - parity unpinned (include/wam.h).  Same early return.
export async function resampleAudioBuffer(sourceBuffer, targetSampleRate, options = {}) {
  if (!sourceBuffer || !targetSampleRate || sourceBuffer.sampleRate === targetSampleRate) return sourceBuffer;
  return makeAB(addon().resample(chansOf(sourceBuffer), sourceBuffer.sampleRate, targetSampleRate, options.quality === 'linear' ? 0 : 1), targetSampleRate);
}
// What web/app.js:823-872 does on every file load, in one call with one upload: DC check (+ removal when significant), loudness and
// true peak of the result, normalizeToLUFS(buffer, targetLufs).  Returns the fields the app stores in fileState.
export function analyzeOnLoad(decodedBuffer, targetLufs = -14, ceilingDB = -1) {
  const r = addon().analyzeOnLoad(chansOf(decodedBuffer), decodedBuffer.sampleRate, targetLufs, ceilingDB);
  const sr = decodedBuffer.sampleRate;
  const originalBuffer = r.dcRemoved ? makeAB(r.cleaned, sr) : decodedBuffer;
  return { dcOffset: { percent: r.dcPercent, removed: !!r.dcRemoved, significant: !!r.dcSignificant, channels: r.dc },
    originalBuffer, originalLufs: r.originalLufs, originalTruePeak: r.originalTruePeak, originalSampleRate: sr,
    normalizedBuffer: r.measured ? makeAB(r.normalized, sr) : originalBuffer, gainApplied: r.gainDb, limiterApplied: !!r.limiterApplied };
}
export function findTruePeak(audioBuffer) {                                          // true-peak.js:46
  return addon().meterSync(chansOf(audioBuffer), 48000, -14).peakDB;   // the peak does not depend on the rate (findTruePeakFromChannels takes none)
}
export function applyGain(audioBuffer, gainDB) { return stage('gain', audioBuffer, [gainDB]); }   // normalizer.js:103
export function normalizeToLUFS(audioBuffer, targetLUFS = -14, ceilingDB = -1, options = {}) {    // normalizer.js:38
  const { lufs, peakDB } = addon().meterSync(chansOf(audioBuffer), audioBuffer.sampleRate, targetLUFS);
  if (!isFinite(lufs)) return audioBuffer;                    // the reference returns the SAME object
  const gainDB = targetLUFS - lufs;
  const gained = applyGain(audioBuffer, gainDB);
  if (options.skipLimiter) return gained;
  if (peakDB + gainDB > ceilingDB) return applyLookaheadLimiter(gained, Math.pow(10, ceilingDB / 20), 3, 100);
  return gained;
}
export function applyLookaheadLimiter(audioBuffer, ceilingLinear = 0.891, lookaheadMs = 3, releaseMs = 100, kneeDB = 3, preserveTransients = true) {
  // web/lib/dsp/limiter.js:155 - the library twin (the worker's, behind DSPWorkerInterface.applyLimiter and the chain, seeds its
  // soft-knee pass differently: SURVEY.md 0.4)
  return stage('limiterLib', audioBuffer, [ceilingLinear, lookaheadMs, releaseMs, kneeDB, preserveTransients ? 1 : 0]);
}
export function applyMasteringSoftClip(audioBuffer, options = {}) {                  // soft-clipper.js:165
  const { ceiling = -1, lookaheadMs = 0.5, releaseMs = 10, drive = 1.5 } = options;
  return stage('softClip', audioBuffer, [ceiling, lookaheadMs, releaseMs, drive]);
}
export function applyFinalFilters(audioBuffer, options = {}) {                       // final-filters.js:96
  const { highpass = true, lowpass = true } = options;
  return stage('finalFilters', audioBuffer, [highpass ? 1 : 0, lowpass ? 1 : 0]);
}
export function processHybridDynamic(audioBuffer, preset = 'gentle') {               // dynamic-processor.js:595
  const id = { gentle: 0, aggressive: 1, mastering: 2 }[preset];                      // any other name is 'gentle' (:604-606)
  return stage('deharsh', audioBuffer, [id === undefined ? 0 : id]);
}
export function applyExciter(audioBuffer) { return stage('exciter', audioBuffer); }                     // exciter.js:223
export function applyTapeWarmth(audioBuffer) { return stage('tapeWarmth', audioBuffer); }               // multiband-saturation.js:365
export function applyMultibandTransient(audioBuffer) { return stage('multibandTransient', audioBuffer); } // multiband-transient.js:15
export function detectDCOffsetBuffer(audioBuffer) {                                  // dc-offset.js:38
  const results = [];
  let total = 0;
  for (let c = 0; c < audioBuffer.numberOfChannels; c++) {
    const mean = addon().dcOffset([audioBuffer.getChannelData(c)]).dc[0];
    const abs = Math.abs(mean);
    results.push({ linear: mean, dB: abs > 0 ? 20 * Math.log10(abs) : -Infinity, percent: abs * 100, significant: abs > 0.001 });
    total += abs;
  }
  const avg = total / audioBuffer.numberOfChannels;
  return { channels: results, average: { linear: avg, dB: avg > 0 ? 20 * Math.log10(avg) : -Infinity, percent: avg * 100 },
    significant: results.some(r => r.significant) };
}
export function removeDCOffset(channel) {                                            // dc-offset.js:67 (in place, returns the offset)
  const r = addon().dcOffset([channel], true);
  channel.set(r.channels[0]);
  return r.dc[0];
}

// encodeWAV(audioBuffer, targetSampleRate, bitDepth, options) - web/ui/encoder.js:51.  bitDepth 24, else 16 (:53).
// 16-bit: options.ditherMode 'tpdf' (default) | 'noise-shaped' | 'none' (:36-40).  The reference draws its dither from
// Math.random (:10-30); here options.ditherSource selects where the same triangular density comes from:
//   'device' (default)  a seeded generator on the GPU (seed = options.ditherSeed, else one Math.random draw)
//   'math-random'       (Math.random() + Math.random()) - 1 per sample-channel on this thread, exactly the reference's draws
//   Float32Array        caller-supplied values in LSB units, one per sample-channel, interleaved order
export function encodeWAV(audioBuffer, targetSampleRate, bitDepth, options = {}) {
  // (the sample count is channel 0's length, encoder.js:63: the reference's own test buffer has no `length` field)
  const nch = audioBuffer.numberOfChannels, n = audioBuffer.getChannelData(0).length;
  const bits = bitDepth === 24 ? 24 : 16;
  const sampleRate = targetSampleRate || audioBuffer.sampleRate;
  let pcm;
  if (bits === 24) pcm = addon().encodePcm24(chansOf(audioBuffer));
  else {
    const req = options ? options.ditherMode : undefined;
    const mode = req === 'none' ? 0 : (req === 'noise-shaped' ? 2 : 1);
    let dither = null, seed = 0;
    const src = options ? options.ditherSource : undefined;
    if (mode !== 0) {
      if (src instanceof Float32Array) dither = src;
      else if (src === 'math-random') { dither = new Float32Array(n * nch); for (let j = 0; j < dither.length; j++) dither[j] = (Math.random() + Math.random()) - 1; }
      else seed = (options && Number.isFinite(options.ditherSeed)) ? (options.ditherSeed >>> 0) : Math.floor(Math.random() * 4294967296);
    }
    pcm = addon().encodePcm16(chansOf(audioBuffer), mode, dither, seed);
  }
  const bps = bits / 8;
  const out = new Uint8Array(44 + pcm.length);
  const v = new DataView(out.buffer);
  const ws = (o, s) => { for (let i = 0; i < s.length; i++) v.setUint8(o + i, s.charCodeAt(i)); };
  ws(0, 'RIFF'); v.setUint32(4, 36 + pcm.length, true); ws(8, 'WAVE'); ws(12, 'fmt '); v.setUint32(16, 16, true);
  v.setUint16(20, 1, true); v.setUint16(22, nch, true); v.setUint32(24, sampleRate, true); v.setUint32(28, sampleRate * nch * bps, true);
  v.setUint16(32, nch * bps, true); v.setUint16(34, bits, true); ws(36, 'data'); v.setUint32(40, pcm.length, true);
  out.set(pcm, 44);
  return out;
}
// encodeWAVAsync - web/ui/encoder.js:143-244.  The caller-visible contract of the reference's chunk loop is kept call for call:
// shouldCancel() is polled before every chunk of options.chunkSize frames (default 65536, at least 1024, :189) and 'Cancelled' is
// thrown when it says so, onProgress(end / numSamples) follows every chunk, the function yields to the event loop between chunks
// (never after the last), and a final onProgress(1) precedes the result.  The quantiser itself is ONE device pass over the whole
// buffer, made when the first chunk is due (the 16-bit noise shaper carries its error from frame to frame, so chunks are not
// independent work); the loop then replays the chunk boundaries.  A cancel in mid-file discards a result that already exists.
export async function encodeWAVAsync(audioBuffer, targetSampleRate, bitDepth, options = {}) {
  const { onProgress = null, shouldCancel = null, chunkSize = 65536 } = options || {};
  const numSamples = audioBuffer.getChannelData(0).length;
  const safeChunkSize = Math.max(1024, Number(chunkSize) || 65536);
  let wav = null;
  for (let i = 0; i < numSamples; i += safeChunkSize) {
    if (shouldCancel && shouldCancel()) throw new Error('Cancelled');
    if (wav === null) wav = encodeWAV(audioBuffer, targetSampleRate, bitDepth, options || {});
    const end = Math.min(i + safeChunkSize, numSamples);
    if (onProgress) onProgress(end / numSamples);
    if (end < numSamples) await new Promise((resolve) => setTimeout(resolve, 0));
  }
  if (wav === null) wav = encodeWAV(audioBuffer, targetSampleRate, bitDepth, options || {});      // an empty buffer: header only
  if (onProgress) onProgress(1);
  return wav;
}
export function stftDb(audioBuffer, nfft = 4096, hop = 1024) {                       // spectrogram.js:113-143, batched
  return addon().stftDb(chansOf(audioBuffer), audioBuffer.sampleRate, nfft, hop);
}

// A batch of tracks of one shape through the full chain (what renderFullChain, worker-interface.js:259-277, does one track at a
// time): options.devices = HIP device indices; entry i renders the contiguous track range [ceil(i T / D), ceil((i + 1) T / D)) on
// its device from its own host thread (no collective - tracks are independent); without it the batch runs on the current device.
// Resolves to one {audioBuffer, lufs, measuredLufs, truePeakDb, inputLufs, gainDb} per track, in track order.
export async function renderBatch(audioBuffers, settings, mode = 'export', options = {}) {
  if (!Array.isArray(audioBuffers) || audioBuffers.length === 0) throw new Error('renderBatch: audioBuffers must be a non-empty array');
  const sr = audioBuffers[0].sampleRate;
  if (audioBuffers.some((ab) => ab.sampleRate !== sr)) throw new Error('renderBatch: tracks of a batch share one sample rate');
  const res = await addon().renderBatch(audioBuffers.map(chansOf), sr, settings, mode, options.devices);
  return res.map((r) => ({ audioBuffer: makeAB(r.channels, sr), lufs: r.lufs, measuredLufs: r.measuredLufs, truePeakDb: r.truePeakDb,
    inputLufs: r.inputLufs, gainDb: r.gainDb }));
}

// ---- web/workers/dsp-worker.js as a Worker-shaped object ---------------------------------------------
// `new WamWorker()` stands where the reference does `new Worker(new URL('./dsp-worker.js', ...))` (worker-interface.js:29-32):
// same message protocol (dsp-worker.js:1-9) - request {type, id, data}, reply {id, success, result} or
// {id, success: false, error}, progress {id, type: 'PROGRESS', progress, status} - one request at a time, FIFO.
// Types on the hot path are served (MEASURE_LUFS, FIND_TRUE_PEAK, NORMALIZE, APPLY_LIMITER, RENDER_FULL_CHAIN,
// dsp-worker.js:1355-1468); any other type is answered like the worker's default branch (:1670).
export class WamWorker {
  constructor(device = 0) { addon().init(device); this.onmessage = null; this.onerror = null; this._tail = Promise.resolve(); this._dead = false; }
  _emit(data) { if (!this._dead && typeof this.onmessage === 'function') this.onmessage({ data }); }
  postMessage(msg /*, transferables */) {
    const { type, id } = msg;
    // postMessage clones (or takes over) the message when it is posted, not when the worker gets to it: snapshot the channels now
    const data = (msg.data && Array.isArray(msg.data.channels)) ? Object.assign({}, msg.data, { channels: msg.data.channels.map((c) => c.slice()) }) : msg.data;
    const progress = (p, status) => this._emit({ id, type: 'PROGRESS', progress: p, status });
    this._tail = this._tail.then(async () => {
      if (this._dead) return;
      try {
        let result;
        switch (type) {
          case 'MEASURE_LUFS': result = { lufs: (await addon().meter(data.channels, data.sampleRate, -14)).lufs }; break;
          case 'FIND_TRUE_PEAK': result = { peakDB: (await addon().meter(data.channels, 48000, -14)).peakDB }; break;
          // (the addon's own closing 1.0 'Complete' is dropped: the handler sends it once, below, like dsp-worker.js:1673)
          case 'NORMALIZE':                                       // targetLUFS / ceilingDB default like the worker function's parameters (:457)
            result = await addon().normalize(data.channels, data.sampleRate, data.targetLUFS === undefined ? -14 : data.targetLUFS,
              data.ceilingDB === undefined ? -1 : data.ceilingDB, (p, s) => { if (p < 1) progress(p, s); });
            break;
          case 'APPLY_LIMITER':
            result = { channels: await addon().stage('limiter', data.channels, data.sampleRate,
              [data.ceilingLinear || 0.891, data.lookaheadMs || 3, data.releaseMs || 100, data.kneeDB || 3, data.preserveTransients !== false ? 1 : 0],
              (p, s) => { if (p < 1) progress(p, s); }) };
            break;
          case 'RENDER_FULL_CHAIN': {
            const r = await addon().renderChain(data.channels, data.sampleRate, data.settings, data.mode, (p, s) => { if (p < 1) progress(p, s); });
            result = { channels: r.channels, lufs: r.lufs, measuredLufs: r.measuredLufs };
            break;
          }
          default: throw new Error(`Unknown message type: ${type}`);
        }
        await new Promise((res) => setTimeout(res, 0));          // progress calls queued by the addon drain before the reply
        progress(1.0, 'Complete');
        this._emit({ id, success: true, result });
      } catch (error) {
        this._emit({ id, success: false, error: error.message });
      }
    });
  }
  terminate() { this._dead = true; }
}

// ---- web/workers/worker-interface.js ------------------------------------------------------
export class DSPWorkerInterface {
  constructor() { this.worker = null; this.pending = new Map(); this.nextId = 0; this.isReady = false; this._tail = Promise.resolve(); }
  async init(device = 0) {                                                           // worker-interface.js:23
    if (this.worker) return;
    addon().init(device);
    this.worker = addon();
    this.isReady = true;
  }
  // one request at a time, FIFO, like the single worker thread (worker-interface.js:93-109)
  _send(run) {
    if (!this.worker) return Promise.reject(new Error('Worker not initialized. Call init() first.'));
    return new Promise((resolve, reject) => {
      const id = this.nextId++;
      this.pending.set(id, { resolve, reject });
      const settle = (fn) => (v) => { if (this.pending.has(id)) { this.pending.delete(id); fn(v); } };
      this._tail = this._tail.then(() => this.pending.has(id) ? run().then(settle(resolve), settle(reject)) : undefined);
    });
  }
  async measureLUFS(audioBuffer) {                                                   // :162
    const ch = snapshotOf(audioBuffer), sr = audioBuffer.sampleRate;
    const r = await this._send(() => addon().meter(ch, sr, -14));
    return { lufs: r.lufs };
  }
  async findTruePeak(audioBuffer) {                                                  // :178
    const ch = snapshotOf(audioBuffer);
    const r = await this._send(() => addon().meter(ch, 48000, -14));      // rate-independent, dsp-worker.js:1361-1363
    return { peakDB: r.peakDB };
  }
  async normalize(audioBuffer, targetLUFS = -14, ceilingDB = -1, onProgress = null) {   // :197
    const ch = snapshotOf(audioBuffer), sr = audioBuffer.sampleRate;
    const r = await this._send(() => addon().normalize(ch, sr, targetLUFS, ceilingDB, onProgress || undefined));
    return { audioBuffer: makeAB(r.channels, sr), currentLUFS: r.currentLUFS, finalLUFS: r.finalLUFS, peakDB: r.peakDB,
      gainApplied: r.gainApplied, limiterApplied: r.limiterApplied };
  }
  async applyLimiter(audioBuffer, options = {}, onProgress = null) {                 // :227, defaults dsp-worker.js:1385-1389
    const ch = snapshotOf(audioBuffer), sr = audioBuffer.sampleRate;
    const p = [options.ceilingLinear || 0.891, options.lookaheadMs || 3, options.releaseMs || 100, options.kneeDB || 3,
      options.preserveTransients !== false ? 1 : 0];
    const out = await this._send(() => addon().stage('limiter', ch, sr, p, onProgress || undefined));
    return { audioBuffer: makeAB(out, sr) };
  }
  async renderFullChain(audioBuffer, settings, mode = 'preview', onProgress = null) {  // :259
    const ch = snapshotOf(audioBuffer), sr = audioBuffer.sampleRate;
    const r = await this._send(() => addon().renderChain(ch, sr, settings, mode, onProgress || undefined));
    return { audioBuffer: makeAB(r.channels, sr), lufs: r.lufs, measuredLufs: r.measuredLufs };
  }
  terminate() {                                                                      // :282
    if (this.worker) {
      this.worker = null; this.isReady = false;
      for (const [, p] of this.pending) p.reject(new Error('Worker terminated'));
      this.pending.clear();
    }
  }
}
let sharedInstance = null;
export function getDSPWorker() { if (!sharedInstance) sharedInstance = new DSPWorkerInterface(); return sharedInstance; }
export async function initDSPWorker() { const w = getDSPWorker(); await w.init(); return w; }
