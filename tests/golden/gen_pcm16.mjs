// gen_pcm16.mjs - golden vectors for the 16-bit branches of encodeWAV (web/ui/encoder.js:97-115), produced by running the
// reference's own encoder under Node with Math.random replaced by a seeded generator, so that the dither it draws
// (encoder.js:10-30: a 256-entry buffer of (random + random) - 1, refilled when exhausted, consumed once per
// sample-channel in interleaved order) is known.  Writes tests/golden/data/pcm16.json + .f32/.u8 files.
//   node tests/golden/gen_pcm16.mjs            (needs /root/reference; nothing of it is copied into the repo)
import fs from 'fs';
import os from 'os';
import path from 'path';
import { fileURLToPath, pathToFileURL } from 'url';
import { synthTrack } from './synth.mjs';

const here = path.dirname(fileURLToPath(import.meta.url));
const REF = process.env.WAM_REF || '/root/reference';
const OUT = path.join(here, 'data');

// Node 12 has no optional chaining: lower the single occurrence (encoder.js:56) in a scratch copy
const src = fs.readFileSync(path.join(REF, 'web/ui/encoder.js'), 'utf8').replace('options?.ditherMode', '(options == null ? undefined : options.ditherMode)');
const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'wam-pcm16-'));
fs.writeFileSync(path.join(dir, 'package.json'), '{"type":"module"}');

class AB {
  constructor({ numberOfChannels, length, sampleRate }) { this.numberOfChannels = numberOfChannels; this.length = length; this.sampleRate = sampleRate; this._c = []; for (let i = 0; i < numberOfChannels; i++) this._c.push(new Float32Array(length)); }
  getChannelData(c) { return this._c[c]; }
}
function toAB(chans, sr) { const ab = new AB({ numberOfChannels: chans.length, length: chans[0].length, sampleRate: sr }); chans.forEach((c, i) => ab._c[i].set(c)); return ab; }
function saveF32(name, a) { fs.writeFileSync(path.join(OUT, name + '.f32'), Buffer.from(a.buffer, a.byteOffset, a.byteLength)); return name + '.f32'; }
function saveU8(name, a) { fs.writeFileSync(path.join(OUT, name + '.u8'), Buffer.from(a)); return name + '.u8'; }

// seeded stand-in for Math.random (xorshift32 -> [0, 1))
function makeRandom(seed) { let s = seed >>> 0; return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; }; }

async function run(mode, seed, chans, sr, tag) {
  // a fresh module instance per run: the dither buffer is module state and is filled at import time
  const file = path.join(dir, `encoder_${tag}.js`);
  fs.writeFileSync(file, src);
  const rnd = makeRandom(seed);
  const draws = [];
  const orig = Math.random;
  Math.random = () => { const v = rnd(); draws.push(v); return v; };
  let wav;
  try {
    const enc = await import(pathToFileURL(file).href);
    wav = enc.encodeWAV(toAB(chans, sr), sr, 16, { ditherMode: mode });
  } finally { Math.random = orig; }
  // the dither values in the order the encoder consumes them
  const dither = new Float32Array(draws.length >> 1);
  for (let j = 0; j < dither.length; j++) dither[j] = (draws[2 * j] + draws[2 * j + 1]) - 1;
  return { wav, dither };
}

async function main() {
const cases = {};
const sr = 48000;
{
  const n = 6000;
  const hot = synthTrack(9, 2, n, sr, { amp: 2.2, clickAmp: 0.9 });         // clips: exercises the clamp at +-1 and at -32768 / 32767
  const edge = [new Float32Array(n), new Float32Array(n)];
  for (let i = 0; i < n; i++) { edge[0][i] = -1.0005 + 2.001 * i / (n - 1); edge[1][i] = ((i % 64) - 31.5) / 32767 * (i % 3 === 0 ? 1 : 0.5); }
  const mono = [synthTrack(10, 1, n, sr, { amp: 0.01 })[0]];                // quiet: dither dominates
  const inputs = { hot, edge, mono };
  for (const [iname, chans] of Object.entries(inputs)) {
    const ins = chans.map((c, i) => saveF32(`pcm16_${iname}.ch${i}`, c));
    for (const mode of ['none', 'tpdf', 'noise-shaped']) {
      const tag = `${iname}_${mode.replace('-', '_')}`;
      const { wav, dither } = await run(mode, 12345 + Object.keys(cases).length, chans, sr, tag);
      cases[tag] = { input: ins, nch: chans.length, n, sr, mode, header_hex: Buffer.from(wav.slice(0, 44)).toString('hex'),
        out: saveU8(`pcm16_${tag}`, wav.slice(44)), dither: saveF32(`pcm16_${tag}.dither`, dither.subarray(0, Math.min(dither.length, n * chans.length + 256))) };
    }
  }
}
fs.writeFileSync(path.join(OUT, 'pcm16.json'), JSON.stringify({ generator: 'tests/golden/gen_pcm16.mjs', node: process.version, cases }, null, 1));
try { fs.rmdirSync(dir, { recursive: true }); } catch (e) { /* ignore */ }
console.log('wrote', Object.keys(cases).length, 'cases');
}
main().catch((e) => { console.error(e); process.exit(1); });
