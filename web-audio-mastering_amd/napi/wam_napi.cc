// wam_napi.cc - thin N-API addon over the C ABI of libwam_hip.so (include/wam.h).
//
// This is the binding a maintainer of the reference would load from the Electron main
// process / Node instead of spawning web/workers/dsp-worker.js: every call copies the
// Float32Array channel data it is given (the reference's interface transfers copies,
// worker-interface.js:116-132), runs the HIP backend, and hands back fresh Float32Arrays.
// Blocking calls come in two flavours: `xSync(...)` and `x(...) -> Promise`
// (napi_create_async_work, so the JS thread never blocks); chain progress is delivered
// through a napi_threadsafe_function with the worker's {progress, status} pairs.
// No DSP happens here.
#define NAPI_VERSION 4
#include <node_api.h>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include "../../include/wam.h"

#define NAPI_CALL(env, call)                                                   \
  do {                                                                         \
    if ((call) != napi_ok) {                                                   \
      napi_throw_error(env, nullptr, "N-API call failed: " #call);             \
      return nullptr;                                                          \
    }                                                                          \
  } while (0)

namespace {

struct Audio {                       // owned copy of the caller's channels
  std::vector<std::vector<float>> ch;
  int64_t n = 0;
  std::vector<const float*> ptrs() const { std::vector<const float*> p; for (auto& c : ch) p.push_back(c.data()); return p; }
};

// max_ch: 2 for the rendering stages; the meters and the DC detector take any channel count, like the reference's loops
bool get_audio(napi_env env, napi_value arr, Audio* a, uint32_t max_ch = 2) {
  bool is_arr = false;
  if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr) { napi_throw_type_error(env, nullptr, "channels must be an array of Float32Array"); return false; }
  uint32_t len = 0; napi_get_array_length(env, arr, &len);
  if (len < 1 || len > max_ch) { napi_throw_range_error(env, nullptr, max_ch == 2 ? "1 or 2 channels supported" : "1 to 64 channels supported"); return false; }
  for (uint32_t i = 0; i < len; i++) {
    napi_value v; napi_get_element(env, arr, i, &v);
    bool is_ta = false; napi_is_typedarray(env, v, &is_ta);
    napi_typedarray_type tt; size_t count; void* data; napi_value ab; size_t off;
    if (!is_ta || napi_get_typedarray_info(env, v, &tt, &count, &data, &ab, &off) != napi_ok || tt != napi_float32_array) {
      napi_throw_type_error(env, nullptr, "channels must be Float32Array"); return false;
    }
    if (i == 0) a->n = (int64_t)count;
    else if ((int64_t)count != a->n) { napi_throw_range_error(env, nullptr, "channels differ in length"); return false; }
    a->ch.emplace_back((const float*)data, (const float*)data + count);
  }
  return true;
}

napi_value make_channels(napi_env env, const std::vector<std::vector<float>>& ch) {
  napi_value arr; napi_create_array_with_length(env, ch.size(), &arr);
  for (size_t i = 0; i < ch.size(); i++) {
    void* data; napi_value ab, ta;
    napi_create_arraybuffer(env, ch[i].size() * sizeof(float), &data, &ab);
    if (!ch[i].empty()) std::memcpy(data, ch[i].data(), ch[i].size() * sizeof(float));
    napi_create_typedarray(env, napi_float32_array, ch[i].size(), ab, 0, &ta);
    napi_set_element(env, arr, (uint32_t)i, ta);
  }
  return arr;
}

double get_num(napi_env env, napi_value obj, const char* key, double dflt) {
  napi_value v; bool has = false;
  if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return dflt;
  napi_get_named_property(env, obj, key, &v);
  napi_valuetype t; napi_typeof(env, v, &t);
  if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); return b ? 1 : 0; }
  if (t == napi_number) { double d; napi_get_value_double(env, v, &d); return d; }
  if (t == napi_string) { napi_value num; if (napi_coerce_to_number(env, v, &num) == napi_ok) { double d; napi_get_value_double(env, num, &d); return d; } }
  if (t == napi_undefined || t == napi_null) return dflt;
  return NAN;
}
bool truthy(double v) { return v == v && v != 0; }

// the reference's flat settings object (web/ui/controls.js:66-89) -> wam_settings
wam_settings get_settings(napi_env env, napi_value o) {
  wam_settings s; std::memset(&s, 0, sizeof(s));
  s.normalize_loudness = truthy(get_num(env, o, "normalizeLoudness", 0));
  s.target_lufs = get_num(env, o, "targetLufs", 0);
  s.true_peak_limit = truthy(get_num(env, o, "truePeakLimit", 0));
  s.true_peak_ceiling = get_num(env, o, "truePeakCeiling", 0);
  s.clean_low_end = truthy(get_num(env, o, "cleanLowEnd", 0));
  s.glue_compression = truthy(get_num(env, o, "glueCompression", 0));
  s.deharsh = truthy(get_num(env, o, "deharsh", 0));
  s.stereo_width = get_num(env, o, "stereoWidth", NAN);
  s.center_bass = truthy(get_num(env, o, "centerBass", 0));
  s.cut_mud = truthy(get_num(env, o, "cutMud", 0));
  s.add_air = truthy(get_num(env, o, "addAir", 0));
  s.tape_warmth = truthy(get_num(env, o, "tapeWarmth", 0));
  s.auto_level = truthy(get_num(env, o, "autoLevel", 0));
  s.add_punch = truthy(get_num(env, o, "addPunch", 0));
  s.input_gain = get_num(env, o, "inputGain", 0);
  s.eq_low = get_num(env, o, "eqLow", 0); s.eq_low_mid = get_num(env, o, "eqLowMid", 0); s.eq_mid = get_num(env, o, "eqMid", 0);
  s.eq_high_mid = get_num(env, o, "eqHighMid", 0); s.eq_high = get_num(env, o, "eqHigh", 0);
  return s;
}

void set_num(napi_env env, napi_value obj, const char* k, double v) { napi_value n; napi_create_double(env, v, &n); napi_set_named_property(env, obj, k, n); }

// ---- one request = one job: work() runs off-thread (or inline for *Sync), finish() builds the JS result
struct Job {
  Audio in;
  std::vector<std::vector<float>> out;
  std::function<int(Job&)> work;
  std::function<napi_value(napi_env, Job&)> finish;
  std::string err; int rc = 0;
  double d[8] = {0}; int32_t i32[4] = {0};
  std::vector<uint8_t> bytes; std::vector<float> f32a, f32b; int64_t dims[2] = {0, 0};
  wam_settings settings; int mode = 1;
  // async plumbing
  napi_async_work aw = nullptr; napi_deferred deferred = nullptr; napi_threadsafe_function tsfn = nullptr;
  void alloc_out() { out.assign(in.ch.size(), std::vector<float>((size_t)in.n)); }
  std::vector<float*> out_ptrs() { std::vector<float*> p; for (auto& c : out) p.push_back(c.data()); return p; }
  void run() { rc = work(*this); if (rc) err = wam_last_error(); }
};

napi_value throw_or_result(napi_env env, Job& j) {
  if (j.rc) { napi_throw_error(env, nullptr, j.err.c_str()); return nullptr; }
  return j.finish(env, j);
}

struct ProgressMsg { double p; std::string s; };
int progress_trampoline(double p, const char* status, void* user) {
  Job* j = (Job*)user;
  if (j->tsfn) napi_call_threadsafe_function(j->tsfn, new ProgressMsg{p, status ? status : ""}, napi_tsfn_nonblocking);
  return 0;
}
void progress_call_js(napi_env env, napi_value cb, void*, void* data) {
  std::unique_ptr<ProgressMsg> m((ProgressMsg*)data);
  if (!env || !cb) return;
  napi_value argv[2], undef; napi_create_double(env, m->p, &argv[0]); napi_create_string_utf8(env, m->s.c_str(), NAPI_AUTO_LENGTH, &argv[1]);
  napi_get_undefined(env, &undef);
  napi_call_function(env, undef, cb, 2, argv, nullptr);
}

void async_execute(napi_env, void* data) { ((Job*)data)->run(); }
void async_complete(napi_env env, napi_status, void* data) {
  Job* j = (Job*)data;
  if (j->rc) { napi_value msg, e; napi_create_string_utf8(env, j->err.c_str(), NAPI_AUTO_LENGTH, &msg); napi_create_error(env, nullptr, msg, &e); napi_reject_deferred(env, j->deferred, e); }
  else napi_resolve_deferred(env, j->deferred, j->finish(env, *j));
  if (j->tsfn) napi_release_threadsafe_function(j->tsfn, napi_tsfn_release);
  napi_delete_async_work(env, j->aw);
  delete j;
}
napi_value start_async(napi_env env, Job* j, napi_value progress_cb) {
  napi_value promise, name;
  NAPI_CALL(env, napi_create_promise(env, &j->deferred, &promise));
  napi_create_string_utf8(env, "wam", NAPI_AUTO_LENGTH, &name);
  if (progress_cb) {
    napi_valuetype t; napi_typeof(env, progress_cb, &t);
    if (t == napi_function) napi_create_threadsafe_function(env, progress_cb, nullptr, name, 0, 1, nullptr, nullptr, nullptr, progress_call_js, &j->tsfn);
  }
  NAPI_CALL(env, napi_create_async_work(env, nullptr, name, async_execute, async_complete, j, &j->aw));
  NAPI_CALL(env, napi_queue_async_work(env, j->aw));
  return promise;
}

// ---- job builders --------------------------------------------------------------------
Job* job_meter(napi_env env, napi_value* a) {       // (channels, sampleRate, fallback)
  auto j = std::make_unique<Job>();
  if (!get_audio(env, a[0], &j->in, 64)) return nullptr;
  napi_get_value_double(env, a[1], &j->d[0]);
  j->d[1] = -14; napi_valuetype t; napi_typeof(env, a[2], &t); if (t == napi_number) napi_get_value_double(env, a[2], &j->d[1]);
  j->work = [](Job& j) { auto p = j.in.ptrs(); return wam_meter(p.data(), (int)p.size(), j.in.n, j.d[0], j.d[1], &j.d[2], &j.d[3]); };
  j->finish = [](napi_env env, Job& j) { napi_value o; napi_create_object(env, &o); set_num(env, o, "lufs", j.d[2]); set_num(env, o, "peakDB", j.d[3]); return o; };
  return j.release();
}
Job* job_window_lufs(napi_env env, napi_value* a) {   // (channels, sampleRate, position, momentary) -> number
  auto j = std::make_unique<Job>();
  if (!get_audio(env, a[0], &j->in, 64)) return nullptr;
  napi_get_value_double(env, a[1], &j->d[0]);
  j->d[1] = 0; napi_valuetype t; napi_typeof(env, a[2], &t); if (t == napi_number) napi_get_value_double(env, a[2], &j->d[1]);
  bool mom = false; napi_value b; if (napi_coerce_to_bool(env, a[3], &b) == napi_ok) napi_get_value_bool(env, b, &mom);
  j->i32[0] = mom ? 1 : 0;
  j->work = [](Job& j) {
    auto p = j.in.ptrs();
    return j.i32[0] ? wam_momentary_lufs(p.data(), (int)p.size(), j.in.n, j.d[0], j.d[1], &j.d[2])
                    : wam_short_term_lufs(p.data(), (int)p.size(), j.in.n, j.d[0], j.d[1], &j.d[2]);
  };
  j->finish = [](napi_env env, Job& j) { napi_value v; napi_create_double(env, j.d[2], &v); return v; };
  return j.release();
}
Job* job_render(napi_env env, napi_value* a, bool with_progress) {   // (channels, sampleRate, settings, mode[, onProgress])
  auto j = std::make_unique<Job>();
  if (!get_audio(env, a[0], &j->in)) return nullptr;
  napi_get_value_double(env, a[1], &j->d[0]);
  j->settings = get_settings(env, a[2]);
  char mode[16] = "preview"; size_t ml = 0; napi_valuetype t; napi_typeof(env, a[3], &t);
  if (t == napi_string) napi_get_value_string_utf8(env, a[3], mode, sizeof(mode), &ml);
  j->mode = std::strcmp(mode, "export") == 0 ? WAM_MODE_EXPORT : WAM_MODE_PREVIEW;          // dsp-worker.js:1473
  j->work = [with_progress](Job& j) {
    j.alloc_out(); auto p = j.in.ptrs(); auto o = j.out_ptrs(); wam_result r;
    int rc = wam_render_chain(p.data(), (int)p.size(), j.in.n, j.d[0], &j.settings, j.mode, o.data(), &r, with_progress ? progress_trampoline : nullptr, &j);
    j.d[1] = r.lufs; j.d[2] = r.measured_lufs; j.d[3] = r.true_peak_db; j.d[4] = r.lufs_in; j.d[5] = r.gain_db; return rc;
  };
  j->finish = [](napi_env env, Job& j) {
    napi_value o; napi_create_object(env, &o);
    napi_set_named_property(env, o, "channels", make_channels(env, j.out));
    set_num(env, o, "lufs", j.d[1]); set_num(env, o, "measuredLufs", j.d[2]); set_num(env, o, "truePeakDb", j.d[3]);
    set_num(env, o, "inputLufs", j.d[4]); set_num(env, o, "gainDb", j.d[5]);
    return o;
  };
  return j.release();
}
Job* job_normalize(napi_env env, napi_value* a) {   // (channels, sampleRate, targetLUFS, ceilingDB[, onProgress])
  auto j = std::make_unique<Job>();
  if (!get_audio(env, a[0], &j->in)) return nullptr;
  napi_get_value_double(env, a[1], &j->d[0]); napi_get_value_double(env, a[2], &j->d[1]); napi_get_value_double(env, a[3], &j->d[2]);
  j->work = [](Job& j) {
    j.alloc_out(); auto p = j.in.ptrs(); auto o = j.out_ptrs(); wam_normalize_result r;
    int rc = wam_normalize(p.data(), (int)p.size(), j.in.n, j.d[0], j.d[1], j.d[2], o.data(), &r, j.tsfn ? progress_trampoline : nullptr, &j);
    j.d[3] = r.current_lufs; j.d[4] = r.final_lufs; j.d[5] = r.peak_db; j.d[6] = r.gain_applied_db; j.i32[0] = r.limiter_applied; j.i32[1] = r.measured; return rc;
  };
  j->finish = [](napi_env env, Job& j) {          // same keys as dsp-worker.js:468,507-526
    napi_value o; napi_create_object(env, &o);
    napi_set_named_property(env, o, "channels", make_channels(env, j.out));
    set_num(env, o, "currentLUFS", j.d[3]); set_num(env, o, "finalLUFS", j.d[4]); set_num(env, o, "peakDB", j.d[5]);
    if (j.i32[1]) { set_num(env, o, "gainApplied", j.d[6]); napi_value b; napi_get_boolean(env, j.i32[0] != 0, &b); napi_set_named_property(env, o, "limiterApplied", b); }
    return o;
  };
  return j.release();
}
// generic stage: (name, channels, sampleRate, params[][, onProgress]) -> channels   (only 'limiter' reports progress: APPLY_LIMITER)
Job* job_stage(napi_env env, napi_value* a) {
  auto j = std::make_unique<Job>();
  char name[32] = ""; size_t nl = 0; napi_get_value_string_utf8(env, a[0], name, sizeof(name), &nl);
  if (!get_audio(env, a[1], &j->in)) return nullptr;
  napi_get_value_double(env, a[2], &j->d[0]);
  double prm[7] = {0}; uint32_t np = 0; bool is_arr = false; napi_is_array(env, a[3], &is_arr);
  if (is_arr) { napi_get_array_length(env, a[3], &np); for (uint32_t i = 0; i < np && i < 7; i++) { napi_value v; napi_get_element(env, a[3], i, &v); napi_value num; napi_coerce_to_number(env, v, &num); napi_get_value_double(env, num, &prm[i]); } }
  for (int i = 0; i < 7; i++) j->d[1 + i] = prm[i];
  const std::string nm = name;
  j->work = [nm](Job& j) -> int {
    j.alloc_out(); auto p = j.in.ptrs(); auto o = j.out_ptrs(); const int nch = (int)p.size(); const double sr = j.d[0]; const double* q = &j.d[1];
    if (nm == "finalFilters") return wam_final_filters(p.data(), nch, j.in.n, sr, q[0] != 0, q[1] != 0, o.data());
    if (nm == "softClip") return wam_soft_clip(p.data(), nch, j.in.n, sr, q[0], q[1], q[2], q[3], o.data());
    if (nm == "limiter") return wam_limiter(p.data(), nch, j.in.n, sr, q[0], q[1], q[2], q[3], q[4] != 0, o.data(), j.tsfn ? progress_trampoline : nullptr, &j);
    if (nm == "limiterLib") return wam_limiter_lib(p.data(), nch, j.in.n, sr, q[0], q[1], q[2], q[3], q[4] != 0, o.data());
    if (nm == "glue") return wam_glue_compressor(p.data(), nch, j.in.n, sr, q[0], q[1], q[2], q[3], o.data());
    if (nm == "stereo") return wam_stereo(p.data(), j.in.n, sr, q[0], q[1] != 0, q[2], o.data());
    if (nm == "eq") return wam_parametric_eq(p.data(), nch, j.in.n, sr, q, q[5] != 0, o.data());
    if (nm == "exciter") return wam_exciter(p.data(), nch, j.in.n, sr, o.data());
    if (nm == "deharsh") return wam_deharsh_preset(p.data(), nch, j.in.n, sr, (int)q[0], o.data());
    if (nm == "tapeWarmth") return wam_tape_warmth(p.data(), nch, j.in.n, sr, o.data());
    if (nm == "multibandTransient") return wam_multiband_transient(p.data(), nch, j.in.n, sr, o.data());
    if (nm == "gain") return wam_apply_gain(p.data(), nch, j.in.n, q[0], o.data());
    if (nm == "removeDC") { double dc[2]; int rc = wam_dc_offset(p.data(), nch, j.in.n, dc, 1, o.data()); j.d[1] = dc[0]; j.d[2] = nch > 1 ? dc[1] : 0; return rc; }
    return WAM_ERR_INVALID;
  };
  j->finish = [](napi_env env, Job& j) { return make_channels(env, j.out); };
  return j.release();
}

template <Job* (*Build)(napi_env, napi_value*)>
napi_value sync_entry(napi_env env, napi_callback_info info) {
  napi_value a[6]; size_t argc = 6; napi_value undef; napi_get_undefined(env, &undef);
  for (auto& v : a) v = undef;
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  std::unique_ptr<Job> j(Build(env, a)); if (!j) return nullptr;
  j->run();
  return throw_or_result(env, *j);
}
// PROGRESS: index of the optional onProgress(progress, status) argument, -1 = the call reports none
template <Job* (*Build)(napi_env, napi_value*), int PROGRESS = -1>
napi_value async_entry(napi_env env, napi_callback_info info) {
  napi_value a[6]; size_t argc = 6; napi_value undef; napi_get_undefined(env, &undef);
  for (auto& v : a) v = undef;
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Job* j = Build(env, a); if (!j) return nullptr;
  return start_async(env, j, PROGRESS >= 0 ? a[PROGRESS] : nullptr);
}
Job* build_render_sync(napi_env e, napi_value* a) { return job_render(e, a, false); }
Job* build_render_async(napi_env e, napi_value* a) { return job_render(e, a, true); }
napi_value render_async(napi_env env, napi_callback_info info) {
  napi_value a[6]; size_t argc = 6; napi_value undef; napi_get_undefined(env, &undef);
  for (auto& v : a) v = undef;
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Job* j = build_render_async(env, a); if (!j) return nullptr;
  return start_async(env, j, a[4]);
}

napi_value encode_pcm24(napi_env env, napi_callback_info info) {     // (channels) -> Uint8Array (no header)
  napi_value a[1]; size_t argc = 1; NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Audio in; if (!get_audio(env, a[0], &in)) return nullptr;
  const size_t bytes = (size_t)3 * in.ch.size() * (size_t)in.n;
  void* data; napi_value ab, ta; NAPI_CALL(env, napi_create_arraybuffer(env, bytes, &data, &ab));
  auto p = in.ptrs();
  if (wam_encode_pcm24(p.data(), (int)p.size(), in.n, (uint8_t*)data)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta));
  return ta;
}
napi_value encode_pcm16(napi_env env, napi_callback_info info) {     // (channels, ditherMode 0|1|2, dither: Float32Array|null, seed) -> Uint8Array
  napi_value a[4]; size_t argc = 4; napi_value undef; napi_get_undefined(env, &undef); for (auto& v : a) v = undef;
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Audio in; if (!get_audio(env, a[0], &in)) return nullptr;
  int32_t mode = 1; napi_get_value_int32(env, a[1], &mode);
  const float* dither = nullptr;
  bool is_ta = false; napi_is_typedarray(env, a[2], &is_ta);
  if (is_ta) {
    napi_typedarray_type tt; size_t count; void* data; napi_value ab; size_t off;
    if (napi_get_typedarray_info(env, a[2], &tt, &count, &data, &ab, &off) != napi_ok || tt != napi_float32_array || count < in.ch.size() * (size_t)in.n) {
      napi_throw_range_error(env, nullptr, "dither must be a Float32Array of nch * n values"); return nullptr;
    }
    dither = (const float*)data;
  }
  uint32_t seed = 0; napi_get_value_uint32(env, a[3], &seed);
  const size_t bytes = (size_t)2 * in.ch.size() * (size_t)in.n;
  void* data; napi_value ab, ta; NAPI_CALL(env, napi_create_arraybuffer(env, bytes, &data, &ab));
  auto p = in.ptrs();
  if (wam_encode_pcm16(p.data(), (int)p.size(), in.n, mode, dither, seed, (uint8_t*)data)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta));
  return ta;
}
napi_value stft_db(napi_env env, napi_callback_info info) {          // (channels, sampleRate, nfft, hop) -> {frames, bins, db: Float32Array}
  napi_value a[4]; size_t argc = 4; NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Audio in; if (!get_audio(env, a[0], &in)) return nullptr;
  double sr; int32_t nfft, hop; napi_get_value_double(env, a[1], &sr); napi_get_value_int32(env, a[2], &nfft); napi_get_value_int32(env, a[3], &hop);
  const int64_t nf = wam_stft_num_frames(in.n, nfft, hop);
  void* data; napi_value ab, ta, o; NAPI_CALL(env, napi_create_arraybuffer(env, sizeof(float) * (size_t)nf * (nfft / 2), &data, &ab));
  auto p = in.ptrs();
  if (wam_stft_db(p.data(), (int)p.size(), in.n, sr, nfft, hop, (float*)data, nullptr)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  NAPI_CALL(env, napi_create_typedarray(env, napi_float32_array, (size_t)nf * (nfft / 2), ab, 0, &ta));
  napi_create_object(env, &o); set_num(env, o, "frames", (double)nf); set_num(env, o, "bins", nfft / 2); napi_set_named_property(env, o, "db", ta);
  return o;
}
napi_value dc_offset(napi_env env, napi_callback_info info) {        // (channels[, remove]) -> {dc: number[], channels?}
  napi_value a[2]; size_t argc = 2; NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Audio in; if (!get_audio(env, a[0], &in, 64)) return nullptr;
  bool remove = false; if (argc >= 2) { napi_value b; if (napi_coerce_to_bool(env, a[1], &b) == napi_ok) napi_get_value_bool(env, b, &remove); }
  std::vector<std::vector<float>> out; std::vector<float*> op;
  if (remove) { out.assign(in.ch.size(), std::vector<float>((size_t)in.n)); for (auto& c : out) op.push_back(c.data()); }
  std::vector<double> dc(in.ch.size(), 0.0); auto p = in.ptrs();
  if (wam_dc_offset(p.data(), (int)p.size(), in.n, dc.data(), remove ? 1 : 0, remove ? op.data() : nullptr)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  napi_value o, arr; napi_create_object(env, &o); napi_create_array_with_length(env, in.ch.size(), &arr);
  for (size_t c = 0; c < in.ch.size(); c++) { napi_value v; napi_create_double(env, dc[c], &v); napi_set_element(env, arr, (uint32_t)c, v); }
  napi_set_named_property(env, o, "dc", arr);
  if (remove) napi_set_named_property(env, o, "channels", make_channels(env, out));
  return o;
}
napi_value analyze_on_load(napi_env env, napi_callback_info info) {   // (channels, sampleRate, targetLufs, ceilingDb) -> {dc, dcPercent, ..., cleaned, normalized}   (app.js:823-872)
  napi_value a[4]; size_t argc = 4; NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Audio in; if (!get_audio(env, a[0], &in)) return nullptr;
  double sr = 0, target = -14, ceiling = -1;
  napi_get_value_double(env, a[1], &sr); if (argc >= 3) napi_get_value_double(env, a[2], &target); if (argc >= 4) napi_get_value_double(env, a[3], &ceiling);
  std::vector<std::vector<float>> cl(in.ch.size(), std::vector<float>((size_t)in.n)), nm(in.ch.size(), std::vector<float>((size_t)in.n));
  std::vector<float*> pc, pn; for (auto& c : cl) pc.push_back(c.data()); for (auto& c : nm) pn.push_back(c.data());
  wam_load_analysis r; auto p = in.ptrs();
  if (wam_analyze_on_load(p.data(), (int)p.size(), in.n, sr, target, ceiling, pc.data(), pn.data(), &r)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  napi_value o, arr; napi_create_object(env, &o); napi_create_array_with_length(env, in.ch.size(), &arr);
  for (size_t c = 0; c < in.ch.size(); c++) { napi_value v; napi_create_double(env, r.dc[c], &v); napi_set_element(env, arr, (uint32_t)c, v); }
  napi_set_named_property(env, o, "dc", arr);
  set_num(env, o, "dcPercent", r.dc_percent); set_num(env, o, "originalLufs", r.original_lufs); set_num(env, o, "originalTruePeak", r.original_true_peak_db);
  set_num(env, o, "gainDb", r.gain_db); set_num(env, o, "dcSignificant", r.dc_significant); set_num(env, o, "dcRemoved", r.dc_removed);
  set_num(env, o, "measured", r.measured); set_num(env, o, "limiterApplied", r.limiter_applied);
  napi_set_named_property(env, o, "cleaned", make_channels(env, cl));
  napi_set_named_property(env, o, "normalized", make_channels(env, nm));
  return o;
}
napi_value resample(napi_env env, napi_callback_info info) {         // (channels, srcRate, dstRate, quality 0|1) -> Float32Array[]   (renderer.js:172-187, unpinned)
  napi_value a[4]; size_t argc = 4; NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Audio in; if (!get_audio(env, a[0], &in)) return nullptr;
  double src = 0, dst = 0; int32_t q = 1;
  napi_get_value_double(env, a[1], &src); napi_get_value_double(env, a[2], &dst); if (argc >= 4) napi_get_value_int32(env, a[3], &q);
  const int64_t n_out = wam_resample_length(in.n, src, dst);
  std::vector<std::vector<float>> out(in.ch.size(), std::vector<float>((size_t)(n_out > 0 ? n_out : 0)));
  std::vector<float*> op; for (auto& c : out) op.push_back(c.data());
  auto p = in.ptrs();
  if (wam_resample(p.data(), (int)p.size(), in.n, src, dst, q, op.data(), n_out)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  return make_channels(env, out);
}
napi_value extract_peaks(napi_env env, napi_callback_info info) {    // (channels, numPeaks) -> Float32Array   (waveform.js:164-181, channel 0)
  napi_value a[2]; size_t argc = 2; NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Audio in; if (!get_audio(env, a[0], &in)) return nullptr;
  int32_t np = 800; if (argc >= 2) napi_get_value_int32(env, a[1], &np);
  if (np < 0) np = 0;
  void* data; napi_value ab, ta; NAPI_CALL(env, napi_create_arraybuffer(env, sizeof(float) * (size_t)np, &data, &ab));
  auto p = in.ptrs();
  if (wam_extract_peaks(p[0], in.n, np, (float*)data)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  NAPI_CALL(env, napi_create_typedarray(env, napi_float32_array, (size_t)np, ab, 0, &ta));
  return ta;
}
napi_value lufs_series(napi_env env, napi_callback_info info) {      // (channels, sampleRate, positions: Float64Array, momentary) -> Float64Array
  napi_value a[4]; size_t argc = 4; NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Audio in; if (!get_audio(env, a[0], &in, 64)) return nullptr;
  double sr = 0; napi_get_value_double(env, a[1], &sr);
  napi_typedarray_type tt; size_t count = 0; void* pos = nullptr; napi_value pab; size_t off;
  if (argc < 3 || napi_get_typedarray_info(env, a[2], &tt, &count, &pos, &pab, &off) != napi_ok || tt != napi_float64_array) {
    napi_throw_type_error(env, nullptr, "positions must be a Float64Array"); return nullptr;
  }
  bool mom = false; if (argc >= 4) { napi_value b; if (napi_coerce_to_bool(env, a[3], &b) == napi_ok) napi_get_value_bool(env, b, &mom); }
  void* data; napi_value ab, ta; NAPI_CALL(env, napi_create_arraybuffer(env, sizeof(double) * count, &data, &ab));
  auto p = in.ptrs();
  if (wam_lufs_series(p.data(), (int)p.size(), in.n, sr, (const double*)pos, (int)count, mom ? 1 : 0, (double*)data)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  NAPI_CALL(env, napi_create_typedarray(env, napi_float64_array, count, ab, 0, &ta));
  return ta;
}
// renderBatch(tracks: Float32Array[][], sampleRate, settings, mode, devices?: number[]) -> Promise<{channels, lufs, measuredLufs, truePeakDb, inputLufs, gainDb}[]>
// wam_batch_render_multi (devices given: contiguous track ranges, one host thread + streams per entry) or wam_batch_render (current device).
struct BatchJob {
  std::vector<Audio> tracks;
  std::vector<std::vector<std::vector<float>>> out;
  std::vector<wam_result> res;
  std::vector<int> devices;
  wam_settings settings; int mode = 1; double sr = 0;
  int rc = 0; std::string err;
  napi_async_work aw = nullptr; napi_deferred deferred = nullptr;
};
void batch_execute(napi_env, void* data) {
  BatchJob* j = (BatchJob*)data;
  const int nt = (int)j->tracks.size(), nch = (int)j->tracks[0].ch.size(); const int64_t n = j->tracks[0].n;
  std::vector<const float*> in; std::vector<float*> out;
  j->out.assign(nt, std::vector<std::vector<float>>(nch, std::vector<float>((size_t)n)));
  for (int t = 0; t < nt; t++) for (int c = 0; c < nch; c++) { in.push_back(j->tracks[t].ch[c].data()); out.push_back(j->out[t][c].data()); }
  j->res.resize(nt);
  j->rc = j->devices.empty() ? wam_batch_render(in.data(), nt, nch, n, j->sr, &j->settings, j->mode, out.data(), j->res.data())
                             : wam_batch_render_multi(in.data(), nt, nch, n, j->sr, &j->settings, j->mode, out.data(), j->res.data(), j->devices.data(), (int)j->devices.size());
  if (j->rc) j->err = wam_last_error();
}
void batch_complete(napi_env env, napi_status, void* data) {
  BatchJob* j = (BatchJob*)data;
  if (j->rc) { napi_value msg, e; napi_create_string_utf8(env, j->err.c_str(), NAPI_AUTO_LENGTH, &msg); napi_create_error(env, nullptr, msg, &e); napi_reject_deferred(env, j->deferred, e); }
  else {
    napi_value arr; napi_create_array_with_length(env, j->out.size(), &arr);
    for (size_t t = 0; t < j->out.size(); t++) {
      napi_value o; napi_create_object(env, &o);
      napi_set_named_property(env, o, "channels", make_channels(env, j->out[t]));
      const wam_result& r = j->res[t];
      set_num(env, o, "lufs", r.lufs); set_num(env, o, "measuredLufs", r.measured_lufs); set_num(env, o, "truePeakDb", r.true_peak_db);
      set_num(env, o, "inputLufs", r.lufs_in); set_num(env, o, "gainDb", r.gain_db);
      napi_set_element(env, arr, (uint32_t)t, o);
    }
    napi_resolve_deferred(env, j->deferred, arr);
  }
  napi_delete_async_work(env, j->aw);
  delete j;
}
napi_value render_batch(napi_env env, napi_callback_info info) {
  napi_value a[5]; size_t argc = 5; napi_value undef; napi_get_undefined(env, &undef); for (auto& v : a) v = undef;
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  bool is_arr = false; uint32_t nt = 0;
  if (napi_is_array(env, a[0], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, a[0], &nt) != napi_ok || nt < 1) {
    napi_throw_type_error(env, nullptr, "tracks must be a non-empty array of channel arrays"); return nullptr;
  }
  auto j = std::make_unique<BatchJob>();
  j->tracks.resize(nt);
  for (uint32_t t = 0; t < nt; t++) {
    napi_value tr; napi_get_element(env, a[0], t, &tr);
    if (!get_audio(env, tr, &j->tracks[t])) return nullptr;
    if (j->tracks[t].ch.size() != j->tracks[0].ch.size() || j->tracks[t].n != j->tracks[0].n) { napi_throw_range_error(env, nullptr, "tracks of a batch share one shape"); return nullptr; }
  }
  napi_get_value_double(env, a[1], &j->sr);
  j->settings = get_settings(env, a[2]);
  char mode[16] = "preview"; size_t ml = 0; napi_valuetype t; napi_typeof(env, a[3], &t);
  if (t == napi_string) napi_get_value_string_utf8(env, a[3], mode, sizeof(mode), &ml);
  j->mode = std::strcmp(mode, "export") == 0 ? WAM_MODE_EXPORT : WAM_MODE_PREVIEW;
  bool dev_arr = false; napi_is_array(env, a[4], &dev_arr);
  if (dev_arr) { uint32_t nd = 0; napi_get_array_length(env, a[4], &nd); for (uint32_t i = 0; i < nd; i++) { napi_value v; napi_get_element(env, a[4], i, &v); int32_t d = 0; napi_get_value_int32(env, v, &d); j->devices.push_back(d); } }
  napi_value promise, name;
  NAPI_CALL(env, napi_create_promise(env, &j->deferred, &promise));
  napi_create_string_utf8(env, "wam-batch", NAPI_AUTO_LENGTH, &name);
  BatchJob* raw = j.release();
  NAPI_CALL(env, napi_create_async_work(env, nullptr, name, batch_execute, batch_complete, raw, &raw->aw));
  NAPI_CALL(env, napi_queue_async_work(env, raw->aw));
  return promise;
}

napi_value init_dev(napi_env env, napi_callback_info info) {
  napi_value a[1]; size_t argc = 1; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
  int32_t dev = 0; if (argc >= 1) napi_get_value_int32(env, a[0], &dev);
  if (wam_init(dev)) { napi_throw_error(env, nullptr, wam_last_error()); return nullptr; }
  napi_value u; napi_get_undefined(env, &u); return u;
}
napi_value device_count(napi_env env, napi_callback_info) { int n = 0; wam_device_count(&n); napi_value v; napi_create_int32(env, n, &v); return v; }

napi_value Init(napi_env env, napi_value exports) {
  if (wam_abi_version() != WAM_ABI_VERSION) {          // a libwam_hip.so with other signatures than this addon was compiled against
    napi_throw_error(env, nullptr, "wam: libwam_hip.so ABI version differs from the one this addon was built against (include/wam.h)");
    return exports;
  }
  const napi_property_descriptor props[] = {
    {"init", nullptr, init_dev, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"deviceCount", nullptr, device_count, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"meterSync", nullptr, sync_entry<job_meter>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"windowLufsSync", nullptr, sync_entry<job_window_lufs>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"meter", nullptr, async_entry<job_meter>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"renderChainSync", nullptr, sync_entry<build_render_sync>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"renderChain", nullptr, render_async, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"renderBatch", nullptr, render_batch, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"normalizeSync", nullptr, sync_entry<job_normalize>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"normalize", nullptr, async_entry<job_normalize, 4>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"stageSync", nullptr, sync_entry<job_stage>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"stage", nullptr, async_entry<job_stage, 4>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"encodePcm24", nullptr, encode_pcm24, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"encodePcm16", nullptr, encode_pcm16, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"stftDb", nullptr, stft_db, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"dcOffset", nullptr, dc_offset, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"extractPeaks", nullptr, extract_peaks, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"resample", nullptr, resample, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"analyzeOnLoad", nullptr, analyze_on_load, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    {"lufsSeries", nullptr, lufs_series, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
