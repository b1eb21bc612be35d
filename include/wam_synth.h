/*
 * wam_synth.h - measurement utilities of libwam_hip.so.  None of these has a
 * counterpart in the reference: they exist so that bench.py and the full-size
 * GPU tests can create multi-GB synthetic inputs on the device and read
 * per-kernel timings.  The generator ("synth v1") is bit-identical to
 * tests/golden/synth.mjs (the inputs of the golden vectors) and oracle/wam_synth.c.
 */
#ifndef WAM_SYNTH_H
#define WAM_SYNTH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct wam_synth_opts {
  double amp, dc, click_amp;
  int64_t quiet_from, quiet_to; double quiet_gain;
  int64_t quiet2_from, quiet2_to; double quiet2_gain;
} wam_synth_opts;

/* out_dev: device [ntracks][nch][n] float32.  Track t uses seed idx0 + t and
 * amplitude amp + t * amp_step. */
int wam_synth_dev(float* out_dev, int ntracks, int nch, int64_t n, double sample_rate, int idx0,
                  const wam_synth_opts* opts, double amp_step, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream, around every
 * kernel the library launches while enabled.  wam_profile_read() synchronises,
 * writes one line per kernel name "name count total_ms\n" into buf and clears
 * the log.  Returns the number of bytes written (excluding the NUL). */
int wam_profile_enable(int on);
int wam_profile_read(char* buf, size_t cap);

/* Time-segmentation knobs ("Segments"): "dh_seg_frames", "dh_warm_frames", "env_seg_tiles",
 * "sc_warm_tiles", "lim_warm_tiles"; "reset" restores the defaults.  Process-wide; meant for tests that make short
 * fixtures cross segment seams, and for tuning runs.
 * "dh_exact" (default 1) selects the de-harsh kernels: 1 = every Float32Array store of the reference reproduced (double
 * transform, sequential f32 band sums); 0 = float32 kernels, 1.9x faster, ~1e-7 on the stage's output, which the soft
 * clipper can amplify past the 1e-5 PCM bar on hard-driven masters (deviations). */
int wam_tune(const char* key, int64_t value);

/* The library keeps the device memory of its intermediates (per device, grow-only, reused by later calls) because a
 * 256-track render group needs tens of GB and allocating that per call costs more than the kernels.  This call returns
 * all idle blocks to the driver. */
int wam_trim_workspace(void);

/* Policy-iteration statistics, all zero unless the library was built with -DWAM_COUNT_ITERS (a measurement build needs
 * -fgpu-rdc): out[2s] = solves, out[2s+1] = iterations; s = 0 follower, 1 follower pair, 2 release smoother. */
int wam_debug_counters(uint64_t out[8], int reset);

/* Shader clock in MHz under sustained f64 load: every CU runs independent v_fma_f64 chains for about `millis` ms and the shader
 * clock counter is read against the constant-rate counter.  bench.py prices its VALU roofline with this, not with the nominal 2.4 GHz. */
int wam_probe_shader_clock(double millis, double* mhz);

/* Host-pointer staging knobs of wam_tune: "stage_pinned" (1: pinned rings + copy threads, csrc/staging.hip; 0: hipMemcpyAsync on
 * the caller's pageable memory), "stage_threads" (1..8), "stage_nt" (1: the staging copies use non-temporal stores),
 * "host_group" (tracks per staging group of wam_batch_render);
 * "stft_literal" 0 fastest kernel, 1 the literal loop nest of spectrogram.js, 2 the register-blocked workgroup kernel;
 * "exc_peaks" 0: tape warmth's window peaks by their own pass over the batch even when the exciter could leave them behind;
 * "dh_win_lds" 0: the de-harsh kernels' instantiation that reads the Hann table from global memory (both: results unchanged). */

#ifdef __cplusplus
}
#endif
#endif
