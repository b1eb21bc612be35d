"""De-harsh synthesis: time per 256 tracks against the segment length (frames per wavefront segment)."""
import sys
sys.path.insert(0, "web-audio-mastering_amd/python"); sys.path.insert(0, ".")
import ctypes as C, torch, wam, bench
sr, n, T = 48000, 48000 * 180, 256
x = torch.empty((T, 2, n), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
bench.synth_dev(x, T, 0, bench.SYNTH, amp_step=0.3 / T)
s = dict(bench.C2_SETTINGS)
wam.init(0)
for seg in (0, 264, 527, 1055, 2110, 4220, 8440):
    wam.tune("dh_seg_frames", seg)
    wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, 2, n, sr, s, "export")
    wam.lib().wam_profile_enable(1); bench.profile_read()
    wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, 2, n, sr, s, "export")
    prof = bench.profile_read(); wam.lib().wam_profile_enable(0)
    print(f"dh_seg_frames {seg}: k_dh_apply {prof['k_dh_apply'][1] / prof['k_dh_apply'][0]:.2f} ms", flush=True)
