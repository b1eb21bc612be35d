#!/usr/bin/env python3
"""bench.py - realtime factor of the MI355X mastering backend (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic 3-min stereo 48 kHz
tracks already resident in HBM: the full export chain + LUFS/true-peak metering
(`--workload chain`, BASELINE.json configs[2] sharded track-wise: `--tracks-per-gpu`
tracks on each GPU; the default, 1,024, is the whole configs[2] batch on one GPU) or metering only
(`--workload meter`, configs[0]).  Tracks are independent, so ranks never exchange
audio: weak scaling, no data-path collective; torch.distributed is
used only for the barrier and the max-over-ranks reduction of the step time.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed
on the launch stream through the library's profiling hooks) and `cpu_baseline`
(the CPU oracle timed on a bounded sample, rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "web-audio-mastering_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402  (first: libwam_hip.so must bind to the same HIP runtime as torch)
import torch.distributed as dist  # noqa: E402
import wam  # noqa: E402
import wam_dist  # noqa: E402

SR = 48000
SECONDS = 180
NCH = 2
N = SR * SECONDS

# Settings of configs[1]/[2] (SURVEY.md 8d "C2"): UI defaults (web/index.html:178-295)
# + glue compression + the 'aifix' EQ preset (web/lib/presets/eq.js:21).
C2_SETTINGS = dict(wam.DEFAULT_SETTINGS, glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2)
SYNTH = dict(amp=1.6, dc=0.0, clickAmp=0.6)

# algorithmic bytes per sample-channel of each kernel (DESIGN.md "Kernels")
KERNEL_BYTES = {
    "k_meter": 4,          # read x
    "k_gain_meter": 8,     # read x, write x*G
    "k_dh_bands": 4,       # read x (the 4x frame overlap is served by L2)
    "k_dh_apply": 8,       # read x, write y
    "k_dh_bands_x": 4,     # the default (store-for-store) de-harsh kernels: same traffic as the float32 pair
    "k_dh_apply_x": 8,
    "k_exciter": 8,
    "k_win_peak": 4,
    "k_warm": 8,
    "k_punch": 8,
    "k_front": 8,
    "k_sc_env": 6,         # read x (4), write the shared gain envelope (4 per frame = 2 per sample-channel)
    "k_sc_apply": 10,      # read x + envelope, write y
    "k_lim_tm": 6,         # read x, write transient map
    "k_lim_gain": 8,       # read x + transient map, write gain envelope
    "k_lim_apply": 10,     # read x + envelope, write y
}


class SynthOpts(C.Structure):
    _fields_ = [("amp", C.c_double), ("dc", C.c_double), ("click_amp", C.c_double),
                ("quiet_from", C.c_int64), ("quiet_to", C.c_int64), ("quiet_gain", C.c_double),
                ("quiet2_from", C.c_int64), ("quiet2_to", C.c_int64), ("quiet2_gain", C.c_double)]


def synth_dev(t, ntracks, idx0, opts, amp_step=0.0, sr=SR):
    o = SynthOpts(opts.get("amp", 1.0), opts.get("dc", 0.0), opts.get("clickAmp", 0.0), -1, -1, 0.0, -1, -1, 0.0)
    rc = wam.lib().wam_synth_dev(C.c_void_p(t.data_ptr()), C.c_int(ntracks), C.c_int(NCH), C.c_int64(t.shape[-1]), C.c_double(sr),
                                 C.c_int(idx0), C.byref(o), C.c_double(amp_step), C.c_void_p(0))
    if rc:
        raise wam.WamError(wam.lib().wam_last_error().decode())


def profile_read():
    buf = C.create_string_buffer(1 << 16)
    wam.lib().wam_profile_read(buf, C.c_size_t(len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


GROUP = 256      # tracks per render group inside the library (csrc/chain_api.hip): one launch of a chain kernel covers a group


def pmc_traffic(kernel, tracks_per_launch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01_pmc_summary.csv: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench, FETCH_SIZE doubled per the gfx950 correction)."""
    import csv
    for name in ("r01_pmc_summary.csv", "r01_pmc_meter_summary.csv"):      # chain passes, meter-workload passes
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            rows = list(csv.DictReader(f))
        meta = [r for r in rows if r["kernel"] == "#tracks_per_launch"]
        ref_tracks = float(meta[0]["launches"]) if meta else None
        for r in rows:
            if r["kernel"] == kernel:
                try:
                    b = (float(r["fetch_GB_per_launch_x2"]) + float(r["write_GB_per_launch"])) * 1e9
                except ValueError:
                    return None
                if ref_tracks and ref_tracks != tracks_per_launch:
                    b *= tracks_per_launch / ref_tracks          # traffic is proportional to the tracks a launch covers
                return b
    return None


def cpu_baseline(workload, sr=SR):
    """Oracle (CPU restatement of the reference, kind 'port') on a bounded sample of the same workload: whole tracks of the
    workload's shape (3-min stereo; a 60 s excerpt for the long-form case), one after the other on one core, until about
    10 s of CPU time have been spent."""
    import oracle_lib as O
    chain = workload in ("chain", "single", "longform")
    secs = 60 if workload == "longform" else 180
    budget, done, dt = 10.0, 0, 0.0
    while dt < budget and done < 64:
        x = O.synth(done, NCH, secs * sr, sr, SYNTH)
        t0 = time.perf_counter()
        if chain:
            O.render_chain(x, sr, C2_SETTINGS, "export")
        elif workload == "stft":
            O.stft_db(x, sr, 4096, 1024)
        else:
            O.measure_lufs(x, sr)
            O.true_peak_db(x)
        dt += time.perf_counter() - t0
        done += 1
    return {"value": done * secs / dt, "unit": "x realtime (audio-s/s)", "cores": 1, "kind": "port",
            "sample": f"{done} track(s) x {secs} s stereo {sr // 1000} kHz, same synthetic generator and settings, oracle/wam_oracle.c "
                      f"(C restatement of the reference JS, one thread), {dt:.1f} s of CPU time"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tracks-per-gpu", type=int, default=None, help="default 1024 (chain, meter, stft), 1 (single, longform)")
    ap.add_argument("--workload", choices=["chain", "meter", "single", "longform", "stft"], default="chain",
                    help="chain: configs[2] (default, the bench line); meter: configs[0] batched; single: configs[1], one 3-min track; "
                         "longform: configs[3], one 60-min 96 kHz track; stft: configs[4], 4096-point STFT hop 1024 over the corpus")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="wam_tune knob for an experiment (e.g. dh_exact=1); recorded in config")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # WAM_BENCH_REHEARSAL=1: run the N-rank code path on a box with ONE GPU (all ranks on cuda:0, gloo for the three collectives);
    # a correctness rehearsal of the sharding / gathering plumbing only - its numbers mean nothing.
    rehearsal = os.environ.get("WAM_BENCH_REHEARSAL") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if (world > 1 and not rehearsal) else 0)
    comm = torch.device("cpu") if rehearsal else dev            # where the collectives' tensors live
    wam.init(dev.index)
    for kv in args.tune:
        wam.tune(kv.split("=")[0], int(kv.split("=")[1]))

    global SR, SECONDS, N
    wl = args.workload
    chain_like = wl in ("chain", "single", "longform")
    if wl == "longform":
        SR, SECONDS = 96000, 3600
    N = SR * SECONDS
    T = args.tracks_per_gpu if args.tracks_per_gpu else (1 if wl in ("single", "longform") else 1024)
    x = torch.empty((T, NCH, N), dtype=torch.float32, device=dev)
    synth_dev(x, T, rank * T, SYNTH, amp_step=0.3 / max(T, 1), sr=SR)      # distinct seed per track, levels 1.6 .. 1.9
    y = torch.empty_like(x) if chain_like else None
    NFFT, HOP = 4096, 1024
    if wl == "stft":
        nframes = (N - NFFT) // HOP + 1
        spec = torch.empty((T, nframes, NFFT // 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        if chain_like:
            return wam.batch_render_dev(x.data_ptr(), y.data_ptr(), T, NCH, N, SR, C2_SETTINGS, "export")
        if wl == "stft":
            wam._check(wam.lib().wam_batch_stft_db_dev(C.c_void_p(x.data_ptr()), C.c_int(T), C.c_int(NCH), C.c_int64(N), C.c_double(SR), C.c_int(NFFT),
                                                       C.c_int(HOP), C.c_void_p(spec.data_ptr()), C.c_void_p(0), C.c_void_p(0)))
            return None
        return wam.batch_meter_dev(x.data_ptr(), T, NCH, N, SR)

    def barrier():
        torch.cuda.synchronize(dev)
        wam_dist.barrier(comm)
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    wam.lib().wam_profile_enable(1)
    profile_read()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = profile_read()
    wam.lib().wam_profile_enable(0)
    dt = wam_dist.max_over_ranks(dt, comm)
    # per-track scalars of every shard, in global track order (the only data that leaves a rank)
    if chain_like:
        local = [[r.lufs, r.true_peak_db, r.lufs_in, r.gain_db] for r in res]
    elif wl == "stft":
        local = [[float(spec[t].max()), float(spec[t, :, 1:].float().mean())] for t in range(min(T, 4))] + [[0.0, 0.0]] * max(0, T - 4)
    else:
        local = [[float(a), float(b)] for a, b in zip(res[0], res[1])]
    table = wam_dist.gather_track_scalars(torch.tensor(local, dtype=torch.float64), world * T, comm)

    if rank == 0:
        audio_s = world * T * SECONDS * args.steps
        ms_per_step = dt / args.steps * 1e3
        # dominant kernel by total device time
        dom = max(prof.items(), key=lambda kv: kv[1][1]) if prof else None
        roofline = None
        if dom:
            name, (cnt, total_ms) = dom
            launches_per_step = cnt / args.steps
            avg_ms = total_ms / cnt
            bpsc = KERNEL_BYTES.get(name)
            tracks_per_launch = min(T, GROUP) if chain_like else T
            if name == "k_stft":
                bpsc = 12 / NCH                     # 12 B per MONO sample: 4 read (both channels: 8) ... stated per sample-channel
            if bpsc is not None:
                alg_bytes = bpsc * tracks_per_launch * NCH * N
                achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
                roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                            "frac": achieved / 8000.0, "traffic": pmc_traffic(name, tracks_per_launch), "avg_launch_ms": avg_ms,
                            "launches_per_step": launches_per_step, "algorithmic_bytes_per_launch": alg_bytes,
                            "note": ("the chain kernels are bound by f64 VALU issue, not HBM (DESIGN.md 5-6): the chain needs ~2,600 VALU "
                                     "lane-instructions per 16 algorithmic bytes against a machine balance of ~5 per byte") if chain_like else
                                    ("two f64 K-weighting biquad scans + the 4-point peak interpolation per 4-byte sample: VALU-bound "
                                     "(DESIGN.md 5)" if wl == "meter" else "f32 butterflies in LDS; see DESIGN.md 5")}
            else:
                roofline = {"bound": "hbm", "kernel": name, "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None,
                            "traffic": None, "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step}
        kernels = {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        kernel_ms_per_step = sum(v[1] for v in prof.values()) / args.steps
        chain_bytes = (16 if chain_like else (6 if wl == "stft" else 4)) * T * NCH * N
        out = {
            "metric": ("x realtime (audio-s/s) full master chain+LUFS" if chain_like else
                       "x realtime (audio-s/s) 4096-pt STFT hop 1024" if wl == "stft" else "x realtime (audio-s/s) LUFS+true-peak meter"),
            "value": audio_s / dt, "unit": "x realtime (audio-s/s)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, numbers meaningless)" if rehearsal else ""),
            "config": {"workload": ({"chain": f"BASELINE configs[2]: {T} tracks/GPU", "single": "BASELINE configs[1]: one track",
                                     "longform": "BASELINE configs[3]: one long-form track"}[wl] +
                                    f" x {SECONDS // 60}-min stereo {SR // 1000} kHz, full export chain "
                                    "(input gain, de-harsh, exciter, tape warmth, punch, 30 Hz HPF + 18 kHz LPF, 5-band EQ 'aifix', glue comp, "
                                    "mono-bass M/S, -14 LUFS normalise, soft clip, TP limiter) + LUFS/true-peak of the result"
                                    if chain_like else
                                    f"BASELINE configs[4]: {T} tracks/GPU x 3-min stereo 48 kHz, 4096-point periodic-Hann STFT, hop 1024, of the mono mix, dB per bin"
                                    if wl == "stft" else
                                    f"BASELINE configs[0] batched: {T} tracks/GPU x 3-min stereo 48 kHz, integrated LUFS + true peak"),
                       "tracks_per_gpu": T, "seconds_per_track": SECONDS, "sample_rate": SR, "channels": NCH,
                       "settings": C2_SETTINGS if chain_like else None, "parallelism": f"track-sharded x{world}, no collective",
                       **({"tune": args.tune} if args.tune else {})},
            "roofline": roofline,
            "chain_hbm_frac": (chain_bytes / (kernel_ms_per_step * 1e-3) / 8e12) if kernel_ms_per_step > 0 else None,
            "kernel_ms_per_step": kernel_ms_per_step,
            "kernels": kernels,
        }
        tb = table.cpu()
        out["sample_result"] = {"lufs": float(tb[0, 0]), "true_peak_db": float(tb[0, 1]), "tracks_reported": int(tb.shape[0]),
                                "lufs_min": float(tb[:, 0].min()), "lufs_max": float(tb[:, 0].max()), "true_peak_db_max": float(tb[:, 1].max())}
        if chain_like:
            out["sample_result"].update(lufs_in=float(tb[0, 2]), gain_db=float(tb[0, 3]))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, SR)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
