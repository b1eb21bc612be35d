#!/usr/bin/env python3
"""bench.py - realtime factor of the MI355X mastering backend (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic 3-min stereo 48 kHz
tracks already resident in HBM: the full export chain + LUFS/true-peak metering
(`--workload chain`, BASELINE.json configs[2]) or metering only (`--workload meter`,
configs[0]).

N > 1.  The batch is the FIXED 1,024-track batch of configs[2], sharded track-wise:
rank r renders tracks [ceil(r T / N), ceil((r + 1) T / N)) (`wam_dist.shard_range`), so the
line reports `"scaling": "strong"`.  `--tracks-per-gpu X` switches to X tracks on every GPU
(`"scaling": "weak"`).  Tracks are independent, so ranks never exchange audio: no data-path
collective; torch.distributed carries only the barrier, the max-over-ranks
reduction of the step time and the per-track scalars that rank 0 prints.

Launch.  Under an external launcher (`python -m torch.distributed.run ... bench.py --gpus N`)
the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  Without one
(`python bench.py --gpus N`, WORLD_SIZE unset) this process starts the N rank processes itself
- fresh children, before anything here has touched the GPU - relays rank 0's JSON line and
exits with the worst exit code.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the
launch stream through the library's profiling hooks), `roofline_valu` (the same kernel and
the whole chain against the f64 VALU issue peak, instruction counts from the committed PMC
passes) and `cpu_baseline` (the CPU oracle timed on a bounded sample, rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "web-audio-mastering_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SR = 48000
SECONDS = 180
NCH = 2
N = SR * SECONDS
BATCH_TRACKS = 1024            # BASELINE.json configs[2] / configs[4]: the corpus

# Settings of configs[1]/[2] (SURVEY.md 8d "C2"): UI defaults (web/index.html:178-295)
# + glue compression + the 'aifix' EQ preset (web/lib/presets/eq.js:21).
DEFAULT_SETTINGS = dict(
    normalizeLoudness=True, targetLufs=-14, truePeakLimit=True, truePeakCeiling=-1, cleanLowEnd=True,
    glueCompression=False, deharsh=True, stereoWidth=100, centerBass=True, cutMud=False, addAir=True,
    tapeWarmth=True, autoLevel=False, addPunch=True, inputGain=0,
    eqLow=0, eqLowMid=0, eqMid=0, eqHighMid=0, eqHigh=0)
C2_SETTINGS = dict(DEFAULT_SETTINGS, glueCompression=True, eqLow=1, eqLowMid=-2, eqMid=1, eqHighMid=-1, eqHigh=2)
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
    "k_dh_bandsum": 8,     # reads the magnitudes k_dh_bands_x left in scratch (4 KB per 512-sample hop = 8 B per sample-channel; not stage I/O)
    "k_sc": 8,             # read x, write y (the envelope stays on chip)
    "k_lim_tm": 6,         # read x, write transient map
    "k_lim_gain": 8,       # read x + transient map, write gain envelope
    "k_lim_apply": 10,     # read x + envelope, write y
}
NOMINAL_MHZ = 2400.0


def valu_peak(mhz):
    """f64 VALU lane-instructions / s: 256 CU x 4 SIMD x 64 lanes x clock, one issue per 4 cycles."""
    return 256 * 4 * 64 * mhz * 1e6 / 4


def source_digest():
    """sha1 over the kernel sources (csrc/*.hip, *.hpp, *.cpp): what a committed PMC summary was taken from.  The GPU box has no
    .git, so this - not a commit hash - is what bench.py can compare at run time; profiles/ingest.py records both."""
    import glob
    import hashlib
    h = hashlib.sha1()
    src = os.path.join(ROOT, "web-audio-mastering_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.hpp")) + glob.glob(os.path.join(src, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]
GROUP = 256      # tracks per render group inside the library (csrc/chain_api.hip): one launch of a chain kernel covers a group
# kernels that go through a group in several launches (channels that fit the de-harsh magnitude scratch): the committed PMC figures
# are per PASS over the launch's tracks (profiles/pmc_summary.py sums their launches)
SPLIT_KERNELS = {"k_dh_bands_x", "k_dh_bandsum"}


class SynthOpts(C.Structure):
    _fields_ = [("amp", C.c_double), ("dc", C.c_double), ("click_amp", C.c_double),
                ("quiet_from", C.c_int64), ("quiet_to", C.c_int64), ("quiet_gain", C.c_double),
                ("quiet2_from", C.c_int64), ("quiet2_to", C.c_int64), ("quiet2_gain", C.c_double)]


def synth_dev(t, ntracks, idx0, opts, amp_step=0.0, sr=SR, amp0=None):
    import wam
    o = SynthOpts(opts.get("amp", 1.0) if amp0 is None else amp0, opts.get("dc", 0.0), opts.get("clickAmp", 0.0), -1, -1, 0.0, -1, -1, 0.0)
    rc = wam.lib().wam_synth_dev(C.c_void_p(t.data_ptr()), C.c_int(ntracks), C.c_int(NCH), C.c_int64(t.shape[-1]), C.c_double(sr),
                                 C.c_int(idx0), C.byref(o), C.c_double(amp_step), C.c_void_p(0))
    if rc:
        raise wam.WamError(wam.lib().wam_last_error().decode())


def profile_read():
    import wam
    buf = C.create_string_buffer(1 << 16)
    wam.lib().wam_profile_read(buf, C.c_size_t(len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


# ---- committed PMC passes (profiles/rNN_pmc_summary.csv: separate rocprofv3 --pmc runs of this same bench) --------------
def _pmc_rows():
    """kernel -> row of the newest committed PMC summary that has it (chain passes, then the meter-workload passes)."""
    import csv
    import glob
    rows, tracks = {}, {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")) + glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_meter_summary.csv")) +
                   glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_stft_summary.csv")), key=os.path.basename)
    for path in files:                                    # later rounds overwrite earlier ones
        with open(path) as f:
            rs = list(csv.DictReader(f))
        meta = {r["kernel"]: r["launches"] for r in rs if r["kernel"].startswith("#")}
        ref = float(meta["#tracks_per_launch"]) if "#tracks_per_launch" in meta else None
        for r in rs:
            if not r["kernel"].startswith("#"):
                rows[r["kernel"]] = r
                tracks[r["kernel"]] = ref
                r["_file"] = os.path.basename(path)
                r["_commit"] = meta.get("#commit")
                r["_digest"] = meta.get("#source_digest")
    return rows, tracks


def pmc_provenance(rows, kernels):
    """Which committed summaries the instruction counts / traffic of `kernels` come from, and whether the kernel sources have
    changed since they were taken."""
    now = source_digest()
    files = {}
    for k_ in kernels:
        r = rows.get(k_)
        if r:
            files[r["_file"]] = {"commit": r.get("_commit"), "source_digest": r.get("_digest"),
                                 "sources_unchanged": (r.get("_digest") == now) if r.get("_digest") else None}
    return {"source_digest_now": now, "files": files}


def pmc_traffic(kernel, tracks_per_launch, rows=None, tracks=None):
    """HBM bytes per launch of `kernel`: FETCH_SIZE (doubled per the gfx950 correction) + WRITE_SIZE, scaled to the tracks a launch covers."""
    if rows is None:
        rows, tracks = _pmc_rows()
    r = rows.get(kernel)
    if not r:
        return None
    try:
        b = (float(r["fetch_GB_per_launch_x2"]) + float(r["write_GB_per_launch"])) * 1e9
    except ValueError:
        return None
    ref = tracks.get(kernel)
    if ref and ref != tracks_per_launch:
        b *= tracks_per_launch / ref
    return b


def pmc_instr(kernel, rows):
    """VALU lane-instructions per sample-channel of `kernel` (SQ_INSTS_VALU x 64 / sample-channels of the profiled launch)."""
    r = rows.get(kernel)
    try:
        return float(r["valu_lane_instr_per_sample_channel"]) if r else None
    except ValueError:
        return None


# ---- CPU baseline -------------------------------------------------------------------------------------------------------
_CPU_WORKER = r"""
import sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import json
import oracle_lib as O
wl, sr, secs = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
settings, synth = json.loads(sys.argv[6]), json.loads(sys.argv[7])
idx = [int(a) for a in sys.argv[8:]]
O.lib()
print("ready", flush=True)
sys.stdin.readline()                      # all workers start together
t0 = time.perf_counter()
for i in idx:
    x = O.synth(i, 2, secs * sr, sr, synth)
    if wl == "chain":
        O.render_chain(x, sr, settings, "export")
    elif wl == "stft":
        O.stft_db(x, sr, 4096, 1024)
    else:
        O.measure_lufs(x, sr); O.true_peak_db(x)
print(time.perf_counter() - t0, flush=True)
"""


def cpu_all_cores(kind, sr, secs, per_proc):
    """The oracle on every host core this job may use: P processes, track-sharded, `per_proc` whole tracks each, started together.
    P = min(16, cpu_count): the GPU box hands a one-GPU job 16 of its cores."""
    procs_n = max(1, min(16, os.cpu_count() or 1))
    procs = []
    for p in range(procs_n):
        idx = [str(1000 + p * per_proc + k) for k in range(per_proc)]
        procs.append(subprocess.Popen([sys.executable, "-c", _CPU_WORKER, os.path.join(ROOT, "tests"), os.path.join(ROOT, "web-audio-mastering_amd", "python"),
                                       kind, str(sr), str(secs), json.dumps(C2_SETTINGS), json.dumps(SYNTH)] + idx,
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=dict(os.environ, OMP_NUM_THREADS="1")))
    try:
        for p in procs:
            assert p.stdout.readline().strip() == "ready"
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        for p in procs:
            p.stdout.readline()
        dt = time.perf_counter() - t0
        for p in procs:
            p.wait(timeout=30)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    tracks = procs_n * per_proc
    return {"value": tracks * secs / dt, "unit": "x realtime (audio-s/s)", "processes": procs_n, "kind": "port",
            "sample": f"{tracks} tracks x {secs} s, {per_proc} per process, {procs_n} single-threaded oracle processes started together, {dt:.1f} s wall"}


def cpu_baseline(workload, sr=SR):
    """Oracle (CPU restatement of the reference, kind 'port') on a bounded sample of the same workload: whole tracks of the
    workload's shape (3-min stereo; a 60 s excerpt for the long-form case), one after the other on ONE core (the reference is
    single-threaded per worker), until about 10 s of CPU time have been spent; then, for the batch workloads, the same port on
    all the host cores of this job, track-sharded over processes."""
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
    out = {"value": done * secs / dt, "unit": "x realtime (audio-s/s)", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
           "sample": f"{done} track(s) x {secs} s stereo {sr // 1000} kHz, same synthetic generator and settings, oracle/wam_oracle.c "
                     f"(C restatement of the reference JS, one thread), {dt:.1f} s of CPU time"}
    if workload in ("chain", "meter", "stft"):
        kind = "chain" if chain else workload
        per_core_s = dt / done
        per_proc = max(1, min(8, int(round(10.0 / per_core_s))))          # ~10 s of wall time
        try:
            out["all_cores"] = cpu_all_cores(kind, sr, secs, per_proc)
        except Exception as e:                                              # a baseline must never sink the bench line
            out["all_cores"] = {"error": str(e)}
    return out


# ---- self-launch ----------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


LAUNCH_TIMEOUT_S = 1500.0      # default limit of a self-launched job (WAM_BENCH_TIMEOUT overrides): a default bench step set is ~1 min per rank


def launch_ranks(n, argv, script=None, extra_env=None, timeout=None):
    """Start `n` rank processes of `script` (default: this file) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set, relay rank 0's stdout to ours and every rank's stderr (and the other ranks' stdout) to stderr with a rank prefix.
    The parent never initialises the GPU; children are fresh interpreters.  Returns the worst exit code; after `timeout`
    seconds (default LAUNCH_TIMEOUT_S) the ranks still running are named on stderr and killed, and the code is 124."""
    if timeout is None:
        timeout = float(os.environ.get("WAM_BENCH_TIMEOUT", LAUNCH_TIMEOUT_S))
    script = script or os.path.abspath(__file__)
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WAM_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))

    def pump(stream, sink, prefix):
        for line in stream:
            sink.write(prefix + line)
            sink.flush()

    threads = []
    for r, p in enumerate(procs):
        threads.append(threading.Thread(target=pump, args=(p.stdout, sys.stdout if r == 0 else sys.stderr, "" if r == 0 else f"[rank {r}] "), daemon=True))
        threads.append(threading.Thread(target=pump, args=(p.stderr, sys.stderr, f"[rank {r}] "), daemon=True))
    for t in threads:
        t.start()
    deadline = time.time() + timeout if timeout else None
    worst = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is not None:
                pending.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:            # one rank failed: the others would wait in a collective for ever
                        q.terminate()
        if deadline and time.time() > deadline:
            stuck = [procs.index(q) for q in pending]
            print(f"bench.py: rank(s) {stuck} still running after {timeout:.0f} s (stuck in rendezvous or a collective?): killed", file=sys.stderr, flush=True)
            for q in pending:
                q.kill()
            worst = worst or 124
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    for t in threads:
        t.join(timeout=5)
    return worst


class stdout_to_stderr:
    """gloo announces its connections on the C stdout; the bench contract is ONE JSON line there, so fd 1 points at fd 2 meanwhile."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def resolve_shard(args, rank, world):
    """(tracks of this rank, first global track index, total tracks, scaling) for the batch workloads."""
    import wam_dist
    if args.workload in ("single", "longform"):
        T = args.tracks_per_gpu or 1
        return T, rank * T, world * T, "weak"            # one track does not shard: replicas only
    if args.tracks_per_gpu:
        T = args.tracks_per_gpu
        return T, rank * T, world * T, "weak"
    total = args.tracks or BATCH_TRACKS
    lo, hi = wam_dist.shard_range(total, rank, world)
    return hi - lo, lo, total, "strong"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tracks", type=int, default=None, help=f"total tracks of the batch, sharded over the ranks (default {BATCH_TRACKS}: BASELINE configs[2]; strong scaling)")
    ap.add_argument("--tracks-per-gpu", type=int, default=None, help="weak-scaling mode: this many tracks on EVERY GPU (default for single / longform: 1)")
    ap.add_argument("--workload", choices=["chain", "meter", "single", "longform", "stft"], default="chain",
                    help="chain: configs[2] (default, the bench line); meter: configs[0] batched; single: configs[1], one 3-min track; "
                         "longform: configs[3], one 60-min 96 kHz track; stft: configs[4], 4096-point STFT hop 1024 over the corpus")
    ap.add_argument("--seconds", type=int, default=None, help="track length (default 180; 3600 for longform) - shorter tracks are for rehearsals only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plan", action="store_true", help="resolve the launch (ranks, shards, scaling), rendezvous over gloo, print the plan as JSON; touches no GPU")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="wam_tune knob for an experiment (e.g. dh_exact=1); recorded in config")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no external launcher: start the ranks ourselves, before anything in this process has touched the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size is used", file=sys.stderr)

    import torch  # noqa: E402  (first: libwam_hip.so must bind to the same HIP runtime as torch)
    import torch.distributed as dist  # noqa: E402
    import wam_dist  # noqa: E402

    # WAM_BENCH_REHEARSAL=1: run the N-rank code path on a box with ONE GPU (all ranks on cuda:0, gloo for the three collectives);
    # a correctness rehearsal of the launch / sharding / gathering plumbing only - its numbers mean nothing.
    rehearsal = os.environ.get("WAM_BENCH_REHEARSAL") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")

    if args.plan:
        if world > 1:
            with stdout_to_stderr():
                dist.init_process_group("gloo", rank=rank, world_size=world)
        T, idx0, total, scaling = resolve_shard(args, rank, world)
        table = wam_dist.gather_track_scalars(torch.tensor([[float(idx0 + t), float(rank)] for t in range(T)], dtype=torch.float64).reshape(T, 2), total)
        if rank == 0:
            owners = [int(v) for v in table[:, 1].tolist()]
            print(json.dumps({"plan": True, "n_gpus": world, "workload": args.workload, "tracks_total": int(table.shape[0]), "scaling": scaling,
                              "shards": [[owners.index(r), len(owners) - owners[::-1].index(r)] if r in owners else [0, 0] for r in range(world)],
                              "track_index_ok": [int(v) for v in table[:, 0].tolist()] == list(range(int(table.shape[0]))),
                              "self_launched": os.environ.get("WAM_BENCH_SELF_LAUNCHED") == "1"}))
        if world > 1:
            dist.destroy_process_group()
        return

    import wam  # noqa: E402
    if world > 1:
        if rehearsal:
            torch.cuda.set_device(0)
            with stdout_to_stderr():
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
    if args.seconds:
        SECONDS = args.seconds
    N = SR * SECONDS
    T, idx0, total, scaling = resolve_shard(args, rank, world)
    amp_step = 0.3 / max(total, 1)                              # distinct seed per track (= its global index), levels 1.6 .. 1.9 over the batch
    x = torch.empty((max(T, 1), NCH, N), dtype=torch.float32, device=dev)
    if T > 0:
        synth_dev(x, T, idx0, SYNTH, amp_step=amp_step, sr=SR, amp0=SYNTH["amp"] + idx0 * amp_step)
    y = torch.empty_like(x) if chain_like else None
    NFFT, HOP = 4096, 1024
    if wl == "stft":
        nframes = (N - NFFT) // HOP + 1
        spec = torch.empty((max(T, 1), nframes, NFFT // 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        if T == 0:
            return [] if chain_like else ([], [])
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
    rank_ms = [float(v) for v in wam_dist.gather_track_scalars(torch.tensor([[dt / args.steps * 1e3]], dtype=torch.float64), world, comm)[:, 0].tolist()] if world > 1 else [dt / args.steps * 1e3]
    dt = wam_dist.max_over_ranks(dt, comm)
    # per-track scalars of every shard, in global track order (the only data that leaves a rank)
    if chain_like:
        local = [[r.lufs, r.true_peak_db, r.lufs_in, r.gain_db] for r in res]
        k = 4
    elif wl == "stft":
        local = [[float(spec[t].max()), float(spec[t, :, 1:].float().mean())] for t in range(min(T, 4))] + [[float("nan"), float("nan")]] * max(0, T - 4)
        k = 2
    else:
        local = [[float(a), float(b)] for a, b in zip(res[0], res[1])]
        k = 2
    table = wam_dist.gather_track_scalars(torch.tensor(local, dtype=torch.float64).reshape(T, k), total, comm)

    # the shader clock the chip holds under f64 load, measured right behind the timed region (the chip is as warm as it gets)
    clock_mhz = None
    if rank == 0:
        mhz = C.c_double(0)
        if hasattr(wam.lib(), "wam_probe_shader_clock") and wam.lib().wam_probe_shader_clock(C.c_double(25.0), C.byref(mhz)) == 0 and mhz.value > 0:
            clock_mhz = mhz.value
    if rank == 0:
        audio_s = total * SECONDS * args.steps
        ms_per_step = dt / args.steps * 1e3
        rows, rtracks = _pmc_rows()
        VALU_PEAK = valu_peak(clock_mhz or NOMINAL_MHZ)
        # dominant kernel by total device time (rank 0's shard; every rank runs the same kernels on its own tracks)
        dom = max(prof.items(), key=lambda kv: kv[1][1]) if prof else None
        roofline = None
        roofline_valu = None
        tracks_per_launch = min(T, GROUP) if chain_like else T
        units_per_launch = tracks_per_launch * NCH * N          # sample-channels one launch of a kernel covers
        if dom:
            name, (cnt, total_ms) = dom
            launches_per_step = cnt / args.steps
            avg_ms = total_ms / cnt
            bpsc = KERNEL_BYTES.get(name)
            pmc_name = name
            if name == "k_stft":
                bpsc = 12 / NCH                     # 12 B per MONO sample: 4 read (both channels: 8) ... stated per sample-channel
                pmc_name = "k_stft_wave"            # the profiling hook's name covers the three STFT kernels; the default 4096-point one is this
            if bpsc is not None:
                alg_bytes = bpsc * units_per_launch
                achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
                roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                            "frac": achieved / 8000.0, "traffic": pmc_traffic(pmc_name, tracks_per_launch, rows, rtracks), "avg_launch_ms": avg_ms,
                            "launches_per_step": launches_per_step, "algorithmic_bytes_per_launch": alg_bytes,
                            "note": ("the chain kernels are bound by f64 VALU issue, not HBM (DESIGN.md 5-6): see roofline_valu") if chain_like else
                                    ("two f64 K-weighting biquad scans + the 4-point peak interpolation per 4-byte sample: VALU-bound "
                                     "(DESIGN.md 5)" if wl == "meter" else
                                     "radix-2 butterflies in double with an f32 store after each (the reference's arithmetic), one wavefront per frame: "
                                     "VALU-bound, see roofline_valu and DESIGN.md 5")}
            else:
                roofline = {"bound": "hbm", "kernel": name, "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None,
                            "traffic": None, "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step}
            ipu = pmc_instr(pmc_name, rows)
            if ipu is not None:
                # (the STFT summary counts per MONO sample: its units are tracks x n)
                units_v = units_per_launch // NCH if wl == "stft" else units_per_launch
                ach = ipu * units_v / (avg_ms * 1e-3)
                roofline_valu = {"bound": "valu_f64", "kernel": pmc_name, "instr_per_unit": ipu, "unit_of_work": "mono sample" if wl == "stft" else "sample-channel",
                                 "units_per_launch": units_v, "avg_launch_ms": avg_ms, "achieved": ach, "peak": VALU_PEAK, "unit": "lane-instr/s", "frac": ach / VALU_PEAK,
                                 "shader_clock_mhz": clock_mhz, "peak_at_nominal_2400_mhz": valu_peak(NOMINAL_MHZ),
                                 "source": rows[pmc_name]["_file"]}
        kernels = {k_: {"launches": v[0], "total_ms": round(v[1], 3)} for k_, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        kernel_ms_per_step = sum(v[1] for v in prof.values()) / args.steps
        chain_bytes = (16 if chain_like else (6 if wl == "stft" else 4)) * T * NCH * N
        if roofline_valu is not None and kernel_ms_per_step > 0 and wl != "stft":
            # the whole step against the same peak: sum over kernels of (instructions per unit x units the kernel covered in a step)
            instr = 0.0
            missing = []
            groups_per_step = -(-T // GROUP) if chain_like else 1
            passes = lambda k_, cnt: groups_per_step if k_ in SPLIT_KERNELS else cnt / args.steps
            for k_, (cnt, _) in prof.items():
                ipu = pmc_instr(k_, rows)
                if ipu is None:
                    missing.append(k_)
                    continue
                instr += ipu * units_per_launch * passes(k_, cnt)         # per pass over a group x passes per step
            roofline_valu["chain"] = {"instr_per_unit": instr / (T * NCH * N), "kernel_ms_per_step": kernel_ms_per_step,
                                      "achieved": instr / (kernel_ms_per_step * 1e-3), "frac": instr / (kernel_ms_per_step * 1e-3) / VALU_PEAK,
                                      "kernels_without_counters": missing}
        traffic_step = None
        if prof:
            tb_, ok = 0.0, True
            groups_per_step = -(-T // GROUP) if chain_like else 1
            for k_, (cnt, _) in prof.items():
                b = pmc_traffic("k_stft_wave" if k_ == "k_stft" else k_, tracks_per_launch, rows, rtracks)
                if b is None:
                    ok = False
                    continue
                tb_ += b * (groups_per_step if k_ in SPLIT_KERNELS else cnt / args.steps)      # per pass over a group x passes per step
            traffic_step = tb_ if ok or tb_ > 0 else None
        cfg_workload = ({"chain": f"BASELINE configs[2]: {total} tracks" + (f", {T} on rank 0" if world > 1 else ""), "single": "BASELINE configs[1]: one track",
                         "longform": "BASELINE configs[3]: one long-form track"}.get(wl, "") +
                        f" x {SECONDS // 60}-min stereo {SR // 1000} kHz, full export chain "
                        "(input gain, de-harsh, exciter, tape warmth, punch, 30 Hz HPF + 18 kHz LPF, 5-band EQ 'aifix', glue comp, "
                        "mono-bass M/S, -14 LUFS normalise, soft clip, TP limiter) + LUFS/true-peak of the result"
                        if chain_like else
                        f"BASELINE configs[4]: {total} tracks x 3-min stereo 48 kHz, 4096-point periodic-Hann STFT, hop 1024, of the mono mix, dB per bin"
                        if wl == "stft" else
                        f"BASELINE configs[0] batched: {total} tracks x 3-min stereo 48 kHz, integrated LUFS + true peak")
        out = {
            "metric": ("x realtime (audio-s/s) full master chain+LUFS" if chain_like else
                       "x realtime (audio-s/s) 4096-pt STFT hop 1024" if wl == "stft" else "x realtime (audio-s/s) LUFS+true-peak meter"),
            "value": audio_s / dt, "unit": "x realtime (audio-s/s)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            # the arithmetic type of the path: IEEE double with f32 stores (the reference's JavaScript numbers over Float32Arrays)
            "dtype": "f64 (f32 stores after every butterfly)" if wl == "stft" else "f64",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, numbers meaningless)" if rehearsal else ""),
            "config": {"workload": cfg_workload, "tracks_total": total, "tracks_per_gpu": T if scaling == "weak" else None,
                       "tracks_rank0": T, "seconds_per_track": SECONDS, "sample_rate": SR, "channels": NCH,
                       "settings": C2_SETTINGS if chain_like else None, "parallelism": f"track-sharded x{world}, no collective",
                       **({"eq": "spec-defined (Web Audio biquad formulas), parity unpinned: the reference's EQ stage throws (SURVEY.md 0.1)"} if chain_like else {}),
                       **({"tune": args.tune} if args.tune else {})},
            "roofline": roofline,
            "roofline_valu": roofline_valu,
            "chain_hbm_frac": (chain_bytes / (kernel_ms_per_step * 1e-3) / 8e12) if kernel_ms_per_step > 0 else None,
            "chain_algorithmic_bytes": chain_bytes,
            "chain_traffic_bytes": traffic_step,
            "kernel_ms_per_step": kernel_ms_per_step,
            "kernels": kernels,
            "shader_clock_mhz": clock_mhz,
            "pmc_provenance": pmc_provenance(rows, ["k_stft_wave" if k_ == "k_stft" else k_ for k_ in prof]),
        }
        if world > 1:
            out["ms_per_step_by_rank"] = rank_ms          # a straggler shows here; ms_per_step is their maximum
        tb = table.cpu()
        if wl == "stft":
            out["sample_result"] = {"spectrum_db_max_track0": float(tb[0, 0]), "spectrum_db_mean_track0": float(tb[0, 1]), "tracks_reported": int(tb.shape[0]),
                                    "frames_per_track": int(nframes), "bins": NFFT // 2}
        else:
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
