#!/usr/bin/env python3
"""Copy the summaries of one profiles/refresh.sh run from gpurun_out/<tag>/ into profiles/ (tracked).
usage: python profiles/ingest.py <tag> <round-prefix, e.g. r01> [--pmc-only]
--pmc-only: just the PMC summaries (refresh.sh runs this on the GPU box between the counter passes and the bench runs, so that the
bench lines price their kernels with counters of the very same build; the commit then comes from $WAM_COMMIT, the box has no .git)."""
import os, shutil, subprocess, sys
tag, pre = sys.argv[1], sys.argv[2]
pmc_only = "--pmc-only" in sys.argv[3:]
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
src = os.path.join(root, "gpurun_out", tag)
cp = lambda a, b: shutil.copyfile(os.path.join(src, a), os.path.join(here, b)) if not pmc_only else None
cp("bench.json", f"{pre}_chain_T1024_bench.json")
cp("bench_meter.json", f"{pre}_meter_T1024_bench.json")
cp("bench_under_rocprof.json", f"{pre}_chain_T1024_bench_under_rocprof.json")
cp(os.path.join("trace", "trace_kernel_stats.csv"), f"{pre}_chain_T1024_kernel_stats.csv")
for extra, name in (("bench_stft.json", "stft_T1024"), ("bench_single.json", "chain_single"), ("bench_longform.json", "chain_longform"), ("bench_T128.json", "chain_T128")):
    if not pmc_only and os.path.exists(os.path.join(src, extra)) and os.path.getsize(os.path.join(src, extra)) > 0:
        cp(extra, f"{pre}_{name}_bench.json")
NSC = 128 * 2 * 8640000      # sample-channels per launch in the PMC passes (128 tracks)
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("WAM_COMMIT", "")
sys.path.insert(0, root)
import bench  # noqa: E402
here_digest = bench.source_digest()


def summary(pmc_dir, units, out_name):
    """pmc_summary.py over one directory of passes, plus the meta rows bench.py reads: tracks per launch, the digest of the kernel
    sources the passes were taken from (written on the GPU box) and the commit whose sources have that digest (HEAD, when they do)."""
    out = subprocess.run([sys.executable, os.path.join(here, "pmc_summary.py"), pmc_dir, str(units)], capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    pad = "," * (lines[0].count(",") - 1)
    dg = open(os.path.join(pmc_dir, "source_digest.txt")).read().strip() if os.path.exists(os.path.join(pmc_dir, "source_digest.txt")) else ""
    meta = ["#tracks_per_launch,128" + pad, "#source_digest," + dg + pad,
            "#commit," + (head if dg and dg == here_digest else f"unknown (sources differ from {head})") + pad]
    open(os.path.join(here, out_name), "w").write("\n".join([lines[0]] + meta + lines[1:]) + "\n")


summary(os.path.join(src, "pmc"), NSC, f"{pre}_pmc_summary.csv")
if os.path.isdir(os.path.join(src, "meter", "pmc")):       # the meter workload's passes (k_meter, k_gate)
    summary(os.path.join(src, "meter", "pmc"), NSC, f"{pre}_pmc_meter_summary.csv")
if os.path.isdir(os.path.join(src, "stft", "pmc")):        # the STFT workload's passes; per MONO sample (12 B per mono input sample)
    summary(os.path.join(src, "stft", "pmc"), NSC // 2, f"{pre}_pmc_stft_summary.csv")
print("ingested", tag)
