#!/usr/bin/env python3
"""Copy the summaries of one profiles/refresh.sh run from gpurun_out/<tag>/ into profiles/ (tracked).
usage: python profiles/ingest.py <tag> <round-prefix, e.g. r01>"""
import os, shutil, subprocess, sys
tag, pre = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
src = os.path.join(root, "gpurun_out", tag)
cp = lambda a, b: shutil.copyfile(os.path.join(src, a), os.path.join(here, b))
cp("bench.json", f"{pre}_chain_T1024_bench.json")
cp("bench_meter.json", f"{pre}_meter_T1024_bench.json")
cp("bench_under_rocprof.json", f"{pre}_chain_T1024_bench_under_rocprof.json")
cp(os.path.join("trace", "trace_kernel_stats.csv"), f"{pre}_chain_T1024_kernel_stats.csv")
NSC = 128 * 2 * 8640000      # sample-channels per launch in the PMC passes (128 tracks)
out = subprocess.run([sys.executable, os.path.join(here, "pmc_summary.py"), os.path.join(src, "pmc"), str(NSC)], capture_output=True, text=True, check=True).stdout
lines = out.splitlines()
ncol = lines[0].count(",")
lines.insert(1, "#tracks_per_launch,128" + "," * (ncol - 1))
open(os.path.join(here, f"{pre}_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
msrc = os.path.join(src, "meter", "pmc")
if os.path.isdir(msrc):       # the meter workload's passes (k_meter, k_gate)
    out = subprocess.run([sys.executable, os.path.join(here, "pmc_summary.py"), msrc, str(NSC)], capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    lines.insert(1, "#tracks_per_launch,128" + "," * (lines[0].count(",") - 1))
    open(os.path.join(here, f"{pre}_pmc_meter_summary.csv"), "w").write("\n".join(lines) + "\n")
print("ingested", tag)
