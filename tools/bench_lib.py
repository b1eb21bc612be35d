"""Run bench.py against another build of the library (a measurement aid): python tools/bench_lib.py <lib.so> [bench args]"""
import os, sys, runpy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "web-audio-mastering_amd", "python"))
import torch  # noqa: F401  (before the library: HIP runtime identity)
import wam
wam.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
