"""N > 1 path on CPU: track sharding + barrier + MAX-reduce + scalar gather with world_size 2 over gloo.

The per-track compute of the real path is a GPU call (wam.batch_render_dev / batch_meter_dev); here each rank
meters its shard with the CPU oracle instead, which is enough to prove that the sharding arithmetic, the seeds
(track index = global index) and the gather put every track's numbers in the right row.
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-audio-mastering_amd", "python"))

import wam_dist  # noqa: E402

WORKER = r"""
import json, os, sys, time
sys.path.insert(0, os.path.join({root!r}, "web-audio-mastering_amd", "python"))
sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
import numpy as np
import wam_dist
import oracle_lib as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
NT, SR, N = {ntracks}, 48000, 48000
lo, hi = wam_dist.shard_range(NT, rank, world)
wam_dist.barrier()
t0 = time.perf_counter()
rows = []
for i in range(lo, hi):
    x = O.synth(i, 2, N, SR, dict(amp=1.0 + 0.1 * i))
    rows.append([O.measure_lufs(x, SR), O.true_peak_db(x)])
wam_dist.barrier()
dt = time.perf_counter() - t0 + (0.5 if rank == 1 else 0.0)     # rank 1 pretends to be slower
slowest = wam_dist.max_over_ranks(dt)
table = wam_dist.gather_track_scalars(np.array(rows, dtype=np.float64).reshape(hi - lo, 2), NT)
if rank == 0:
    json.dump(dict(table=table.tolist(), slowest=slowest, mine=dt, lo=lo, hi=hi), open({out!r}, "w"))
dist.destroy_process_group()
"""


def test_shard_ranges_partition():
    for nt in (0, 1, 2, 5, 7, 128, 1000, 1024):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                lo, hi = wam_dist.shard_range(nt, r, world)
                assert 0 <= lo <= hi <= nt
                got += list(range(lo, hi))
                for i in range(lo, hi):
                    assert wam_dist.owner_of(i, nt, world) == r
            assert got == list(range(nt))
            sizes = [wam_dist.shard_range(nt, r, world)[1] - wam_dist.shard_range(nt, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert wam_dist.shard_range(1024, 3, 8) == (384, 512)


def test_world2_gloo_gather(tmp_path):
    import oracle_lib as O
    ntracks = 5                                   # odd on purpose: shards of 3 and 2... (ceil split: 0..2 | 3..4)
    out = str(tmp_path / "rank0.json")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, ntracks=ntracks, out=out))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        o, _ = p.communicate(timeout=240)
        assert p.returncode == 0, o
    got = json.load(open(out))
    want = []
    for i in range(ntracks):
        x = O.synth(i, 2, 48000, 48000, dict(amp=1.0 + 0.1 * i))
        want.append([O.measure_lufs(x, 48000), O.true_peak_db(x)])
    assert np.array_equal(np.array(got["table"]), np.array(want))          # same code, same inputs: identical
    assert got["slowest"] >= got["mine"] and got["slowest"] >= 0.5          # MAX over ranks saw rank 1's time
    assert (got["lo"], got["hi"]) == wam_dist.shard_range(ntracks, 0, 2)
