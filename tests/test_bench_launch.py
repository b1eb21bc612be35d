"""bench.py's own N-rank launch (no external launcher), on CPU: `--plan` resolves ranks / shards / scaling and
rendezvouses over gloo without touching a GPU, so the spawn logic that `python bench.py --gpus N` uses on the GPU
box is the code that runs here."""
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
sys.path.insert(0, ROOT)


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "WAM_BENCH_SELF_LAUNCHED")}
    env.update(OMP_NUM_THREADS="1", **extra)
    return env


def run_plan(*argv, env=None):
    p = subprocess.run([sys.executable, BENCH, "--plan", *argv], env=env or clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout            # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_self_launch_two_ranks_fixed_batch():
    plan = run_plan("--gpus", "2")
    assert plan["n_gpus"] == 2 and plan["self_launched"] is True
    assert plan["tracks_total"] == 1024 and plan["scaling"] == "strong"          # BASELINE configs[2]: the fixed batch, sharded
    assert plan["shards"] == [[0, 512], [512, 1024]] and plan["track_index_ok"]


def test_self_launch_uneven_shards_and_weak_mode():
    plan = run_plan("--gpus", "3", "--tracks", "10")
    assert plan["shards"] == [[0, 4], [4, 7], [7, 10]] and plan["scaling"] == "strong" and plan["track_index_ok"]
    plan = run_plan("--gpus", "2", "--tracks-per-gpu", "5")
    assert plan["tracks_total"] == 10 and plan["scaling"] == "weak" and plan["shards"] == [[0, 5], [5, 10]]
    plan = run_plan("--gpus", "2", "--workload", "single")                       # one track does not shard: replicas
    assert plan["tracks_total"] == 2 and plan["scaling"] == "weak"


def test_single_rank_does_not_spawn():
    plan = run_plan()
    assert plan["n_gpus"] == 1 and plan["self_launched"] is False and plan["shards"] == [[0, 1024]]


def test_external_launcher_env_is_respected():
    """WORLD_SIZE set (torch.distributed.run's contract): no children, the ranks are the launcher's."""
    import bench
    port = bench.free_port()
    procs = []
    for r in range(2):
        env = clean_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--plan", "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    plan = json.loads(outs[0][0].strip().splitlines()[-1])
    assert plan["n_gpus"] == 2 and plan["self_launched"] is False and plan["shards"] == [[0, 512], [512, 1024]]
    assert outs[1][0].strip() == ""                                               # only rank 0 prints


def test_failing_rank_ends_the_job(tmp_path):
    """A rank that dies must not leave the others waiting in a collective: the launcher ends them and reports the failure."""
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(textwrap.dedent("""
        import os, sys, time
        assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert os.environ["LOCAL_RANK"] == os.environ["RANK"]
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)
    """))
    t0 = time.time()
    rc = bench.launch_ranks(3, [], script=str(stub), timeout=120)
    assert rc == 7 and time.time() - t0 < 60


def test_cpu_all_cores_leg_shape():
    import bench
    r = bench.cpu_all_cores("meter", 48000, 2, 1)
    assert r["kind"] == "port" and r["processes"] >= 1 and r["value"] > 0


def test_stuck_rank_is_named_and_the_job_ends_non_zero(tmp_path, capfd, monkeypatch):
    """A rank that never finishes (stuck in rendezvous or in a collective) must not hold the job until the driver's own limit: after
    the launcher's timeout - a default exists, WAM_BENCH_TIMEOUT overrides it - the ranks still running are named and killed."""
    import bench
    assert bench.LAUNCH_TIMEOUT_S > 0                                              # there IS a default
    stub = tmp_path / "stub.py"
    stub.write_text(textwrap.dedent("""
        import os, time
        if os.environ["RANK"] == "0":
            raise SystemExit(0)
        time.sleep(600)
    """))
    monkeypatch.setenv("WAM_BENCH_TIMEOUT", "3")
    t0 = time.time()
    rc = bench.launch_ranks(3, [], script=str(stub))                               # no explicit timeout: the default path
    err = capfd.readouterr().err
    assert rc == 124 and time.time() - t0 < 30
    assert "rank(s) [1, 2] still running after 3 s" in err


def test_per_rank_step_times_are_gathered_in_rank_order():
    """The N > 1 bench line carries every rank's own ms_per_step (a straggler is visible; the headline is their maximum): the
    gathering is wam_dist.gather_track_scalars with one row per rank - checked here over gloo with three ranks."""
    code = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, os.path.join(%r, "web-audio-mastering_amd", "python"))
        import torch, torch.distributed as dist, wam_dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ms = 100.0 + 7 * rank
        got = wam_dist.gather_track_scalars(torch.tensor([[ms]], dtype=torch.float64), world, "cpu")[:, 0].tolist()
        worst = wam_dist.max_over_ranks(ms, "cpu")
        if rank == 0:
            print(json.dumps({"by_rank": got, "max": worst}))
        dist.destroy_process_group()
    """ % ROOT)
    import bench
    port = bench.free_port()
    procs = [subprocess.Popen([sys.executable, "-c", code], env=clean_env(RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r = json.loads(outs[0][0].strip().splitlines()[-1])
    assert r["by_rank"] == [100.0, 107.0, 114.0] and r["max"] == 114.0
