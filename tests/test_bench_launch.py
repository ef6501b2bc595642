"""bench.py's launch / rendezvous / reporting logic on a GPU-less host.

`python bench.py --gpus 2 --steps 20 --warmup 5` is what the driver types; without WORLD_SIZE in the
environment bench.py must start the two ranks itself (torch.distributed.run), and rank 0 must print ONE
JSON line with n_gpus = 2.  Here the ranks run the kernels' CPU emulation build under gloo at toy sizes
(DPC_BENCH_DRY_RUN=1, refused without the test hooks): the numbers mean nothing, the plumbing is what is
checked.  The same command on a GPU box runs the real thing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
                                                            "MASTER_PORT")}
    env.update(DPC_BENCH_DRY_RUN="1", DPC_TEST_HOOKS="1", OMP_NUM_THREADS="1")
    env.pop("DPC_POISON_BUFFERS", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_self_launches_and_prints_one_line(emu_library):
    r = _run(["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["unit"] == "views/s"
    assert j["config"]["global_batch"] == 2 * 2
    assert j["value"] > 0 and j["ms_per_step"] > 0 and "NOT a measurement" in j["data"]
    assert "cpu_baseline" not in j                   # rank 0 at N = 1 only


def test_config_3_ddp_training_step_self_launches(emu_library):
    """BASELINE configs[3]: `bench.py --gpus 2 --config 3` = the training step under DistributedDataParallel, one
    rank per GPU (here: gloo ranks on the emulation tier at toy sizes), fused point dropout on; one JSON line."""
    r = _run(["--gpus", "2", "--config", "3", "--steps", "2", "--warmup", "1", "--keep-prob", "0.5"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["training_step"] is True
    assert "DDP" in j["config"]["parallelism"] and j["config"]["global_batch"] == 2 * j["config"]["global_batch"] // 2
    assert j["value"] > 0 and j["steps_per_s"] > 0


def test_strong_scaling_splits_the_global_batch(emu_library):
    """SURVEY.md 8(e) secondary mode: the config's batch split over the ranks (here 2 views -> 1 per rank)."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["global_batch"] == 2


def test_config_3_graph_mode_uses_the_recordable_reducer(emu_library):
    """`bench.py --gpus 2 --config 3 --graph`: the gradient all-reduce comes from GradBuckets (hooks inside the
    backward pass) instead of DDP, so that a GPU run can record the whole step; under gloo it runs eagerly."""
    r = _run(["--gpus", "2", "--config", "3", "--graph", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and "GradBuckets" in j["config"]["parallelism"] and j["config"]["hip_graph"] is False
    assert "dry run" in j["config"]["hip_graph_note"] and j["value"] > 0


def test_gpus_1_runs_in_process(emu_library):
    r = _run(["--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1


def test_world_size_mismatch_is_an_error(emu_library):
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_dry_run_is_refused_without_test_hooks(emu_library):
    r = _run(["--steps", "1", "--warmup", "0"], {"DPC_TEST_HOOKS": "0"})
    assert r.returncode != 0 and "DPC_TEST_HOOKS" in r.stderr


def test_value_is_global_views_over_the_slowest_ranks_time(emu_library):
    """N > 1: `value` = (views of ALL ranks) x steps / (the MAX over ranks of the timed region) -- the last rank is made
    several seconds per step slower here (a step of the emulated kernels takes ~2 s by itself), and the line must carry its
    time, not rank 0's; the line also says which collective library and world size the ranks saw."""
    nap = 4.0
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0"], {"DPC_BENCH_TEST_SLEEP_LAST_RANK": str(nap)})
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 2
    assert j["ms_per_step"] >= nap * 1e3, j["ms_per_step"]                       # rank 1's pace
    views = j["config"]["global_batch"]
    assert views == 2 * 2
    assert abs(j["value"] - views / (j["ms_per_step"] * 1e-3)) <= 1e-9 * j["value"]
    assert "backend gloo, world 2" in j["config"]["parallelism"]


def test_force_dist_takes_the_distributed_path_with_one_rank(emu_library):
    """`bench.py --gpus 1 --force-dist`: a ONE-rank process group (here gloo; on a GPU box backend nccl = RCCL) and the
    distributed code path -- barriers and the MAX all-reduce go through the group, the line says what the rank saw."""
    r = _run(["--steps", "2", "--warmup", "1", "--force-dist"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["global_batch"] == 2
    assert "backend gloo, world 1" in j["config"]["parallelism"] and "forced one-rank group" in j["config"]["parallelism"]


def test_force_dist_training_step_runs_ddp_and_the_recordable_reducer(emu_library):
    """--config 3 under a forced one-rank group: the eager step is wrapped in DistributedDataParallel, --graph uses
    GradBuckets (whose bucket all-reduces are then issued although the world is 1)."""
    r = _run(["--config", "3", "--steps", "2", "--warmup", "1", "--force-dist"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and "(DDP)" in j["config"]["parallelism"] and "world 1" in j["config"]["parallelism"]
    assert "configs[2]" in j["config"]["workload"] and "DDP (RCCL all-reduce)" in j["config"]["workload"]
    r = _run(["--config", "3", "--graph", "--steps", "2", "--warmup", "1"], {"DPC_FORCE_DIST": "1"})   # the env form
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert "(GradBuckets)" in j["config"]["parallelism"] and "forced one-rank group" in j["config"]["parallelism"]
