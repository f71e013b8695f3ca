"""bench.py's N > 1 control flow without a GPU: `run_contract` + `finish` driven under gloo at world 1 / 2 / 3 with
a device-free stand-in of `BodyJob`.  What must hold for the driver's 8-GPU run to be readable:

* every rank times exactly `steps` steps after `warmup` untimed ones, the exchange runs once untimed and once inside the bracket;
* the reported time is the MAX over ranks (the slow rank's), value = all ranks' frames / that time;
* only rank 0 emits a line, only rank 0 runs the extra measurement legs, and it runs them AFTER the process group is gone — no
  rank sits in a collective while rank 0 measures;
* every rank checks what it timed (selfcheck) before leaving.
"""
import json
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class StubJob:
    """Same methods as bench.BodyJob; a step is a sleep whose length depends on the rank, rows are the global clip numbers."""

    def __init__(self, world, rank, log):
        self.world, self.rank, self.log, self.steps_run = world, rank, log, []
        self.rows = None
        self.fail_rank = -1

    def _say(self, what):
        with open(self.log, "a") as f:
            f.write(json.dumps({"rank": self.rank, "what": what, "t": time.perf_counter(),
                                "group_alive": dist.is_initialized()}) + "\n")

    def warm(self, steps):
        self._say(f"warm {steps}")

    def run_steps(self, k):
        self._say(f"run_steps {k}")
        time.sleep(0.01 * k * (1 + self.rank))                  # rank r is (1 + r) times slower: the max is the last rank's
        clip0 = self.rank * k * 4
        self.rows = torch.arange(clip0, clip0 + k * 4, dtype=torch.float32).view(-1, 1, 1) * torch.ones(1, 3, 2)

    def sync(self):
        pass

    def scalar(self, x):
        return torch.tensor([x], dtype=torch.float64)

    def gather(self):
        from talkshow_amd.parallel import gather_sequences
        self._say("gather")
        allp = gather_sequences(self.rows)
        assert torch.equal(allp[:, 0, 0], torch.arange(allp.shape[0], dtype=torch.float32))      # clip k lands at row k
        return {"gather_bytes_per_rank": int(self.rows.numel() * 4), "gathered_shape": list(allp.shape)}

    def frames_per_step(self):
        return 4 * 300

    def describe(self):
        return {"workload": "stub"}

    def selfcheck(self):
        self._say("selfcheck")
        if self.rank == self.fail_rank:
            raise RuntimeError(f"selfcheck: rank {self.rank} found a difference")
        return {"selfcheck": "ok"}

    def extras(self, out):
        self._say("extras")
        time.sleep(0.5)                                          # long enough that a rank waiting in a collective would show, also on a busy host
        out["roofline"] = {"frac": 0.0}


def _worker(rank, world, port, steps, warmup, tmp):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    job = StubJob(world, rank, os.path.join(tmp, f"log{rank}.jsonl"))
    dt, t_compute, info = bench.run_contract(job, dist, world, rank, steps, warmup)
    lines = []
    bench.finish(job, dist, world, rank, steps, warmup, dt, info, emit=lines.append)
    with open(os.path.join(tmp, f"out{rank}.json"), "w") as f:
        json.dump({"lines": lines, "dt": dt, "t_compute": t_compute, "end": time.perf_counter(),
                   "group_alive_at_exit": dist.is_initialized()}, f)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_bench_contract_control_flow(tmp_path, world):
    steps, warmup = 5, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, steps, warmup, str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(tmp_path / f"out{r}.json")) for r in range(world)]
    logs = [[json.loads(l) for l in open(tmp_path / f"log{r}.jsonl")] for r in range(world)]
    # one JSON line, from rank 0 only
    assert len(outs[0]["lines"]) == 1 and all(o["lines"] == [] for o in outs[1:])
    line = json.loads(outs[0]["lines"][0])
    assert line["n_gpus"] == world and line["steps"] == steps and line["warmup"] == warmup and line["selfcheck"] == "ok"
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None
    # max over ranks: every rank reports the same time, at least the slowest rank's compute time
    assert len({round(o["dt"], 9) for o in outs}) == 1
    assert outs[0]["dt"] >= max(o["t_compute"] for o in outs) and outs[0]["dt"] >= 0.01 * steps * world
    assert abs(line["value"] - world * steps * 4 * 300 / outs[0]["dt"]) < 1e-6 * line["value"]
    assert abs(line["ms_per_step"] - outs[0]["dt"] / steps * 1e3) < 1e-9
    if world > 1:
        assert line["rccl"]["ranks_seen"] == world and line["rccl"]["gathered_shape"] == [world * steps * 4, 3, 2]
    else:
        assert line["rccl"] is None
    for r, log in enumerate(logs):
        what = [e["what"] for e in log]
        want = [f"warm {steps}", f"run_steps {warmup}"] + (["gather"] if world > 1 else []) + [f"run_steps {steps}"] \
            + (["gather"] if world > 1 else []) + ["selfcheck"] + (["extras"] if r == 0 else [])
        assert what == want, (r, what)
        assert not outs[r]["group_alive_at_exit"]
    # rank 0's extras run with no process group alive, and no other rank waits for them: the others are done before they end
    ex = [e for e in logs[0] if e["what"] == "extras"][0]
    assert ex["group_alive"] is False
    for r in range(1, world):
        assert outs[r]["end"] < outs[0]["end"] - 0.25, "a rank other than 0 was still around while rank 0 ran its extras"


def _failing_worker(rank, world, port, tmp):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    job = StubJob(world, rank, os.path.join(tmp, f"log{rank}.jsonl"))
    job.fail_rank = 1
    dt, _, info = bench.run_contract(job, dist, world, rank, 2, 1)
    t0 = time.perf_counter()
    try:
        bench.finish(job, dist, world, rank, 2, 1, dt, info, emit=lambda line: None)
        outcome = "returned"
    except RuntimeError as e:
        outcome = str(e)
    with open(os.path.join(tmp, f"fail{rank}.json"), "w") as f:
        json.dump({"outcome": outcome, "seconds": time.perf_counter() - t0, "group_alive": dist.is_initialized()}, f)


def test_a_failed_selfcheck_does_not_strand_the_other_ranks(tmp_path):
    """`finish` used to raise on the failing rank before the barrier, leaving the others in it until the RCCL time-out.
    Now the failure is all-reduced, every rank leaves the process group, and every rank raises — at once."""
    world = 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_failing_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(tmp_path / f"fail{r}.json")) for r in range(world)]
    assert "rank 1 found a difference" in outs[1]["outcome"]
    assert all("selfcheck failed on another rank" in outs[r]["outcome"] for r in (0, 2))
    assert all(o["seconds"] < 20 and not o["group_alive"] for o in outs)


def test_collectives_can_be_forced_at_world_one(tmp_path):
    """TS_BENCH_FORCE_COLLECTIVES=1 (tools/rccl_smoke.sh): the N > 1 code path at world 1 — here on gloo."""
    import bench
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        job = StubJob(1, 0, str(tmp_path / "log.jsonl"))
        dt, _, info = bench.run_contract(job, dist, 1, 0, 3, 1, collectives=True)
        lines = []
        bench.finish(job, dist, 1, 0, 3, 1, dt, info, emit=lines.append, collectives=True)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    line = json.loads(lines[0])
    assert line["rccl"]["ranks_seen"] == 1 and line["rccl"]["gathered_shape"] == [12, 3, 2] and line["n_gpus"] == 1
    what = [json.loads(l)["what"] for l in open(tmp_path / "log.jsonl")]
    assert what.count("gather") == 2


def _repeat_worker(rank, world, port, tmp):
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    job = StubJob(world, rank, os.path.join(tmp, f"log{rank}.jsonl"))
    job.regions = []
    job.region_begin = lambda: job.regions.append("begin")
    job.region_end = lambda: job.regions.append("end")
    # the second of the three regions is slow on every rank: the median must not be it, nor the fastest
    slow = iter([1.0, 1.0, 4.0, 2.0])                                  # warm-up run, then regions 0, 1, 2
    plain = job.run_steps
    job.run_steps = lambda k: (time.sleep(0.02 * next(slow)), plain(k))[1]
    dt, _, info = bench.run_contract(job, dist, world, rank, 3, 1, repeats=3)
    lines = []
    bench.finish(job, dist, world, rank, 3, 1, dt, info, emit=lines.append)
    with open(os.path.join(tmp, f"rep{rank}.json"), "w") as f:
        json.dump({"lines": lines, "dt": dt, "runs": job.contract_runs, "regions": job.regions}, f)


@pytest.mark.parametrize("world", [1, 2])
def test_repeated_timed_regions_report_the_median(tmp_path, world):
    """`--repeats 3` = three bracketed regions of exactly `steps` steps in the same warm state; `value` is the
    median region's, every region's time rides along in `runs_ms`, each region has its own barriers / exchange / MAX over ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_repeat_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(tmp_path / f"rep{r}.json")) for r in range(world)]
    line = json.loads(outs[0]["lines"][0])
    runs = outs[0]["runs"]["runs_ms"]
    assert len(runs) == 3 and line["runs_ms"] == runs and outs[0]["regions"] == ["begin", "end"] * 3
    assert runs[1] > runs[2] > runs[0]                                   # 4x, 2x, 1x of the extra sleep
    assert outs[0]["runs"]["median_index"] == 2 and abs(outs[0]["dt"] * 1e3 - runs[2]) < 1e-9
    assert abs(line["ms_per_step"] - runs[2] / 3) < 1e-9 and abs(line["value"] - world * 3 * 4 * 300 / outs[0]["dt"]) < 1e-6 * line["value"]
    assert 0 <= line["host_cpu_s"] and line["runs_spread"] > 0.2
    assert all(o["runs"]["runs_ms"] == runs and o["dt"] == outs[0]["dt"] for o in outs) if world > 1 else True   # MAX over ranks per region
    for r in range(world):
        what = [json.loads(l)["what"] for l in open(tmp_path / f"log{r}.jsonl")]
        assert what.count("run_steps 3") == 3 and what.count("gather") == (4 if world > 1 else 0)


def test_bench_spawns_its_own_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus N` with WORLD_SIZE unset used to die on an assert.  `needs_spawn` / `spawn_ranks`
    start the N ranks through torch.distributed.run (the driver's launcher) — here a probe script on gloo at world 2 stands in for
    bench.py's GPU body: both ranks come up with the launcher's environment, rendezvous on 127.0.0.1 and all-reduce."""
    import sys
    import bench
    assert bench.needs_spawn({}, 8) and bench.needs_spawn({"PATH": "x"}, 2)
    assert not bench.needs_spawn({}, 1) and not bench.needs_spawn({"WORLD_SIZE": "8", "RANK": "3"}, 8)
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import json, os, sys, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(dist.get_rank() + 1)]); dist.all_reduce(t)\n"
        "json.dump({'rank': dist.get_rank(), 'world': dist.get_world_size(), 'sum': t.item(), 'args': sys.argv[1:],\n"
        "           'spawned': os.environ.get('TS_BENCH_SPAWNED'), 'addr': os.environ['MASTER_ADDR']},\n"
        "          open(os.path.join(sys.argv[1], f'probe{dist.get_rank()}.json'), 'w'))\n"
        "dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    rc = bench.spawn_ranks(2, str(probe), [str(tmp_path), "--steps", "20"], env=env, timeout=240)
    assert rc == 0
    got = [json.load(open(tmp_path / f"probe{r}.json")) for r in range(2)]
    assert [g["rank"] for g in got] == [0, 1] and all(g["world"] == 2 and g["sum"] == 3.0 and g["spawned"] == "1" for g in got)
    assert all(g["addr"] == "127.0.0.1" and g["args"] == [str(tmp_path), "--steps", "20"] for g in got)
    # and a failing rank's exit code comes back
    bad = tmp_path / "bad.py"
    bad.write_text("import sys; sys.exit(3)\n")
    assert bench.spawn_ranks(2, str(bad), [], env=env, timeout=240) != 0
