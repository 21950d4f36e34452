"""The N > 1 path on CPU: two gloo processes shard the images round-robin and all-gather the
per-image (bpp, psnr, mse) rows -- the only exchange the path has."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_items, q):
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as graft
    graft.load_package()
    from shallow_ntc_amd import distributed as D
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    idx = D.shard_indices(num_items, rank, world)
    rows = np.array([[i + 0.25, 30.0 + i, 100.0 - i] for i in idx], np.float64).reshape(len(idx), 3)
    table = D.gather_rows(rows, idx, num_items)
    t = D.max_over_ranks(1.0 + rank)
    assert D.all_ranks(1.0 + rank) == [1.0 + r for r in range(world)]      # every rank's own step time, in rank order, on every rank
    D.barrier()
    q.put((rank, table, t, idx))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_gather_rows():
    world, num_items = 2, 7          # ragged: rank 0 holds 4 images, rank 1 holds 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.array([[i + 0.25, 30.0 + i, 100.0 - i] for i in range(num_items)])
    seen = []
    for rank, table, t, idx in outs:
        np.testing.assert_array_equal(table, want)     # every rank ends with the full table, ordered by image id
        assert t == 2.0                                # max over ranks
        seen += idx
    assert sorted(seen) == list(range(num_items))      # a partition: no image decoded twice or dropped


def test_single_process_paths():
    sys.path.insert(0, str(ROOT))
    from shallow_ntc_amd import distributed as D
    assert D.shard_indices(24, 3, 8) == [3, 11, 19]
    assert D.shard_indices(0, 0, 2) == []
    t = D.gather_rows(np.array([[1.0, 2.0]]), [1], 3)
    assert np.isnan(t[0]).all() and (t[1] == [1.0, 2.0]).all()
    assert D.max_over_ranks(3.5) == 3.5


def _reducer_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import torch
    import __graft_entry__ as graft
    graft.load_package()
    from shallow_ntc_amd import distributed as D
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init(backend="gloo")
    flat = torch.arange(40, dtype=torch.float32) * (rank + 1)          # rank r holds (r + 1) * [0..39]
    slices = dict(synthesis=(0, 12), prior=(12, 12), hyper=(12, 28), analysis=(28, 40))   # one empty bucket
    red = D.BucketReducer(flat, slices)
    for name in slices:                                                   # backward order
        red.launch(name)
    scale = red.finish()
    q.put((rank, (flat * scale).numpy(), scale))
    red2 = D.BucketReducer(flat, slices)
    red2.launch("synthesis")
    try:
        red2.finish()
        q.put((rank, "no error", None))
    except RuntimeError as e:
        q.put((rank, str(e), None))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_gradient_all_reduce():
    """The training step's exchange: every bucket of the flat gradient buffer is summed across ranks, the returned
    factor turns sums into means, and forgetting a bucket is an error rather than a silently stale gradient."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(40, dtype=np.float32) * 1.5                       # mean of 1x and 2x
    for rank, val, scale in outs:
        if scale is not None:
            assert scale == 0.5
            np.testing.assert_array_equal(val, want)
        else:
            assert "never launched" in val


def _run_bench(extra_env, *argv):
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` outside torchrun spawns its own ranks (the parent never touches the GPU), the ranks
    rendezvous over gloo, run the path's collectives, and rank 0's line comes back through the parent with the world in it."""
    import json
    r = _run_bench(dict(SNTC_DIST_BACKEND="gloo"), "--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["value"] is None
    assert line["rccl"]["backend"] == "gloo" and line["rccl"]["world"] == 2
    assert sorted(d["rank"] for d in line["rccl"]["devices"]) == [0, 1]
    assert len({d["host_pid"] for d in line["rccl"]["devices"]}) == 2          # two processes, not one


def test_bench_self_launches_eight_ranks():
    """The driver's widest launch (`--gpus 8`: one process per GPU of the node) on the CPU: eight gloo ranks rendezvous, run
    the path's collectives incl. the schedule broadcast from rank 0, and rank 0's line names all eight."""
    import json
    r = _run_bench(dict(SNTC_DIST_BACKEND="gloo", OMP_NUM_THREADS="1"), "--gpus", "8", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["rccl"]["world"] == 8
    assert sorted(d["rank"] for d in line["rccl"]["devices"]) == list(range(8))
    assert len({d["host_pid"] for d in line["rccl"]["devices"]}) == 8


def test_bench_launcher_propagates_a_failing_rank():
    r = _run_bench(dict(SNTC_DIST_BACKEND="gloo", SNTC_LAUNCH_CHECK_FAIL_RANK="1"), "--gpus", "2", "--launch-check")
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr


def _units_worker(rank, world, port, num_units, q):
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as graft
    graft.load_package()
    from shallow_ntc_amd import distributed as D
    if world > 1:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        D.init(backend="gloo")
    calls = []

    def fn(u):                       # a unit's row depends on the unit alone (as an image's metrics do)
        calls.append(u)
        lam, img = divmod(u, 5)
        return [0.1 * lam + 0.01 * img, 30.0 - img, float(u) ** 0.5, lam]

    table = D.run_units(num_units, fn)
    q.put((rank, table, calls))
    D.shutdown()


def test_unit_dealing_gives_the_same_table_for_any_world_size():
    """configs[3] / configs[4] drivers (tools/rd_sweep.py, tools/itinf_sweep.py) are `run_units` + a GPU function per unit:
    1 rank and 2 ranks must produce the same table, every unit computed exactly once, incl. a world larger than the
    number of units (a rank with nothing to do still takes part in the gather)."""
    ctx = mp.get_context("spawn")
    tables = {}
    for world, num_units in ((1, 13), (2, 13), (2, 1), (8, 13), (8, 5)):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_units_worker, args=(r, world, port, num_units, q)) for r in range(world)]
        for p in procs:
            p.start()
        outs = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        seen = sorted(u for _, _, calls in outs for u in calls)
        assert seen == list(range(num_units))
        for _, table, _ in outs:
            assert table.shape == (num_units, 4) and not np.isnan(table).any()
            np.testing.assert_array_equal(table, outs[0][1])
        tables[(world, num_units)] = outs[0][1]
    np.testing.assert_array_equal(tables[(1, 13)], tables[(2, 13)])
    np.testing.assert_array_equal(tables[(1, 13)], tables[(8, 13)])          # the node's eight ranks: same table again
