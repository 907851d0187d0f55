"""The N>1 path on the HIP library: two ranks, each driving `Engine` on its contiguous shard of the batch, exchange
only the [W,5] convergence table.  A gpurun box has ONE device, so both ranks use cuda:0 and the table travels over gloo
(RCCL refuses two ranks on one device); on a multi-GPU node the same code runs with backend "nccl" and one device per rank
(`bench.py --gpus N`).  Trajectories must equal the single-process HIP run bit for bit (slots never interact) and the table
must hold every rank's row."""
import os
import socket
import sys

import numpy as np
import pytest

import isls_problems as P

pytestmark = pytest.mark.gpu

B_GLOBAL, HORIZON, OUTER, L, J = 45, 40, 3, 12, 3


def _solve_shard(lo, hi, table, rank):
    """`OUTER` outer DP-form iLQR-ADMM iterations of the trajectories [lo, hi) of config 2 through the C driver, exactly the
    step bench.py times; the reduction of the last iteration lands in row `rank` of `table`."""
    import torch

    from isls import models
    from isls.engine import Engine
    cfg = P.config2(batch=B_GLOBAL, N=HORIZON, seed=0)
    n, m, b = cfg["n"], cfg["m"], hi - lo
    eng = Engine(b, HORIZON, n, m, dtype=torch.float64, device="cuda:0")
    mdl = models.LTI(cfg["A"], cfg["B"])
    eng.set_model(mdl.model_id, mdl.params())
    eng.set_quadratic_cost(cfg["zs"][lo:hi], cfg["Qs"], cfg["seq"], cfg["u_std"])
    eng.set_nominal(np.repeat(cfg["x0"][lo:hi, None, :], HORIZON, axis=1), cfg["u0"][lo:hi])
    eng.set_admm(rho_u=cfg["rho_u"], u_box=(cfg["u_lo"], cfg["u_hi"]), relax=cfg["relax"])
    eng.build_outer(L, J, tol_abs=0.0, tol_rel=0.0, ff_nseg=2)
    for _ in range(OUTER):
        eng.linearize()
        eng.expand()
        eng.run_outer()
        eng.accept_x_step()
        eng.reduce(table=table, rank=rank)
    torch.cuda.synchronize()
    return eng


def _rank_main(rank, world, port, ret):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "ilqr-admm_amd"), root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    from isls.shard import TableExchange, allreduce_table, shard_range, summarize
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(B_GLOBAL, rank, world)
    table = torch.full((world, 5), 7.0, dtype=torch.float64, device="cuda:0")     # stale contents must not survive
    eng = _solve_shard(lo, hi, table, rank)
    own = table.cpu()
    gathered = allreduce_table(own.clone(), world)           # gloo: the 40-byte-per-rank all-reduce, staged through the host
    # the same exchange the way bench.py posts it: asynchronous, on the device table, rotating buffers
    xch = TableExchange(world, rank, torch.float64, "cuda:0")
    for _ in range(3):
        xch.post(lambda t, r: eng.reduce(table=t, rank=r))
    assert torch.equal(xch.latest(lag=1).cpu(), gathered) and torch.equal(xch.finish().cpu(), gathered)
    ret[rank] = dict(x=eng.xhat.cpu().numpy(), u=eng.uhat.cpu().numpy(), cost=eng.cost.cpu().numpy(),
                     own=own.numpy(), table=gathered.numpy(), total=summarize(gathered).numpy())
    dist.destroy_process_group()


def test_two_ranks_on_the_hip_path_match_single_process():
    import torch
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_main, args=(world, port, ret), nprocs=world, join=True)
    # the same batch in one process
    one = torch.zeros(1, 5, dtype=torch.float64, device="cuda:0")
    eng = _solve_shard(0, B_GLOBAL, one, 0)
    x = np.concatenate([ret[r]["x"] for r in range(world)])
    u = np.concatenate([ret[r]["u"] for r in range(world)])
    cost = np.concatenate([ret[r]["cost"] for r in range(world)])
    assert np.array_equal(x, eng.xhat.cpu().numpy()) and np.array_equal(u, eng.uhat.cpu().numpy())
    assert np.array_equal(cost, eng.cost.cpu().numpy())
    full = one.cpu().numpy()[0]
    for r in range(world):
        own, table, total = ret[r]["own"], ret[r]["table"], ret[r]["total"]
        assert np.all(own[1 - r] == 0.0) and own[r, 3] == ret[r]["x"].shape[0]      # other row zeroed, own row = shard
        assert np.array_equal(table, ret[0]["table"])                                # every rank sees the same table
        assert np.array_equal(table[r], own[r])
        assert abs(total[0] - full[0]) <= 1e-12 * abs(full[0])                       # sum of shard sums vs one sum
        assert np.array_equal(total[1:], full[1:])                                   # maxima and counts are exact


def test_bench_two_rank_rehearsal():
    """`bench.py --gpus 2` end to end on the one device of the test box (ISLS_BENCH_REHEARSAL=1: both ranks on cuda:0, the
    table over gloo): the launcher spawns its ranks as child processes, every rank runs the timed loop with the asynchronous
    table exchange, rank 0 prints ONE line that says n_gpus = 2 and carries both shards in its convergence summary."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ISLS_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "300",
                        "--horizon", "40"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rehearsal"] is True and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["convergence"]["active"] == 600.0 and out["convergence"]["failed"] == 0.0
    # same keys as the N = 1 line: the CPU baseline is timed at N = 1 only and is null here; the collective layer reports what it saw
    assert out["value"] > 0 and out["cpu_baseline"] is None
    assert out["n_ranks_joined"] == 2 and out["collective_backend"] == "gloo"
