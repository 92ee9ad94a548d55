"""ts_allreduce_* (csrc/ts_collective.hip) on one MI355X: a world-1 RCCL communicator through the C ABI.  The multi-rank
behaviour of the exchange (scaling, ordering, identical replicas) is covered on CPU by the world-2 gloo tests
(tests/test_dp_gloo.py, test_dp_dqn_gloo.py, test_dp_sac_gloo.py), which accept the same `allreduce=` hook."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_allreduce_world_one_through_the_c_abi():
    from tianshou_amd import _lib

    lib = _lib.load()
    uid = (C.c_uint8 * 128)()
    _lib.check(lib.ts_allreduce_unique_id(uid))
    assert any(uid)                                            # RCCL filled the id
    comm = C.c_void_p()
    _lib.check(lib.ts_allreduce_init(uid, _lib.i64(0), _lib.i64(1), C.c_int(0), C.byref(comm)))
    x = torch.randn(1_000_003, device="cuda")
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                              # issued on the caller's stream, whichever it is
        _lib.check(lib.ts_allreduce(comm, _lib.ptr(x), _lib.i64(x.numel()), _lib.current_stream(x.device)))
        y = x * 2
    side.synchronize()
    assert torch.equal(x, ref) and torch.equal(y, ref * 2)     # sum over one rank
    _lib.check(lib.ts_allreduce(comm, None, _lib.i64(0), _lib.current_stream(x.device)))      # empty exchange
    with pytest.raises(_lib.EngineError):
        _lib.check(lib.ts_allreduce(None, _lib.ptr(x), _lib.i64(4), _lib.current_stream(x.device)))
    with pytest.raises(_lib.EngineError):
        _lib.check(lib.ts_allreduce_init(uid, _lib.i64(2), _lib.i64(2), C.c_int(0), C.byref(C.c_void_p())))
    _lib.check(lib.ts_allreduce_destroy(comm))


def test_native_allreduce_as_the_exchange_of_the_dp_wrappers():
    """NativeAllReduce without a process group = world 1; DataParallelSAC takes it as `allreduce=`."""
    from tianshou_amd.collective import NativeAllReduce
    from tianshou_amd.distributed import DataParallelSAC

    ar = NativeAllReduce(torch.device("cuda", 0))
    assert (ar.rank, ar.world) == (0, 1)
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    assert torch.equal(ar(x), torch.arange(1000, dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        ar(torch.zeros(4, dtype=torch.float64, device="cuda"))
    dp = DataParallelSAC(engine=None, allreduce=ar)
    dp.world = 2                                               # force the exchange path on one rank: sum, then * 1/2
    buf = torch.full((10,), 3.0, device="cuda")
    dp._reduce(buf)
    assert torch.equal(buf, torch.full((10,), 1.5, device="cuda"))
    ar.close()


# --------------------------------------------------------------------------------------------------------------------
# one-shot all-reduce over HIP-IPC-mapped peer buffers: two PROCESSES sharing the one GPU of the test box
# --------------------------------------------------------------------------------------------------------------------
def _one_shot_worker(rank, world, port, sizes, iters, out_q):
    import os
    import time

    import numpy as np
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("TS_SMALL_ALLREDUCE_SPINS", "4000000")       # a peer that never arrives fails in seconds, not minutes
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tianshou_amd.collective import NativeAllReduce

        torch.cuda.set_device(0)
        ar = NativeAllReduce(torch.device("cuda", 0), rccl=False)       # same GPU: no RCCL communicator possible
        assert ar.small_capacity == 16384 and (ar.rank, ar.world) == (rank, world)
        bad = 0
        rng = np.random.default_rng(100 + rank)
        load = torch.randn(2048, 2048, device="cuda")
        for it in range(iters):
            n = sizes[it % len(sizes)]
            g = torch.Generator().manual_seed(1000 * it + rank)
            mine = torch.randn(n, generator=g)
            parts = [torch.empty(n) for _ in range(world)]
            dist.all_gather(parts, mine)                                # the reference sum, rank order, on the host
            want = parts[0].clone()
            for p in parts[1:]:
                want += p
            if rng.random() < 0.5:                                       # uneven load: one rank arrives late / under load
                time.sleep(float(rng.random()) * 2e-3)
                load = load @ load * 1e-3
            x = mine.cuda()
            ar(x)
            if rng.random() < 0.3:
                load = load @ load * 1e-3
            got = x.cpu()
            bad += 0 if torch.equal(got, want) else 1                    # every word, bit for bit
        ar.check()
        ar.close()
        out_q.put((rank, bad))
    finally:
        dist.destroy_process_group()


def test_one_shot_allreduce_between_two_processes_on_one_gpu():
    """ts_allreduce_small_*: IPC handles exchanged over gloo, 300 calls of mixed sizes under uneven load, every word compared
    with the host sum (for two ranks a + b is the same float whatever the order, so this is also what RCCL returns)."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    sizes = [11089, 1, 16384, 7, 4096, 11085 + 4, 2]
    procs = [ctx.Process(target=_one_shot_worker, args=(r, 2, port, sizes, 300, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, "a rank hung"
    assert all(p.exitcode == 0 for p in procs)
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: 0, 1: 0}


# --------------------------------------------------------------------------------------------------------------------
# DataParallelPPO between two PROCESSES on the one GPU against the single-process update, at the full C2 minibatch
# --------------------------------------------------------------------------------------------------------------------
DP_OBS, DP_ACT, DP_N, DP_MB, DP_REPEAT = 17, 6, 2 * 65536, 65536, 2


def _dp_problem():
    """The same synthetic batch, weights and per-rank permutations in every process (seeded)."""
    import numpy as np
    import torch

    from oracle import oracle_ppo as OP

    rng = np.random.default_rng(42)
    data = dict(obs=rng.standard_normal((DP_N, DP_OBS), dtype=np.float32), act=rng.standard_normal((DP_N, DP_ACT), dtype=np.float32) * 0.5,
                adv=rng.standard_normal(DP_N, dtype=np.float32), returns=rng.standard_normal(DP_N, dtype=np.float32),
                logp_old=(rng.standard_normal(DP_N, dtype=np.float32) * 0.1 - 5.0), v_s=rng.standard_normal(DP_N, dtype=np.float32) * 0.3)
    flat = OP.flatten_params(OP.init_params(DP_OBS, DP_ACT, seed=9))
    half = DP_N // 2
    perms = [[np.random.default_rng(100 * r + rep).permutation(half) for rep in range(DP_REPEAT)] for r in range(2)]
    return data, flat, perms


def _dp_cfg():
    from tianshou_amd import ppo as P

    return P.PPOConfig(eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, value_clip=True, advantage_normalization=True,
                       return_scaling=False, lr=3e-4)


def _dp_worker(rank, world, port, out_q):
    import os

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("TS_SMALL_ALLREDUCE_SPINS", "40000000")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tianshou_amd import ppo as P
        from tianshou_amd.collective import NativeAllReduce
        from tianshou_amd.distributed import DataParallelPPO

        torch.cuda.set_device(0)
        data, flat, perms = _dp_problem()
        half = DP_N // 2
        lo = rank * half
        b = {k: torch.as_tensor(v[lo:lo + half]).cuda().contiguous() for k, v in data.items()}
        eng = P.PPOEngine(DP_OBS, DP_ACT, flat.cuda(), _dp_cfg())
        ar = NativeAllReduce(torch.device("cuda", 0), rccl=False)          # one device: the one-shot IPC exchange
        dp = DataParallelPPO(eng, allreduce=ar)
        losses, steps = dp.update(b, DP_MB // world, DP_REPEAT, perms[rank])
        torch.cuda.synchronize()
        ar.check()
        out_q.put((rank, steps, losses.cpu().numpy(), eng.params.cpu().numpy(), eng.adam_m.cpu().numpy()))
        dist.barrier()
        ar.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_update_on_one_gpu_equals_the_single_process_update():
    """BASELINE configs[3] as far as one GPU can show it: two processes share the device, each holds half of
    a 131,072-row batch and takes 32,768 rows of every 65,536-row GLOBAL minibatch (`ppo_step2_kernel` launches, per-minibatch
    advantage statistics and the gradient summed over the ranks through `ts_ppo_dp_step`'s one-shot IPC all-reduce); the single
    process runs the same four minibatches -- the union of the two ranks' rows -- through `ts_ppo_update`.  Same global batches,
    same weights: losses agree to 1e-6, parameters and Adam moments to the rounding of a differently partitioned fp32 sum, and
    the two replicas are bit-identical."""
    import socket

    import numpy as np
    import torch.multiprocessing as mp

    from tianshou_amd import ppo as P

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(120)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive and all(p.exitcode == 0 for p in procs)
    # the two replicas: identical losses, parameters and moments, bit for bit
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(a, b)
    # single process on the union batches
    data, flat, perms = _dp_problem()
    half, mb = DP_N // 2, DP_MB // 2
    eng = P.PPOEngine(DP_OBS, DP_ACT, flat.cuda(), _dp_cfg())
    b = {k: torch.as_tensor(v).cuda().contiguous() for k, v in data.items()}
    union = [np.concatenate([np.concatenate([perms[0][rep][c * mb:(c + 1) * mb], perms[1][rep][c * mb:(c + 1) * mb] + half])
                             for c in range(half // mb)]) for rep in range(DP_REPEAT)]
    assert all(len(u) == DP_N and len(np.unique(u)) == DP_N for u in union)
    losses, steps = eng.update(b, DP_MB, DP_REPEAT, union)
    torch.cuda.synchronize()
    assert steps == res[0][0] == DP_REPEAT * (DP_N // DP_MB)
    np.testing.assert_allclose(res[0][1], losses.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(res[0][2], eng.params.cpu().numpy(), rtol=1e-5, atol=0.02 * 3e-4)
    ref_m = eng.adam_m.cpu().numpy()
    np.testing.assert_allclose(res[0][3], ref_m, rtol=1e-4, atol=1e-6 * float(np.abs(ref_m).max()))


def _bench_dry_run(cmd_tail, launcher, timeout):
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TS_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(root, "bench.py"), *cmd_tail]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_dry_run_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, two ranks), with TS_BENCH_ONE_GPU=1: both ranks on
    cuda:0, rendezvous over gloo, the gradient exchange of every minibatch step on the one-shot IPC all-reduce inside
    ts_ppo_dp_step.  Checks the N > 1 code path end to end (one rollout sharded by env id, shard-local preprocessing with
    global return statistics, the data-parallel update on 32,768-row local minibatches, max-over-ranks timing, the weak leg
    beside it, one JSON line from rank 0) -- not a measurement."""
    d = _bench_dry_run(["--gpus", "2", "--steps", "1", "--warmup", "1"], launcher=2, timeout=600)
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["gradient_steps_per_step"] == 160 and d["config"]["transitions_per_step"] == 1 << 20
    assert d["config"]["parallelism"] == "dp2" and "one-shot" in d["config"]["exchange"]
    assert "configs[3]" in d["config"]["workload"] and "32768 rows per rank" in d["config"]["workload"]
    assert d["weak_scaling"]["scaling"] == "weak" and d["weak_scaling"]["value"] > 0
    assert d["exchange_us"] > 0 and d["exchange_ranks"] == {"communicator": 2, "rccl_reported": 0}     # one-shot only here
    assert d["roofline"]["kernel"] == "ppo_step2_kernel" and 0.0 < d["roofline"]["frac"] < 1.0
    assert d["roofline"]["rows_per_launch"] == 32768


def test_bench_eight_ranks_spawns_itself_on_one_gpu():
    """`python bench.py --gpus 8` without a launcher: bench.py starts torch.distributed.run itself.  Eight IPC peers on one
    device exercise the 8-way rank-ordered sum of the one-shot all-reduce and the 64-sub-buffer shards / 8,192-row local
    minibatches of BASELINE configs[3] (strong leg only, to bound the time eight time-sharing ranks need)."""
    d = _bench_dry_run(["--gpus", "8", "--steps", "1", "--warmup", "1", "--scaling", "strong"], launcher=0, timeout=900)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["gradient_steps_per_step"] == 160 and "8192 rows per rank" in d["config"]["workload"]
    assert "64 sub-buffers per rank" in d["config"]["workload"]
    assert d["exchange_ranks"]["communicator"] == 8 and d["exchange_us"] > 0
    assert all(abs(x) < 1e3 for x in d["final_losses"])
