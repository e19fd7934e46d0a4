"""World-size-2 gloo test (CPU) of the frame-sharded data-parallel plumbing (fsgs_amd/dist.py,
SURVEY.md s8e): one all-reduce over the flat gradient bucket, identical parameters after the step,
densification statistics reduced as (SUM, SUM, MAX)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fsgs_amd.model import PARAM_NAMES, GaussianCloud


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(P, seed):
    rng = np.random.default_rng(seed)
    return {"_xyz": rng.standard_normal((P, 3)), "_features_dc": rng.standard_normal((P, 1, 3)),
            "_features_rest": rng.standard_normal((P, 15, 3)), "_opacity": rng.standard_normal((P, 1)),
            "_scaling": rng.standard_normal((P, 3)), "_rotation": rng.standard_normal((P, 4))}


def _worker(rank, world, port, use_bucket, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fsgs_amd import dist as fdist

    r, w, _ = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    P = 257
    pc = GaussianCloud(_params(P, 0), device="cpu")  # identical replicas
    pc.training_setup(fused=False)  # torch Adam on CPU: only the collective plumbing is under test here
    bucket = fdist.GradBucket(pc) if use_bucket else None
    for step in range(3):
        if bucket is not None:
            bucket.attach(pc)
        # each rank "renders its own camera": a rank-dependent loss over the shared cloud
        loss = sum(((pc.params[k] * (rank + 1 + step)) ** 2).sum() * (0.1 + i) for i, k in enumerate(PARAM_NAMES))
        loss.backward()
        fdist.sync_gradients(pc, bucket)
        # the all-reduced gradient equals the sum of both ranks' gradients
        expect = sum(2 * (q + 1 + step) ** 2 for q in range(world))
        g = pc.params["_xyz"].grad
        assert torch.allclose(g, pc.params["_xyz"].detach() * expect * 0.1, rtol=1e-5, atol=1e-6)
        pc.optimizer.step()
        if bucket is None:
            pc.optimizer.zero_grad(set_to_none=True)
    pc.variables["xyz_gradient_accum"] += rank + 1.0
    pc.variables["denom"] += 1.0
    pc.variables["max_radii2D"] += float(rank)
    fdist.sync_densification_stats(pc)
    assert torch.all(pc.variables["xyz_gradient_accum"] == 3.0)
    assert torch.all(pc.variables["denom"] == 2.0)
    assert torch.all(pc.variables["max_radii2D"] == 1.0)
    torch.save({k: pc.params[k].detach() for k in PARAM_NAMES}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_bucket", [True, False])
def test_two_ranks_stay_in_lockstep(tmp_path, use_bucket):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, use_bucket, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "rank0.pt"))
    b = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for k in PARAM_NAMES:
        assert torch.equal(a[k], b[k]), k  # replicas must be bit-identical after synchronised steps


def _compact_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fsgs_amd import dist as fdist

    fdist.init_from_env(backend="gloo")
    P = 1000
    base = torch.arange(P * 14, dtype=torch.float32).reshape(P, 14)
    # one collective
    gc = base * (rank + 1)
    fdist.all_reduce_compact(gc)
    assert torch.equal(gc, base * 3)
    # chunked + pipelined: every row is reduced exactly once BEFORE its Adam chunk runs, chunk bounds are multiples of 256
    gc = base * (rank + 1)
    seen = []

    def adam_rows(lo, hi):
        assert lo % 256 == 0 and (hi % 256 == 0 or hi == P)
        assert torch.equal(gc[lo:hi], base[lo:hi] * 3)
        seen.append((lo, hi))

    fdist.PipelinedCompactReducer(3)(gc, adam_rows)
    assert seen[0][0] == 0 and seen[-1][1] == P and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    # producer-side: chunks are handed over one by one as they are "produced" (rows beyond the produced ones still hold
    # this rank's partial sums); Adam runs per chunk, in production order, only on rows whose all-reduce is done
    gc = base * (rank + 1)
    seen.clear()
    red = fdist.ProducerPipelinedReducer(3)
    bounds = red.bounds(P)
    assert bounds[0][0] == 0 and bounds[-1][1] == P and all(lo % 256 == 0 for lo, _ in bounds) and len(bounds) == 2
    for n, (lo, hi) in enumerate(bounds):
        red.produced(gc, lo, hi)
        assert torch.equal(gc[lo:hi], base[lo:hi] * 3)
        if n + 1 < len(bounds):
            nlo, nhi = bounds[n + 1]
            assert torch.equal(gc[nlo:nhi], base[nlo:nhi] * (rank + 1))  # not yet exchanged
    red.finish(adam_rows)
    assert seen == bounds and red._pending == []
    # and its consumer-side form (several views per step)
    gc = base * (rank + 1)
    seen.clear()
    fdist.ProducerPipelinedReducer(2)(gc, adam_rows)
    assert [s[0] for s in seen] == [0, 512] and seen[-1][1] == P
    # a step that died between produced() and finish() (an FsgsError in a later chunk) must not leak its chunks into the
    # next step -- after a densification they may lie past the cloud -- and an empty cloud still yields one (0, 0) chunk
    # (the backward call that writes the step's loss), which neither exchanges nor steps anything
    red = fdist.ProducerPipelinedReducer(2)
    red.bounds(P)
    red.produced(gc, 0, 512)
    assert len(red._pending) == 1
    small = base[:300] * (rank + 1)
    seen.clear()
    P_small = 300
    for lo, hi in red.bounds(P_small):
        red.produced(small, lo, hi)
    red.finish(lambda lo, hi: seen.append((lo, hi)))
    assert seen == [(0, 256), (256, 300)] and torch.equal(small, base[:300] * 3)
    assert red.bounds(0) == [(0, 0)]
    red.produced(small[:0], 0, 0)
    seen.clear()
    red.finish(lambda lo, hi: seen.append((lo, hi)))
    assert seen == [] and red._pending == []
    dist.barrier()
    dist.destroy_process_group()


def test_compact_gradient_reducers():
    """all_reduce_compact / PipelinedCompactReducer / ProducerPipelinedReducer (fsgs_amd/dist.py): the [P,14] gradient
    of the HIP step driver, as one collective, consumer-side chunks and producer-side chunks."""
    mp.spawn(_compact_worker, args=(2, _free_port(), ""), nprocs=2, join=True)


def _producer_world4_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fsgs_amd import dist as fdist

    fdist.init_from_env(backend="gloo")
    red = fdist.ProducerPipelinedReducer(4)
    scale = sum(q + 1 for q in range(world))
    # three "steps" with a densification between them: P is never a multiple of 256 x chunks, grows, then shrinks below
    # one chunk; the compact gradient buffer is re-made whenever P changes (fast_step._Buffers does the same)
    for step, P in enumerate((1037, 1037, 2911, 130)):
        base = torch.arange(P * 14, dtype=torch.float32).reshape(P, 14) + step
        gc = base * (rank + 1)
        bounds = red.bounds(P)
        assert bounds[0][0] == 0 and bounds[-1][1] == P and len(bounds) <= 4
        assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:])) and all(lo % 256 == 0 for lo, _ in bounds)
        done = []
        for lo, hi in bounds:
            red.produced(gc, lo, hi)

        def adam_rows(lo, hi):
            assert torch.equal(gc[lo:hi], base[lo:hi] * scale)  # this chunk's exchange has landed
            done.append((lo, hi))

        red.finish(adam_rows)
        assert done == bounds and torch.equal(gc, base * scale)
    dist.barrier()
    dist.destroy_process_group()


def test_producer_pipelined_reducer_world_four_uneven_rows_and_a_changing_cloud():
    """dist.ProducerPipelinedReducer over four gloo ranks: row counts that are not multiples of 256 x chunks, a cloud
    that grows and shrinks between steps (densify / prune), every row exchanged exactly once before its Adam chunk."""
    mp.spawn(_producer_world4_worker, args=(4, _free_port(), ""), nprocs=4, join=True)


def _direct_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fsgs_amd import dist as fdist

    fdist.init_from_env(backend="gloo")
    red = fdist.DirectAllReduce()
    scale = sum(q + 1 for q in range(world))
    for P in (1024, 1037, 3, 1037, 64):  # multiples of 4 N floats and not; growing and shrinking (the buffers are re-made)
        base = torch.arange(P * 14, dtype=torch.float32).reshape(P, 14) - 7.0
        gc = base * (rank + 1)
        red(gc)
        assert torch.equal(gc, base * scale), P
    with pytest.raises(ValueError):
        red(torch.zeros(8, 14)[:, :7])  # a strided view cannot be reduced in place
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_direct_reduce_scatter_all_gather_equals_all_reduce(world):
    """dist.DirectAllReduce (all-to-all of shards, local sum, all-gather -- SURVEY.md s5's direct form for point-to-point
    xGMI) gives the all-reduce's result in place, for element counts that need the padded staging path too."""
    mp.spawn(_direct_worker, args=(world, _free_port(), ""), nprocs=world, join=True)


def _densify_world8_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fsgs_amd import dist as fdist

    fdist.init_from_env(backend="gloo")
    torch.manual_seed(1234)  # every rank seeds alike (bench.py / the harness do): the split samples must come out alike
    P = 700
    pr = _params(P, 5)
    pr["_scaling"] = pr["_scaling"] * 0.5 - 4.0  # around exp(-4): both sides of 1 % of the scene radius
    pr["_opacity"] = pr["_opacity"] * 2.0        # logits on both sides of sigmoid^-1(0.05)
    pc = GaussianCloud(pr, device="cpu", scene_radius=1.5)
    pc.training_setup(fused=False)
    # Adam moments exist and are identical replicas (one step on a rank-independent gradient)
    for k in PARAM_NAMES:
        pc.params[k].grad = torch.ones_like(pc.params[k]) * 0.01
    pc.optimizer.step()
    pc.optimizer.zero_grad(set_to_none=True)
    # every rank has seen its OWN camera: a rank-dependent statistic over a rank-dependent visible subset
    g = torch.Generator().manual_seed(100 + rank)
    vis = torch.rand(P, generator=g) < 0.6
    pc.variables["xyz_gradient_accum"][vis] += (torch.rand(int(vis.sum()), 1, generator=g) * 8e-4)
    pc.variables["denom"][vis] += 1.0
    pc.variables["max_radii2D"][vis] = torch.maximum(pc.variables["max_radii2D"][vis],
                                                     torch.randint(1, 40, (int(vis.sum()),), generator=g).float())
    mine = [pc.variables[k].clone() for k in ("xyz_gradient_accum", "denom", "max_radii2D")]
    fdist.sync_densification_stats(pc)
    # (SUM, SUM, MAX) over the eight ranks, checked against a gather of the ranks' own values
    for k, own, op in zip(("xyz_gradient_accum", "denom", "max_radii2D"), mine, ("sum", "sum", "max")):
        parts = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(parts, own)
        want = torch.stack(parts).sum(0) if op == "sum" else torch.stack(parts).max(0).values
        assert torch.allclose(pc.variables[k], want, rtol=1e-6, atol=0), k
    pc.densify_and_prune(2e-4, 0.05, 20)
    Pn = pc.num_points
    assert Pn != P  # something was cloned / split / pruned
    # lock-step: size, every parameter, both Adam moments of every group -- bit-identical on all eight ranks
    sig = [torch.tensor([float(Pn)], dtype=torch.float64)]
    for grp in pc.optimizer.param_groups:
        p_ = grp["params"][0]
        st = pc.optimizer.state[p_]
        for t_ in (p_.detach(), st["exp_avg"], st["exp_avg_sq"]):
            assert t_.shape[0] == Pn
            sig.append(torch.stack([t_.double().sum(), (t_.double() ** 2).sum(), t_.double().reshape(Pn, -1)[:, 0].dot(
                torch.arange(Pn, dtype=torch.float64))]))
    sig = torch.cat([x.reshape(-1) for x in sig])
    parts = [torch.empty_like(sig) for _ in range(world)]
    dist.all_gather(parts, sig)
    assert all(torch.equal(parts[0], q) for q in parts[1:]), "ranks left the densification in different states"
    assert all(float(pc.variables[k].abs().sum()) == 0.0 and pc.variables[k].shape[0] == Pn
               for k in ("xyz_gradient_accum", "denom", "max_radii2D"))
    # ... and the exchange that follows works on the new size (a shard count that does not divide it)
    red = fdist.DirectAllReduce()
    gc = torch.full((Pn, 14), float(rank + 1))
    red(gc)
    assert torch.equal(gc, torch.full((Pn, 14), float(sum(range(1, world + 1)))))
    dist.barrier()
    dist.destroy_process_group()


def test_world_eight_densification_stays_in_lockstep():
    """The world size of configuration C3 (8 ranks, one camera each): densification statistics reduced as (SUM, SUM, MAX),
    then densify_and_prune with identically seeded generators leaves all eight replicas -- parameters and Adam moments --
    bit-identical, and the direct exchange pads its shards for the new row count (VERDICT r3 #2b)."""
    mp.spawn(_densify_world8_worker, args=(8, _free_port(), ""), nprocs=8, join=True)
