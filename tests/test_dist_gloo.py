"""World-size-2 gloo test (CPU) of the data-parallel plumbing: bucketed gradient all-reduce == mean of per-rank
gradients, sharding rule for grouped encoders, parameter broadcast."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from multilingual_text_to_speech_amd import dist as D
    r, w, _ = D.init(backend='gloo')
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    D.broadcast_parameters(model, 0)
    start = [p.detach().clone() for p in model.parameters()]
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(4, 7, generator=g)
    model(x).pow(2).mean().backward()
    local = [p.grad.clone() for p in model.parameters()]
    buckets = D.GradientBuckets(model.parameters(), bucket_bytes=64)   # tiny buckets -> several collectives
    assert len(buckets.buckets) > 1
    buckets.all_reduce()
    torch.save(dict(start=start, local=local, reduced=[p.grad.clone() for p in model.parameters()]), f'{out}/r{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'r0.pt')
    r1 = torch.load(tmp_path / 'r1.pt')
    for a, b in zip(r0['start'], r1['start']):
        assert torch.equal(a, b), 'broadcast_parameters must equalise replicas'
    for l0, l1, g0, g1 in zip(r0['local'], r1['local'], r0['reduced'], r1['reduced']):
        torch.testing.assert_close(g0, (l0 + l1) / 2, rtol=1e-6, atol=1e-7)
        assert torch.equal(g0, g1)


def test_shard_bounds_respects_language_groups():
    from multilingual_text_to_speech_amd.dist import shard_bounds
    assert shard_bounds(64, 1, 2) == (32, 64)
    assert shard_bounds(240, 3, 8, groups=5) == (90, 120)
    with pytest.raises(ValueError):
        shard_bounds(256, 0, 8, groups=5)        # BASELINE's 256 is not divisible by 5 languages x 8 ranks
