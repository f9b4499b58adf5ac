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


def _worker_overlap(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from multilingual_text_to_speech_amd import dist as D
    D.init(backend='gloo')
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    unused = torch.nn.Parameter(torch.ones(4))                      # never receives a gradient
    params = list(model.parameters()) + [unused]
    buckets = D.GradientBuckets(params, bucket_bytes=32, overlap=True)
    assert len(buckets.buckets) > 2
    results = []
    for step in range(2):                                           # second step checks re-arming / zeroing
        buckets.zero_grad()
        g = torch.Generator().manual_seed(100 * step + rank)
        x = torch.randn(4, 7, generator=g)
        ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
        ref.load_state_dict(model.state_dict())
        ref(x).pow(2).mean().backward()
        model(x).pow(2).mean().backward()                           # hooks launch the collectives during this call
        launched = sum(w is not None for w in buckets._works)
        buckets.all_reduce()
        assert all(p.grad.data_ptr() >= buckets.flat[buckets._bucket_of[id(p)]].data_ptr() for p in params)   # still views
        results.append(dict(local=[p.grad.clone() for p in ref.parameters()], reduced=[p.grad.clone() for p in model.parameters()],
                            unused=unused.grad.clone(), launched=launched))
    torch.save(results, f'{out}/o{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucket_allreduce_world2(tmp_path):
    """Gradient-as-bucket-view + post-accumulate hooks: collectives start inside backward, results equal the mean."""
    port = _free_port()
    mp.spawn(_worker_overlap, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'o0.pt')
    r1 = torch.load(tmp_path / 'o1.pt')
    for s0, s1 in zip(r0, r1):
        assert s0['launched'] >= 2, 'complete buckets must be launched by the hooks before all_reduce() is called'
        assert not s0['unused'].any()
        for l0, l1, g0, g1 in zip(s0['local'], s1['local'], s0['reduced'], s1['reduced']):
            torch.testing.assert_close(g0, (l0 + l1) / 2, rtol=1e-6, atol=1e-7)
            assert torch.equal(g0, g1)


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


# ---- train.evaluate: validation batches dealt round-robin to the ranks, all ranks get the single-process means ------------------
class _StubModel(torch.nn.Module):
    def forward(self, text, text_length, target, target_length, speakers, languages, tf):
        return target * 0.5, target * 0.25, target.sum(1), None, None, None


class _StubCrit:
    def __call__(self, tl, ml, pre, tgt, post, tgt2, stop, stop_t, align, spk, spk_pred, enc, cls):
        parts = {'mel_pre': (pre - tgt).pow(2).mean(), 'mel_pos': (post - tgt).pow(2).mean(), 'stop_token': stop.abs().mean()}
        return sum(parts.values()), parts


def _eval_batches():
    g = torch.Generator().manual_seed(5)
    out = []
    for i in range(5):                                   # odd count: the ranks get 3 and 2 batches
        mel = torch.randn(2, 4, 6 + i, generator=g)
        out.append((torch.zeros(2, 3, dtype=torch.long), torch.tensor([3, 3]), mel, None, torch.tensor([6 + i] * 2),
                    torch.zeros(2, 6 + i), None, None))
    return out


def _worker_eval(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from multilingual_text_to_speech_amd import dist as D
    import train
    D.init(backend='gloo')
    res = train.evaluate(None, _eval_batches(), _StubModel(), _StubCrit(), torch.device('cpu'), rank, world)
    torch.save(res, f'{out}/e{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluation_equals_single_process(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import train
    single = train.evaluate(None, _eval_batches(), _StubModel(), _StubCrit(), torch.device('cpu'))
    port = _free_port()
    mp.spawn(_worker_eval, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(f'{tmp_path}/e{r}.pt')
        assert set(got) == set(single)
        for k in single:
            assert abs(got[k] - single[k]) <= 1e-6 * max(1.0, abs(single[k])), (k, got[k], single[k])


# ---- classifier loss under data parallelism: local mean -> share of the global mean ---------------------------------------------
def _worker_scale(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from multilingual_text_to_speech_amd import dist as D
    D.init(backend='gloo')
    n_local = [10, 30][rank]
    vals = torch.arange(n_local, dtype=torch.float32) + 100.0 * rank          # per-item losses of this rank's valid characters
    share = D.global_mean_scale(n_local, torch.device('cpu'))
    contrib = vals.mean() * share[0]                                           # what the rank adds to its loss
    t = contrib.clone()
    dist.all_reduce(t)
    torch.save(dict(share=share, mean_of_ranks=t / world), f'{out}/s{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_local_mean_is_rescaled_to_the_global_mean(tmp_path):
    from multilingual_text_to_speech_amd import dist as D
    assert float(D.global_mean_scale(17, torch.device('cpu'))) == 1.0         # not data parallel: no change
    port = _free_port()
    mp.spawn(_worker_scale, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    everything = torch.cat((torch.arange(10, dtype=torch.float32), torch.arange(30, dtype=torch.float32) + 100.0))
    for r in range(2):
        got = torch.load(f'{tmp_path}/s{r}.pt')
        assert abs(float(got['share']) - [0.5, 1.5][r]) < 1e-6
        assert abs(float(got['mean_of_ranks']) - float(everything.mean())) < 1e-4


def _worker_guard(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from multilingual_text_to_speech_amd import dist as D
    D.init(backend='gloo')
    seen = []
    for step, bad_rank in enumerate((None, 1, None, 0)):
        flag = torch.zeros(2, dtype=torch.int32)            # [invalid input, persistent kernel error] of this rank's GPU
        if bad_rank == rank:
            flag[1] = 2
        seen.append(D.agree_on_guard(flag).tolist())
    torch.save(seen, f'{out}/guard{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_guard_words_are_an_all_rank_decision(tmp_path):
    """The guarded optimizer step (mtts.h AdamArgs.guard) skips on the device when this GPU's error word is set.  Under data
    parallelism the word is MAX-all-reduced first (dist.agree_on_guard, called by FusedAdam.step): when ONE rank's persistent decoder
    gave up, EVERY rank sees the word, skips the step and raises at its next poll - replicas never diverge (reference semantics:
    one model, one optimizer step per global batch, train.py:84-85,173-179)."""
    port = _free_port()
    mp.spawn(_worker_guard, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0, s1 = torch.load(tmp_path / 'guard0.pt'), torch.load(tmp_path / 'guard1.pt')
    assert s0 == s1 == [[0, 0], [0, 2], [0, 0], [0, 2]]


def test_agree_on_guard_is_a_no_op_outside_data_parallel():
    from multilingual_text_to_speech_amd import dist as D
    flag = torch.tensor([0, 2], dtype=torch.int32)
    assert D.agree_on_guard(flag) is flag and flag.tolist() == [0, 2]
