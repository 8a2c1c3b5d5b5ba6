"""Data-parallel path on CPU: two ranks over gloo run the product's real ``training_step`` /
``GanTrainer`` (guards, critic + generator updates, flat gradient buckets, one all-reduce per
backward).  Only the generator's forward is replaced by a dense CPU stand-in, because the edge
attention kernels exist for the GPU only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

import hivt_gan_amd as hv


class CpuHiVT(hv.HiVT):
    """Real HiVT parameters, losses and step logic; forward = temporal encoder + decoder only."""

    def forward(self, data):
        T = self.historical_steps
        x = data["x"][:, :T].transpose(0, 1)
        x = F.pad(x, (0, self.decoder.hidden_size - x.shape[-1]))
        local = self.local_encoder.temporal_encoder(x=x, padding_mask=data["padding_mask"][:, :T])
        glob = self.global_interactor.multihead_proj(self.global_interactor.norm(local))
        glob = glob.view(-1, self.num_modes, self.decoder.hidden_size).transpose(0, 1)
        return self.decoder(local_embed=local, global_embed=glob)

    def _supervised_losses(self, data, y_hat, pi):
        reg_mask = ~data["padding_mask"][:, self.historical_steps:]
        l2 = (torch.norm(y_hat[..., :2] - data["y"], p=2, dim=-1) * reg_mask).sum(dim=-1)
        best = l2.argmin(dim=0)
        y_best = y_hat[best, torch.arange(y_hat.shape[1])]
        reg = self.reg_loss(y_best[reg_mask], data["y"][reg_mask])
        cnt = reg_mask.sum(dim=-1)
        keep = cnt > 0
        soft = F.softmax(-l2[:, keep] / (cnt[keep] + 1e-6), dim=0).t().detach()
        return reg, self.cls_loss(pi[keep], soft), best


def make_model():
    torch.manual_seed(2022)
    # critic_lr = 0 keeps the critics fixed between the two backwards of a step, so the generator
    # gradients of the distributed run can be reproduced exactly in one process
    m = CpuHiVT(embed_dim=16, num_heads=2, dropout=0.0, num_temporal_layers=1, num_global_layers=1, use_gan=True,
                rotate=False, check_finite=True, critic_lr=0.0)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, poison_rank, out_dir, critic_steps=1, overlap=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        model = make_model()
        model.critic_steps = critic_steps
        if rank == 1:  # construction-time broadcast must make this irrelevant
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.123)
        trainer = hv.GanTrainer(model, overlap=overlap)
        assert trainer.distributed and set(trainer.buckets) == {"generator", "critic"}
        assert len(trainer.buckets["generator"].chunks) == 4  # agent-agent, temporal + lane, interaction, decoder
        reduced = {}
        calls = []
        plain_sync = trainer._sync

        def recording_sync(group, bad=None):
            share = plain_sync(group, bad)
            bucket = trainer.buckets[group]
            reduced[group] = bucket.gradients()
            calls.append(group)
            return share
        trainer._sync = recording_sync
        batch = hv.make_batch(2, first_index=10 * rank)
        if rank == poison_rank:
            batch["y"][0, 0, 0] = float("nan")
        collectives = []
        plain_all_reduce = dist.all_reduce

        def counting_all_reduce(tensor, *a, **k):
            collectives.append(tensor.numel())
            return plain_all_reduce(tensor, *a, **k)
        dist.all_reduce = counting_all_reduce
        try:
            logs = trainer.step(batch)
        finally:
            dist.all_reduce = plain_all_reduce
        assert calls == ["critic"] * critic_steps + ["generator"], calls
        if not overlap:
            # one collective per optimizer group and backward pass (the whole flat buffer with its contribution
            # flag), plus the one for the packed logs
            sizes = [trainer.buckets["critic"].total + 1] * critic_steps + [trainer.buckets["generator"].total + 1]
            assert collectives[:-1] == sizes and len(collectives) == len(sizes) + 1, (collectives, sizes)
        # the logs are averaged over the ranks that produced them, in one collective
        if poison_rank < 0:
            both = [None, None]
            dist.all_gather_object(both, float(logs["reg_loss"]))
            assert both[0] == both[1]
        elif rank == poison_rank:  # averaged over the ranks that produced the entry: this one alone
            assert float(logs["skipped"]) == 1.0 and "reg_loss" not in logs
        else:
            assert "skipped" not in logs and bool(torch.isfinite(logs["reg_loss"]))
        state = torch.cat([p.detach().flatten() for p in model.parameters()])
        gathered = [torch.empty_like(state) for _ in range(world)]
        dist.all_gather(gathered, state)
        assert torch.equal(gathered[0], gathered[1]), "ranks diverged after one step"
        assert set(reduced) == {"generator", "critic"}, "every rank must enter both all-reduces"
        if rank == 0:
            torch.save(reduced, os.path.join(out_dir, "dp_grads.pt"))
    finally:
        dist.destroy_process_group()


def single_process_reference(poison_rank):
    """Same two batches in one process: average the per-batch gradients by hand."""
    model = make_model()
    (opt_g, opt_d), _ = model.configure_optimizers()
    batches = [hv.make_batch(2, first_index=0), hv.make_batch(2, first_index=10)]
    gen, cr = model.generator_parameters(), model.critic_parameters()
    captured = {"critic": [], "generator": []}

    class Hold:  # optimizer stand-in: keep the weights fixed while both batches are differentiated
        def zero_grad(self, set_to_none=False):
            for p in gen + cr:
                p.grad = None

        def step(self):
            pass

    sn_state = {k: v.clone() for k, v in model.state_dict().items() if k.endswith("_u") or k.endswith("_v")}
    for r, b in enumerate(batches):
        model.load_state_dict(sn_state, strict=False)  # every rank starts from the same power-iteration state

        def grab(group, bad=None, r=r):
            ps = cr if group == "critic" else gen
            captured[group].append([torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in ps])
        if r == poison_rank:
            captured["critic"].append([torch.zeros_like(p) for p in cr])
            captured["generator"].append([torch.zeros_like(p) for p in gen])
        else:
            model.training_step(b, Hold(), Hold(), sync_grads=grab)
    return {group: torch.cat([((captured[group][0][i] + captured[group][1][i]) / 2).flatten()
                              for i in range(len(captured[group][0]))]) for group in captured}


@pytest.mark.parametrize("poison_rank,overlap", [(-1, False), (1, False), (-1, True), (1, True)])
def test_two_rank_step_matches_manual_average(tmp_path, poison_rank, overlap):
    """``overlap``: every chunk of a bucket is reduced from a gradient hook during the backward pass
    (asynchronous collectives) instead of all at once after it; the reduced values must be the same."""
    port = free_port()
    mp.spawn(worker, args=(2, port, poison_rank, str(tmp_path), 1, overlap), nprocs=2, join=True)
    dp = torch.load(os.path.join(str(tmp_path), "dp_grads.pt"))
    ref = single_process_reference(poison_rank)
    for group in ("critic", "generator"):
        assert dp[group].shape == ref[group].shape
        assert float(ref[group].abs().max()) > 0
        # the two sides sum in different orders (worker processes run other thread counts than this
        # one), so agreement is judged against the scale of the gradient vector
        err = float((dp[group] - ref[group]).abs().max())
        assert err <= 1e-6 * float(ref[group].abs().max()), f"{group} gradients differ by {err:.3e}"


def discarding_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hivt_gan_amd.trainer import FlatGradBucket
        params = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 2))]
        bucket = FlatGradBucket(params, overlap=False)
        for step in range(2):  # the second pass starts from gradients that are views of the buffer
            for i, p in enumerate(params):
                g = torch.full_like(p, float(i + 1 + step))
                if rank == 1:
                    g[..., 0] = float("nan")  # what a discarded batch may leave behind
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.copy_(g)
            share = bucket.finish(torch.tensor(rank == 1))
            # the discarding rank enters the sum as zeros (selected, so its NaNs stay out) and the sum is still
            # divided by the world size, as a zero-loss rank under the reference's DDP is
            assert float(share) == 0.5
            for i, p in enumerate(params):
                assert torch.equal(p.grad, torch.full_like(p, (i + 1 + step) / 2)), (rank, step, p.grad)
        share = bucket.finish(torch.tensor(True))
        assert float(share) == 0.0 and all(float(p.grad.abs().max()) == 0.0 for p in params)
    finally:
        dist.destroy_process_group()


def test_discarding_rank_counts_as_a_zero_gradient_contributor():
    mp.spawn(discarding_worker, args=(2, free_port()), nprocs=2, join=True)


def test_two_critic_steps_with_a_poisoned_rank_stay_in_lock_step(tmp_path):
    """critic_steps = 2: the rank that discards its batch must enter two critic reductions and one
    generator reduction like the healthy rank (the worker asserts the sequence and identical replicas)."""
    port = free_port()
    mp.spawn(worker, args=(2, port, 1, str(tmp_path), 2, False), nprocs=2, join=True)


def test_flat_bucket_packs_and_hands_back_views():
    lin = torch.nn.Linear(3, 2)
    extra = torch.nn.Parameter(torch.zeros(5))  # never receives a gradient
    from hivt_gan_amd.trainer import FlatGradBucket
    bucket = FlatGradBucket(list(lin.parameters()) + [extra])
    lin(torch.ones(4, 3)).sum().backward()
    flat = bucket.pack()
    assert bucket.offsets == [0, 32, 64] and flat.shape == (96,)  # every gradient on a 128-byte boundary
    assert torch.equal(flat[:6].view(2, 3), torch.full((2, 3), 4.0)) and torch.all(flat[6:32] == 0)
    assert torch.all(flat[32:34] == 4.0) and torch.all(flat[34:] == 0)
    flat.mul_(0.5)
    bucket.unpack()
    assert lin.weight.grad.data_ptr() == flat.data_ptr()
    assert torch.equal(lin.weight.grad, torch.full((2, 3), 2.0)) and torch.all(extra.grad == 0)
