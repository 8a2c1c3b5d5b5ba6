"""The training step recorded into a HIP graph (``GraphedStep``) against the same steps launched eagerly:
identical parameters after several steps, a tripped guard that leaves the model untouched, fresh dropout
masks on every replay, and no runtime memset node in the recording."""
import pytest
import torch

import hivt_gan_amd as hv
from hivt_gan_amd import ops
from hivt_gan_amd.graph import GraphedStep

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def make(dropout, lr=5e-4, critic_lr=1e-4, weight_decay=1e-4):
    torch.manual_seed(11)
    m = hv.HiVT(embed_dim=64, num_heads=8, dropout=dropout, use_gan=True, num_temporal_layers=2, num_global_layers=2,
                lr=lr, critic_lr=critic_lr, weight_decay=weight_decay).to(DEV)
    m.train()
    if dropout == 0.0:
        for critic in m.critics.values():
            for mod in critic.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
    return m


def batches_and_spec(model, count=3, scenes=5):
    raw = [hv.make_batch(scenes, first_index=400 + 10 * i).to(DEV) for i in range(count)]
    spec = hv.GanTrainer.calibrate(model, raw)
    return [hv.pad_batch(b, spec) for b in raw], spec


def eager_run(model, spec, padded, order):
    trainer = hv.GanTrainer(model, distributed=False, edge_capacity=spec.edge_capacity)
    for i in order:
        b = padded[i].clone()
        logs = trainer.step(b)
    return trainer, logs


def test_replayed_steps_equal_eager_steps():
    order = [0, 1, 2, 1]
    ref_model = make(0.0)
    padded, spec = batches_and_spec(ref_model)
    # eager reference: the same steps and nothing else -- the recording's warm-up steps leave no trace in the
    # parameters, the optimizer state or the critics' power-iteration vectors, and the capture executes nothing
    eager_run(ref_model, spec, padded, order)
    model = make(0.0)
    trainer = hv.GanTrainer(model, distributed=False, edge_capacity=spec.edge_capacity)
    step = GraphedStep(trainer, spec, padded[0], warmup=3)
    assert step.nodes.get("memset", 0) == 0 and step.nodes.get("kernel", 0) > 100, step.nodes
    # batches laid out in one buffer arrive in a single copy, plain ones tensor by tensor: same steps either way
    from hivt_gan_amd.graph import pack_inputs
    packed = [pack_inputs(b) if i % 2 == 0 else b for i, b in enumerate(padded)]
    assert packed[0]["_arena_layout"] == step.static["_arena_layout"]
    for i in order:
        logs = step.run(packed[i])
    torch.cuda.synchronize()
    assert float(logs["skipped"]) == 0.0
    for (k, a), (_, b) in zip(model.state_dict().items(), ref_model.state_dict().items()):
        assert torch.equal(a, b), f"{k} differs between replayed and eager steps"
    assert trainer.overflowed_steps() == 0
    for k in ("x", "padding_mask", "edge_index", "lane_actor_vectors"):
        assert torch.equal(step.static[k], padded[order[-1]][k]), k


def test_tripped_guard_in_a_replay_leaves_the_model_untouched():
    model = make(0.0)
    padded, spec = batches_and_spec(model, count=2)
    trainer = hv.GanTrainer(model, distributed=False, edge_capacity=spec.edge_capacity)
    step = GraphedStep(trainer, spec, padded[0])
    step.run(padded[1])
    before = {k: v.clone() for k, v in model.state_dict().items()}
    poisoned = padded[0].clone()
    poisoned["y"][3, 2, 0] = float("nan")
    logs = step.run(poisoned)
    torch.cuda.synchronize()
    assert float(logs["skipped"]) == 1.0
    for k, v in model.state_dict().items():
        if k.endswith("_u") or k.endswith("_v"):  # the power iteration of the critics' forward passes still ran
            continue
        assert torch.equal(v, before[k]), k
    logs = step.run(padded[1])
    assert float(logs["skipped"]) == 0.0


def test_every_replay_draws_new_dropout_masks():
    # learning rates and weight decay 0: the parameters stay put, so only the masks can change the losses
    model = make(0.1, lr=0.0, critic_lr=0.0, weight_decay=0.0)
    padded, spec = batches_and_spec(model, count=1)
    trainer = hv.GanTrainer(model, distributed=False, edge_capacity=spec.edge_capacity)
    step = GraphedStep(trainer, spec, padded[0])
    seen = []
    for _ in range(3):
        seen.append(float(step.run(padded[0])["reg_loss"]))
    assert len(set(seen)) == 3, seen
    model0 = make(0.0, lr=0.0, critic_lr=0.0, weight_decay=0.0)
    trainer0 = hv.GanTrainer(model0, distributed=False, edge_capacity=spec.edge_capacity)
    step0 = GraphedStep(trainer0, spec, padded[0])
    same = [float(step0.run(padded[0])["reg_loss"]) for _ in range(3)]
    assert len(set(same)) == 1, same
    assert ops.DEVICE_SEED is None  # the recording hands the seed scalar back


def test_recorded_data_parallel_step_in_segments_world_of_one():
    """The data-parallel step recorded as graph segments with the all-reduces between them (RCCL, a world of
    one rank): same parameters as the single-process recording after the same steps, and a step that the
    only rank discards leaves the model untouched."""
    import socket
    import torch.distributed as dist
    order = [0, 1, 0]
    ref_model = make(0.0)
    padded, spec = batches_and_spec(ref_model, count=2)
    ref_tr = hv.GanTrainer(ref_model, distributed=False, edge_capacity=spec.edge_capacity)
    ref_step = GraphedStep(ref_tr, spec, padded[0])
    for i in order:
        ref_step.run(padded[i])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=DEV)
    try:
        model = make(0.0)
        tr = hv.GanTrainer(model, distributed=True, edge_capacity=spec.edge_capacity)
        step = GraphedStep(tr, spec, padded[0])
        kinds = [k for k, _ in step.segments]
        assert kinds == ["graph", "call", "graph", "call", "graph"], kinds  # critic reduction, generator reduction
        assert step.nodes.get("memset", 0) == 0
        for i in order:
            logs = step.run(padded[i])
        torch.cuda.synchronize()
        assert float(logs["skipped"]) == 0.0
        for (k, a), (_, b) in zip(model.state_dict().items(), ref_model.state_dict().items()):
            assert torch.equal(a, b), f"{k}: segmented data-parallel recording differs from the single-process one"
        before = {k: v.clone() for k, v in model.state_dict().items()}
        poisoned = padded[0].clone()
        poisoned["y"][1, 0, 0] = float("nan")
        logs = step.run(poisoned)
        assert float(logs["skipped"]) == 1.0
        for k, v in model.state_dict().items():
            if not (k.endswith("_u") or k.endswith("_v")):
                assert torch.equal(v, before[k]), k
    finally:
        dist.destroy_process_group()


def test_two_recordings_of_one_trainer_with_their_own_shapes():
    """Shape buckets: two recordings made on one trainer, each at the padded shape of its own batch, replayed in
    turn -- the same training run as the eager steps at the larger shape (padding is inert, so only the order of
    a few sums moves), and neither recording disturbs the trainer's own capacities."""
    small = hv.make_batch(4, first_index=400).to(DEV)
    large = hv.make_batch(6, first_index=430).to(DEV)
    ref_model = make(0.0)
    spec_all = hv.GanTrainer.calibrate(ref_model, [small, large])
    padded_all = [hv.pad_batch(small, spec_all), hv.pad_batch(large, spec_all)]
    # (the recordings' warm-up steps leave the training state as they found it, and capturing executes nothing)
    order = [0, 1, 0, 0, 1]
    eager_run(ref_model, spec_all, padded_all, order)
    model = make(0.0)
    trainer = hv.GanTrainer(model, distributed=False, edge_capacity=spec_all.edge_capacity)
    specs = [hv.GanTrainer.calibrate(model, [b], slack=1.0, node_slack=1.0) for b in (small, large)]
    assert specs[0].num_nodes < specs[1].num_nodes and specs[0].edge_capacity != specs[1].edge_capacity
    padded = [hv.pad_batch(b, sp) for b, sp in zip((small, large), specs)]
    steps = [GraphedStep(trainer, sp, pb, warmup=3, seed_offset=i + 1) for i, (sp, pb) in enumerate(zip(specs, padded))]
    assert trainer.edge_capacity == tuple(spec_all.edge_capacity)
    for i in order:
        logs = steps[i].run(padded[i])
    torch.cuda.synchronize()
    assert float(logs["skipped"]) == 0.0 and trainer.overflowed_steps() == 0
    # compared through what the two models compute on a fresh batch, not parameter by parameter: AdamW turns the
    # rounding noise of gradients that are zero by construction (the key biases, which a softmax cannot see) into
    # steps of the size of the learning rate, in a direction that depends on the order of the sums
    probe = hv.make_batch(3, first_index=470).to(DEV)
    outs = []
    for m in (model, ref_model):
        m.eval()
        with torch.no_grad():
            outs.append(m(probe.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()), float((a - b).abs().max())


def test_recording_leaves_the_training_state_untouched():
    """``GraphedStep`` warms up with real steps on whatever its static inputs hold; none of that may survive: the
    parameters, the optimizer state, the critics' power-iteration vectors and the overflow counter are bit-identical
    before and after a recording, also when a trained state is recorded again (the reference makes exactly one
    update per batch, ``models/hivt.py:287-342``)."""
    model = make(0.1)
    padded, spec = batches_and_spec(model, count=2)
    trainer = hv.GanTrainer(model, distributed=False, edge_capacity=spec.edge_capacity)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    step = GraphedStep(trainer, spec, padded[0], warmup=3)
    torch.cuda.synchronize()
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), f"{k} changed while recording"
    for opt in (trainer.opt_g, trainer.opt_d):
        for st in opt.state.values():
            assert float(st["step"]) == 0.0 and float(st["exp_avg"].abs().max()) == 0.0 \
                and float(st["exp_avg_sq"].abs().max()) == 0.0
    assert trainer.overflowed_steps() == 0
    step.run(padded[1])
    step.run(padded[0])
    torch.cuda.synchronize()
    trained = {k: v.clone() for k, v in model.state_dict().items()}
    moments = [{k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for st in trainer.opt_g.state.values()]
    step.record()  # e.g. a second shape bucket made later, or a schedule that forces a new recording
    torch.cuda.synchronize()
    for k, v in model.state_dict().items():
        assert torch.equal(v, trained[k]), f"{k} changed while recording again"
    for st, ref in zip(trainer.opt_g.state.values(), moments):
        for k, v in ref.items():
            assert torch.equal(st[k], v), k


def test_learning_rate_change_needs_no_new_recording():
    """The recorded update reads its learning rate from a device scalar: after ``end_epoch`` the same recording keeps
    replaying and applies the new rate (equal to eager steps through the same schedule)."""
    order = [0, 1, 0, 1]
    ref_model = make(0.0)
    padded, spec = batches_and_spec(ref_model, count=2)
    ref = hv.GanTrainer(ref_model, distributed=False, edge_capacity=spec.edge_capacity)
    for n, i in enumerate(order):
        ref.step(padded[i].clone())
        if n == 1:
            for _ in range(20):
                ref.end_epoch()
    model = make(0.0)
    trainer = hv.GanTrainer(model, distributed=False, edge_capacity=spec.edge_capacity)
    step = GraphedStep(trainer, spec, padded[0])
    recordings = id(step.segments[0][1])
    for n, i in enumerate(order):
        step.run(padded[i])
        if n == 1:
            for _ in range(20):
                trainer.end_epoch()
    torch.cuda.synchronize()
    assert id(step.segments[0][1]) == recordings, "the step was recorded again"
    assert trainer.opt_g.param_groups[0]["lr"] != 5e-4
    for (k, a), (_, b) in zip(model.state_dict().items(), ref_model.state_dict().items()):
        assert torch.equal(a, b), f"{k} differs after the learning-rate change"


def test_a_batch_with_another_candidate_list_is_refused():
    """The recording built its agent-agent edges from all ordered pairs; a batch that says its candidate list is
    something else must not be replayed through it."""
    model = make(0.0)
    padded, spec = batches_and_spec(model, count=1)
    trainer = hv.GanTrainer(model, distributed=False, edge_capacity=spec.edge_capacity)
    step = GraphedStep(trainer, spec, padded[0])
    assert bool(step.static.get("all_pairs", False))
    other = padded[0].clone()
    other["all_pairs"] = False
    with pytest.raises(ValueError, match="all_pairs"):
        step.run(other)
    step.run(padded[0])
