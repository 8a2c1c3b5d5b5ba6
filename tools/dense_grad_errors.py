"""Per-tensor gradient errors of the dense-scene generator step: the GPU path against the oracle in float64, next to
the float32 CPU oracle's own error against float64 (what fp32 rounding alone costs on this graph).

    python tools/dense_grad_errors.py [top]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hivt_gan_amd as hv  # noqa: E402
import hivt_gan_amd.decoder  # noqa: E402,F401
import hivt_gan_amd.local_encoder  # noqa: E402,F401
from oracle import hivt_oracle as O  # noqa: E402
from test_gpu_modules import CFG64, batch64, rel_err, state64, is_shift_invariant  # noqa: E402


def oracle_grads(model, b, cfg, double):
    sd = state64(model.state_dict()) if double else \
        {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    batch = batch64(b) if double else b
    y_ref, pi_ref, y_rot = O.hivt_forward(sd, batch, dict(cfg), training=False)
    reg, cls, _ = O.supervised_losses(y_ref, pi_ref, y_rot, b["padding_mask"], 20)
    (reg + cls).backward()
    return {k: v.grad for k, v in sd.items() if getattr(v, "grad", None) is not None}


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    off = set(sys.argv[2:])  # fused paths to switch off: prepare decoder temporal table
    if "prepare" in off:
        hv.ops.batch_prepare_supported = lambda *a, **k: False
    if "decoder" in off:
        hv.decoder.MLPDecoder._forward_fused = None
        orig = hv.decoder.MLPDecoder.forward

        def plain(self, local_embed, global_embed):
            hid = self.hidden_size
            self.hidden_size = 10 ** 6  # fails the fused route's width test
            try:
                return orig(self, local_embed, global_embed)
            finally:
                self.hidden_size = hid
        hv.decoder.MLPDecoder.forward = plain
    if "temporal" in off:
        hv.ops.temporal_input = None
        real = hv.local_encoder.TemporalEncoder.forward

        def fwd(self, x, padding_mask, rows=None):
            x = hv.ops.expand_rows_with_tokens(x, rows, padding_mask.t(), self.padding_token.squeeze(1))
            x = hv.ops.append_token_add_steps(x, self.cls_token.view(-1), self.pos_embed.squeeze(1))
            return self.transformer_encoder.forward_last(x)
        hv.local_encoder.TemporalEncoder.forward = fwd
    if "torchrot" in off or "torchrot_x" in off or "torchrot_y" in off:
        real_prepare = hv.ops.batch_prepare

        def prepare(data, steps):
            prep = real_prepare(data, steps)
            rot = prep["rotate_mat"]
            if "torchrot_y" in off or "torchrot" in off:
                y = data["y"]
                prep["y"] = y[..., 0:1] * rot[:, None, 0, :] + y[..., 1:2] * rot[:, None, 1, :]
            if "torchrot_x" in off or "torchrot" in off:
                x = data["x"][:, :steps].transpose(0, 1)
                prep["x_rot"] = (x[..., 0:1] * rot[:, 0, :] + x[..., 1:2] * rot[:, 1, :]).reshape(-1, 2).contiguous()
            return prep
        hv.ops.batch_prepare = prepare
    print("switched off:", sorted(off))
    cfg = dict(CFG64, local_radius=150.0)
    torch.manual_seed(2022)
    model = hv.HiVT(**cfg, lr=5e-4, weight_decay=1e-4, T_max=64).to("cuda:0")
    model.train()
    b = hv.make_batch(1, num_agents=200, num_lanes=1000, all_valid=True, local_radius=150.0)
    g64 = oracle_grads(model, b, cfg, True)
    g32 = oracle_grads(model, b, cfg, False)
    g = b.to("cuda:0")
    y_hat, pi = model(g)
    reg, cls, _ = model._supervised_losses(g, y_hat, pi)
    (reg + cls).backward()
    rows = []
    for k, p in model.named_parameters():
        if k not in g64 or is_shift_invariant(k) or float(g64[k].abs().max()) == 0:
            continue
        rows.append((rel_err(p.grad, g64[k]), rel_err(g32[k], g64[k]), k))
    rows.sort(reverse=True)
    print("gpu_vs_f64  cpu32_vs_f64  tensor")
    for e_gpu, e_cpu, k in rows[:top]:
        print(f"{e_gpu:.2e}    {e_cpu:.2e}     {k}")
    print("tensors with gpu error above the cpu fp32 oracle's:", sum(1 for a, c, _ in rows if a > c), "of", len(rows))


if __name__ == "__main__":
    main()
