"""batch_prepare against the framework's operators and float64 on one synthetic batch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hivt_gan_amd as hv  # noqa: E402

b = hv.make_batch(1, num_agents=200, num_lanes=1000, all_valid=True, local_radius=150.0).to("cuda:0")
prep = hv.ops.batch_prepare(b, 20)
ang = b["rotate_angles"]
sin, cos = torch.sin(ang), torch.cos(ang)
rot = torch.stack([cos, -sin, sin, cos], dim=-1).view(-1, 2, 2)
a64 = ang.double()
rot64 = torch.stack([a64.cos(), -a64.sin(), a64.sin(), a64.cos()], dim=-1).view(-1, 2, 2)
print("rot kernel vs torch  max abs", float((prep["rotate_mat"] - rot).abs().max()))
print("rot kernel vs f64    max abs", float((prep["rotate_mat"].double() - rot64).abs().max()))
print("rot torch  vs f64    max abs", float((rot.double() - rot64).abs().max()))
y = b["y"]
y_t = y[..., 0:1] * rot[:, None, 0, :] + y[..., 1:2] * rot[:, None, 1, :]
y64 = y.double()[..., 0:1] * rot64[:, None, 0, :] + y.double()[..., 1:2] * rot64[:, None, 1, :]
print("y   kernel vs torch  max abs", float((prep["y"] - y_t).abs().max()), "scale", float(y_t.abs().max()))
print("y   kernel vs f64", float((prep["y"].double() - y64).abs().max()), "torch vs f64", float((y_t.double() - y64).abs().max()))
x = b["x"][:, :20].transpose(0, 1)
x_t = x[..., 0:1] * rot[:, 0, :] + x[..., 1:2] * rot[:, 1, :]
print("x   kernel vs torch  max abs", float((prep["x_rot"].view(20, -1, 2) - x_t).abs().max()), "scale", float(x_t.abs().max()))
print("x_steps equal", bool(torch.equal(prep["x_steps"].view(20, -1, 2), x)))
