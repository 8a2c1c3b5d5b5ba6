import collections, sys, os, torch
sys.path.insert(0, "/root/repo")
import hivt_gan_amd as hv
from bench import model_config
dev = torch.device("cuda", 0)
torch.manual_seed(2022)
model = hv.HiVT(**model_config(128)).to(dev)
model.train()
b = hv.make_batch(128).to(dev)
y_hat, pi = model(b)
loss = y_hat.square().mean() + pi.square().mean()
seen, stack = set(), [loss.grad_fn]
fan = collections.defaultdict(list)
while stack:
    f = stack.pop()
    if f is None or f in seen:
        continue
    seen.add(f)
    for i, (g, nr) in enumerate(f.next_functions):
        if g is not None:
            fan[(g, nr)].append((type(f).__name__, i))
            stack.append(g)
for (g, nr), prods in fan.items():
    if len(prods) > 1 and "AccumulateGrad" not in type(g).__name__:
        print(type(g).__name__, "input", nr, "<-", prods)
