import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mobocmf_amd.util import synthetic
dev = torch.device("cuda:0")
cfg = dict(synthetic.CONFIGS["C3"])
sur = bench.build_surrogates(cfg, [0, 1, 2], dev)
gens = [torch.Generator(device=dev) for _ in sur]
streams = [torch.cuda.Stream(device=dev) for _ in sur]
torch.cuda.synchronize()
for _ in range(3): bench.one_step(sur, cfg, gens, streams)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): bench.one_step(sur, cfg, gens, streams)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue per bench step: %.2f ms, total per step %.2f ms" % ((t1 - t0) * 100, (t2 - t0) * 100))
# host-only cost: tiny problem (GPU work negligible)
cfg2 = dict(d=8, L=2, M=128, N=128, S=1, outputs=3)
sur2 = bench.build_surrogates(cfg2, [0, 1, 2], dev)
for _ in range(3): bench.one_step(sur2, cfg2, gens, streams)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): bench.one_step(sur2, cfg2, gens, streams)
torch.cuda.synchronize()
print("tiny problem, per bench step: %.2f ms" % ((time.perf_counter() - t0) * 50))
