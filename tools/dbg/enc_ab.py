import sys, time
sys.path.insert(0, ".")
import torch
from gauspcc_amd import arithmetic
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
n = 50_000_000
mean = (torch.randn(n, generator=g) * 2).to(dev); scale = (torch.rand(n, generator=g) * 3 + 0.05).to(dev)
q = (torch.rand(n, generator=g) * 0.5 + 0.75).to(dev); x = (mean + torch.randn(n, generator=g).to(dev) * scale).contiguous()
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mn, mx, b, c = arithmetic.encode_gaussian(x, mean, scale, q, 10000)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    xd = arithmetic.decode_gaussian(mean, scale, q, mn, mx, b, c, 10000)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enc %.2f ms dec %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for rep in range(6):
    mn, mx, b, c = arithmetic.encode_gaussian(x, mean, scale, q, 10000)
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
