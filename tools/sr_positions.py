"""The boxes of the short-range probes (tools/sr_dense_time.py, sr_dense_check.py, soak_p3m.py):
uniform, or bench.py's clustered box — 64 Gaussian clumps of sigma = L/40 with 80 % of the
particles."""
import torch


def positions(dist, n, L, gen):
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)
    if dist == 'clustered':
        centres = torch.rand((64, 3), dtype=torch.float64, device='cuda', generator=gen)*L
        which = torch.randint(0, 64, (n,), device='cuda', generator=gen)
        blob = centres[which] + torch.randn((n, 3), dtype=torch.float64, device='cuda',
                                            generator=gen)*(L/40)
        keep = torch.rand(n, dtype=torch.float64, device='cuda', generator=gen) < 0.2
        pos = torch.where(keep[:, None], pos*L, torch.remainder(blob, L))
        return pos.clamp_(0.0, L*(1 - 1e-13)).contiguous()
    return pos*(L*(1 - 1e-13))
