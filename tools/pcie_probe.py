"""Raw host<->device copy rates of this box with pinned memory (what bounds the end_to_end block of bench.py):
python tools/pcie_probe.py"""
import time
import torch
n = 128 * 1241 * 376
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(3)]
ho = torch.empty(8323840, dtype=torch.uint8).pin_memory()
do = torch.empty(8323840, dtype=torch.uint8, device="cuda")
up, down = torch.cuda.Stream(), torch.cuda.Stream()
def run(K, both, pieces=1):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(up):
            if pieces == 1:
                d[i % 3].copy_(h, non_blocking=True)
            else:
                m = n // pieces
                for p in range(pieces):
                    d[i % 3][p * m:(p + 1) * m].copy_(h[p * m:(p + 1) * m], non_blocking=True)
        if both:
            with torch.cuda.stream(down):
                ho.copy_(do, non_blocking=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t
for both in (False, True):
    for pieces in (1, 2, 8):
        run(10, both, pieces)
        dt = run(100, both, pieces)
        print("H2D %d x %.1f MB in %d piece(s)%s: %.2f GB/s (%.3f ms per batch)" % (100, n / 1e6, pieces, " + D2H 8.3 MB alongside" if both else "", n * 100 / dt / 1e9, dt * 10))
