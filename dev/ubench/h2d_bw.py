"""H2D bandwidth from pinned host memory against the chunk size of the copy (one stream, back-to-back chunks), and with two streams.   gpurun -- 'python dev/ubench/h2d_bw.py'"""
import time, torch
dev = torch.device("cuda:0")
tot = 1 << 30
h = torch.empty(tot, dtype=torch.uint8).pin_memory(); h.fill_(1)
d = torch.empty(tot, dtype=torch.uint8, device=dev)
for chunk_mb in (1, 4, 16, 64, 256, 1024):
    c = chunk_mb << 20
    for n_st in (1, 2):
        sts = [torch.cuda.Stream() for _ in range(n_st)]
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i, off in enumerate(range(0, tot, c)):
                with torch.cuda.stream(sts[i % n_st]):
                    d[off:off + c].copy_(h[off:off + c], non_blocking=True)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"chunk {chunk_mb:5d} MB, {n_st} stream(s): {tot / best / 1e9:6.1f} GB/s")
