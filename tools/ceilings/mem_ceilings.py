"""Dev: achievable HBM rates on this box for pure read (sum), pure write (fill) and copy, 1 GiB buffers."""
import torch
dev = torch.device("cuda:0")
n = 1 << 28
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


GB = n * 4 / 1e12
print("fill  (write 1 GiB): %.2f TB/s" % (GB / t(lambda: b.fill_(1.0))))
print("zero  (memset)     : %.2f TB/s" % (GB / t(lambda: b.zero_())))
print("copy  (r + w)      : %.2f TB/s" % (2 * GB / t(lambda: b.copy_(a))))
print("sum   (read 1 GiB) : %.2f TB/s" % (GB / t(lambda: a.sum())))
c = torch.empty(n // 4, dtype=torch.float32, device=dev)
print("read 4 : write 1 (cat-like: b[:n/4] = a.view(4,-1).sum(0)): %.2f TB/s" % (1.25 * GB / t(lambda: torch.sum(a.view(4, -1), dim=0, out=c))))
