import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wseg_amd import _lib as L
dev = "cuda"
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
# correctness (asymmetric data), incl. odd/even tile counts
for (M, N, K) in [(256, 256, 64), (512, 768, 192), (768, 256, 448), (256, 512, 128)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev)
    ref = A.float() @ B.float().t()
    for var in (0, 1):
        C.zero_(); L.gemm256_probe(A, B, C, M, N, K, var); torch.cuda.synchronize()
        err = float((C - ref).abs().max() / ref.abs().max())
        print("check", M, N, K, "variant", var, "max rel err", err, flush=True)
        assert err < 1e-2
for (M, N, K) in [(50176, 512, 4608), (50176, 2048, 9216), (50176, 4096, 2048), (8192, 8192, 8192), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev)*0.05).bfloat16()
    C = torch.empty(M, N, device=dev)
    for var in (0, 1, 0, 1):
        t = timeit(lambda: L.gemm256_probe(A, B, C, M, N, K, var))
        print(f"gemm256 v{var} {M}x{N}x{K}: {t:.3f} ms  {2.0*M*N*K/t/1e9:.1f} TF/s", flush=True)
