"""Yardstick, not product: what the vendor fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt) reaches on the GEMM shapes of the
mid-network convs (config 4: 4 frames, 1024^2), forward / backward-data form [M,K]x[K,N] and weight-gradient form
[K,M]x[M,N].  The convs are implicit GEMMs over NHWC taps, so this is an upper-ish bound for a plain-GEMM kernel of that size."""
import torch


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.backends.cuda.matmul.allow_tf32 = False
for name, M, N, K in [('L2.s1', 262144, 32, 128), ('L3.s1', 65536, 64, 256), ('L4.s1', 16384, 128, 512), ('L5.s1', 4096, 256, 1024),
                      ('L6.s1', 1024, 512, 2048), ('L8.s2', 4096, 512, 768)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(K, N, device='cuda'); g = torch.randn(M, N, device='cuda')
    gf = 2.0 * M * N * K / 1e9
    t1 = timeit(lambda: torch.mm(a, b))
    t2 = timeit(lambda: torch.mm(a.t(), g))
    t3 = timeit(lambda: torch.mm(g, b.t()))
    print('%-6s M %6d N %4d K %5d  %.2f GF   fwd %.1f us (%.0f TF)   wgrad-form %.1f us (%.0f TF)   dgrad-form %.1f us (%.0f TF)'
          % (name, M, N, K, gf, t1, gf / t1 * 1e3, t2, gf / t2 * 1e3, t3, gf / t3 * 1e3))
