"""Library reference point for the four GEMM shapes of one DiT layer (torch.matmul -> hipBLASLt).
Not part of the product path: prints the rate the vendor library reaches so that the hand-written
kernel in csrc/gemm_bf16.hip can be judged against it (run under rocprofv3 to see which macro tiles
the library picks)."""
import torch
for (M, N, K) in [(4680, 4608, 1536), (4680, 8960, 1536), (4680, 1536, 8960), (4680, 1536, 1536)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, b.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        torch.matmul(a, b.t())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    print("hipblaslt ref", M, N, K, "%.1f us %.0f TF/s" % (us, 2 * M * N * K / us / 1e6))
