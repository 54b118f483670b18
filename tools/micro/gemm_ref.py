"""Reference point: vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the GEMM shapes the 3x3 convs reduce to
(M = B*H*W pixels, N = Cout, K = 9*Cin), fp16, for comparison with conv_mfma_kernel's TFLOP/s (tools only)."""
import torch
shapes = [('64@160^2', 819200, 64, 576), ('128@80^2', 204800, 128, 1152), ('256@40^2', 51200, 256, 2304),
          ('512@20^2', 12800, 512, 4608), ('256@20^2', 12800, 256, 2304), ('128@40^2', 51200, 128, 1152), ('64@80^2', 204800, 64, 576)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device='cuda', dtype=torch.float16)
    b = torch.randn(K, N, device='cuda', dtype=torch.float16)
    for _ in range(3): torch.matmul(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.matmul(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('%-10s M=%7d N=%4d K=%5d: %7.1f us  %7.1f TFLOP/s' % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
