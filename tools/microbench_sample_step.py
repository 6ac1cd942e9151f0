"""One step of DiffusionLayer.sample (csrc/sample_step.hip) alone: us per launch at N rows x C, recorded into a HIP graph (the Python
wrapper costs more than the kernel).  Diagnostic builds (-DDGDM_STEP_DIAG=1|2|4: no weight loads / no GroupNorm arithmetic / no MFMAs)
through tools/run_with_lib.py say where the time goes.   python tools/microbench_sample_step.py [rows] [C]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops
from dgdm_histopath_lab_amd.core.diffusion import DiffusionLayer

rows_list = [int(r) for r in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10000]
C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda:0"
dl = DiffusionLayer(C, 2 * C, num_timesteps=10).to(dev).eval()
dn = dl.denoise_net
for rows in rows_list:
    x = torch.randn(rows, C, device=dev); z = torch.randn(rows, C, device=dev)
    bias0 = dl.time_bias(torch.tensor([3], device=dev))[0].contiguous()
    w0x = dn[0].weight[:, :C]
    fn = lambda: ops.denoise_ddpm_step(x, z, w0x, dn[4].weight, dn[8].weight, bias0, dn[1], dn[4].bias, dn[5], dn[8].bias, 0.5, 0.8, 0.9, 0.1, False)
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    print("rows %d C %d: %.1f us per launch (lib %s)" % (rows, C, a.elapsed_time(b) * 1e3 / 100, os.path.basename(os.path.dirname(ops._lib.LIB_PATH))))
