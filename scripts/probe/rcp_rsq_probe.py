#!/usr/bin/env python3
"""Accuracy of the gfx950 v_rcp_f64 / v_rsq_f64 seeds and of the Newton / Goldschmidt steps the kernels build on them
(afhip_numerics.h: rcp_newton1; afhip_sine.h: sine_arc).  Compiles a scratch kernel with hipcc at run time; prints max relative errors."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np
import torch

SRC = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void k(const double* x, double* o, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y = __builtin_amdgcn_rcp(v);
    o[i] = y;
    double e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y);
    o[n + i] = y;
    e = __builtin_fma(-v, y, 1.0); y = __builtin_fma(y, e, y);
    o[2 * n + i] = y;
    double r = __builtin_amdgcn_rsq(v);
    o[3 * n + i] = r;
    double g = v * r, h = 0.5 * r;
    double t = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, t, g); h = __builtin_fma(h, t, h);
    o[4 * n + i] = g;                                   // sqrt after the coupled step
    double d = __builtin_fma(-g, g, v);
    g = __builtin_fma(d, h, g);
    o[5 * n + i] = g;                                   // + one residual correction
    d = __builtin_fma(-g, g, v);
    g = __builtin_fma(d, h, g);
    o[6 * n + i] = g;                                   // + the second
}
extern "C" void launch(const double* x, double* o, long n, void* st) {
    hipLaunchKernelGGL(k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, x, o, n);
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "p.hip"), "w").write(SRC)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                       os.path.join(d, "p.hip"), "-o", os.path.join(d, "p.so")])
lib = ctypes.CDLL(os.path.join(d, "p.so"))
rng = np.random.default_rng(3)
n = 1 << 22
x = np.concatenate([rng.uniform(1e-3, 60, n // 2), np.exp(rng.uniform(-36, 5, n // 2))])
xt = torch.from_numpy(x).cuda()
o = torch.empty(7 * n, dtype=torch.float64, device="cuda")
lib.launch(ctypes.c_void_p(xt.data_ptr()), ctypes.c_void_p(o.data_ptr()), ctypes.c_long(n), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = o.cpu().numpy().reshape(7, n)
xl = x.astype(np.longdouble)
rcp, sq = 1 / xl, np.sqrt(xl)
names = ["v_rcp_f64", "rcp + 1 Newton", "rcp + 2 Newton", "v_rsq_f64", "sqrt: coupled step", "sqrt: + 1 correction", "sqrt: + 2 corrections"]
refs = [rcp, rcp, rcp, 1 / sq, sq, sq, sq]
for nm, got, ref in zip(names, o, refs):
    rel = np.abs((got.astype(np.longdouble) - ref) / ref).astype(np.float64)
    print(f"{nm:24s} max rel err {rel.max():.3e} = 2^{np.log2(rel.max()):.1f}   (ulp = 2^-52 = 2.2e-16: {rel.max() / 2.0 ** -52:.2f} ulp)")
