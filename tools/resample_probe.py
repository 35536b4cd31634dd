"""Time of `resample_kernel` (sample_pdf, base_neural_render.py:27-115) alone at the bench's shapes, and a bit-exactness check of the same
call against the oracle on the first 4 096 rays.  GPU box, repository root:  python tools/resample_probe.py [> profiles/rNN_resample.txt]
NEDDF_LIB_PATH=<other libneddf_hip.so> in front of it gives the A/B partner in the same call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from neddf_amd import Context  # noqa: E402
from oracle import oracle  # noqa: E402

dev = torch.device("cuda:0")
ctx = Context.get(dev)
rng = np.random.default_rng(3)
print("library:", os.environ.get("NEDDF_LIB_PATH", "neddf_amd/csrc/libneddf_hip.so"))
for name, B, n, nf, cat in (("C3 800x800, 65 coarse + 129 fine merged to 194", 640000, 65, 129, True),
                            ("C5 1008x756, 65 + 129 -> 194", 762048, 65, 129, True),
                            ("129 + 257 -> 386", 200000, 129, 257, True),
                            ("33 + 60 -> 93", 640000, 33, 60, True),
                            ("65 coarse, 129 fine, not merged", 640000, 65, 129, False)):
    d = np.sort(rng.uniform(2, 6, (B, n)).astype(np.float32), axis=1)
    w = (rng.uniform(0, 1, (B, n - 1)) ** 6).astype(np.float32)
    u = rng.uniform(0, 1, (B, nf)).astype(np.float32)
    dd, uu = torch.from_numpy(d).to(dev), torch.from_numpy(u).to(dev)
    ww = torch.from_numpy(w).to(dev)
    out = ctx.importance_resample(dd, ww.clone(), uu, cat)
    oo, _, fb = oracle.sample_pdf(d[:4096], w[:4096].copy(), u[:4096], cat)
    exact = bool(np.array_equal(out[:4096].cpu().numpy(), oo)) and not fb
    ws = [ww.clone() for _ in range(12)]
    torch.cuda.synchronize()
    t = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.importance_resample(dd, ws[i], uu, cat)
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    ms = float(np.median(t[2:]))
    no = nf + n if cat else nf
    byts = B * 4 * ((n - 1) * 2 + n + nf + no)          # weights read + written back, dists, uniforms in; samples out
    print("%-48s %7d rays  %7.3f ms (call incl. flag memset + fallback pass)  %6.1f GB/s  %.2f ns/ray  bit-exact vs oracle (4 096 rays): %s"
          % (name, B, ms, byts / ms / 1e6, ms * 1e6 / B, exact))
