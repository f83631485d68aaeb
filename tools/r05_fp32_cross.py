"""fp32, kitchen: the small-batch path against the per-op kernels at 93 ... 768 samples (where the library hands over; GPU box)."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from bench import build_model
from beso_amd import _lib, synthetic as S
from beso_amd.runtime import set_plan
cfg = S.SHAPES["kitchen"]
m = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), "fp32", "cuda:0")
for B in (93, 128, 256, 372, 512, 768):
    s, g, a = (torch.from_numpy(v).to("cuda:0") for v in S.make_inputs(cfg, B, seed=1))
    sg = torch.full((B,), 0.3, device="cuda:0")
    row = []
    for hint in (_lib.PLAN_SMALL, _lib.PLAN_PER_OP, 0):
        set_plan(forward=hint)
        with torch.no_grad():
            for _ in range(20): m(s, a, g, sg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): m(s, a, g, sg)
            e1.record(); torch.cuda.synchronize()
        row.append(round(e0.elapsed_time(e1) * 20, 1))
    set_plan(forward=0)
    print(f"fp32 B={B}: small {row[0]} us   per-op {row[1]} us   library {row[2]} us", flush=True)
