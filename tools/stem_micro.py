import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphganformer_amd import _lib
from morphganformer_amd.lpips import PerceptualLoss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
P = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, allow_random_backbone=True)
x = (torch.rand(n, 3, 1024, 1024, device="cuda") * 2 - 1)
P.set_target(x[:1].contiguous())
f = P._features(n, 1024, 1024)
out = torch.zeros(n, device="cuda")
sc = torch.empty(n * int(_lib.lib().mgf_reduce_scratch_floats()), device="cuda")
def run():
    f.stem(x, feat_ref=P._target_taps[0], lin=P.lins[0], dist_out=out, scratch=sc)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"stem n={n}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
