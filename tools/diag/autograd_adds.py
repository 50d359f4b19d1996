"""Which tensors does autograd sum in the train step?  torch.profiler over one bench step, aten::add / add_ events with their shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'timbre-trap_amd')):
    sys.path.insert(0, p)
import torch
import bench
from timbre_trap.utils import FusedAdamW
model = bench.build_model(2, 128, 'cuda')
opt = FusedAdamW(model.parameters(), lr=1e-4, max_norm=10.0)
step = bench.make_train_step(model, opt, 1)
audio, target = bench.synthetic_batch(64, 0, 'cuda')
for _ in range(2):
    step(audio, target)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(audio, target)
    torch.cuda.synchronize()
rows = [e for e in prof.events() if e.name in ('aten::add', 'aten::add_', 'aten::cat', 'aten::zeros', 'aten::zero_', 'aten::fill_', 'aten::mul', 'aten::copy_', 'aten::sum')]
for e in rows:
    dev = getattr(e, 'device_time_total', 0) or getattr(e, 'cuda_time_total', 0)
    if dev > 5:
        print('%-12s %8.1f us  %s' % (e.name, dev, e.input_shapes))
