"""
ORACLE (test infrastructure only -- never imported by the product path).

One optimisation step of reference ``experiments/train.py:404-496`` on the CPU:
CQT (oracle/nsgt.py) -> autoencoder forward with consistency (oracle/autoencoder.py) ->
three losses (oracle/objectives.py) -> autograd backward -> clip_grad_norm_(10) (train.py:493)
-> AdamW(lr=1e-3, default betas/eps/weight_decay) (train.py:334, :496).

Used by tests (parity of the HIP train step) and by bench.py's ``cpu_baseline`` leg.
"""

import numpy as np
import torch

from . import autoencoder as ae
from . import nsgt
from . import objectives as obj


class OracleTrainer:
    def __init__(self, state_dict, lr=1e-3, feature_size=540, max_norm=10.0, dtype=torch.float32):
        self.params = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in state_dict.items()}
        self.optimizer = torch.optim.AdamW(list(self.params.values()), lr=lr)
        self.feature_size = feature_size
        self.max_norm = max_norm
        self.dtype = dtype

    def step(self, coefficients, ground_truth, multipliers=None):
        """coefficients (B,2,F,T), ground_truth (B_mpe,F,T) -> dict of float losses (+ grad norm)."""
        coefficients = coefficients.to(self.dtype)
        outputs = ae.forward(coefficients, self.params, consistency=True, feature_size=self.feature_size)
        total, parts = obj.total_loss(outputs, coefficients, ground_truth.to(self.dtype), multipliers)
        self.optimizer.zero_grad()
        total.backward()
        norm = torch.nn.utils.clip_grad_norm_(list(self.params.values()), self.max_norm)
        self.optimizer.step()
        out = {k: float(v.detach()) for k, v in parts.items()}
        out['grad_norm'] = float(norm)
        return out

    def grads(self):
        return {k: v.grad.detach().clone() for k, v in self.params.items()}


def cqt_forward_torch(audio, tab):
    """torch (B,1,n*N) float -> torch (B,2,F,n*M) float32 through the numpy oracle."""
    c = nsgt.wrapper_forward(audio.detach().cpu().numpy(), tab)
    return torch.from_numpy(np.ascontiguousarray(c)).to(torch.float32)
