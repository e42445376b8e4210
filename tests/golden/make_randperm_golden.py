"""Fixture for ATen's large-n branch of the CPU randperm (n >= UINT32_MAX // 20: inside-out shuffle on 64-bit draws):
the first 256 entries of torch.randperm(n) and the generator state it leaves, from torch itself (20 s, 1.7 GB -- too
heavy for the CPU suite, hence a fixture).  `python tests/golden/make_randperm_golden.py` rewrites randperm_large.npz."""
import hashlib
import os

import numpy as np
import torch

n = (2 ** 32 - 1) // 20   # the first n of the branch
torch.manual_seed(5)
start = torch.get_rng_state().numpy().copy()
ids = torch.randperm(n)[:256].numpy().astype(np.int64)
end = torch.get_rng_state().numpy()
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "randperm_large.npz")
np.savez_compressed(out, n=np.int64(n), seed=np.int64(5), ids=ids, start_state=start, end_state=end,
                    torch_version=np.array(torch.__version__))
print(out, hashlib.sha1(end.tobytes()).hexdigest())
