"""Host time of one sample draw (gh_torch_randperm_prefix) at bench sizes on this host.  python tools/draw_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphem_rapids_amd import _native
torch.manual_seed(0)
st = torch.get_rng_state().numpy().copy()
print("twist:", _native.torch_randperm_isa())
for E in (400000, 4000000, 16000000):
    _native.torch_randperm_prefix(st, E, 256, 3)
    t0 = time.perf_counter(); _native.torch_randperm_prefix(st, E, 256, 30); dt = (time.perf_counter() - t0) / 30
    print(E, f"{dt * 1e6:.1f} us per draw, {dt / E * 1e9:.4f} ns per word", flush=True)
