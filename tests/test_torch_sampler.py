"""The host-side sampler of the parity mode (include/graphem_hip.h gh_torch_randperm_prefix): the ids AND the generator
state torch.randperm(E)[:S] produces (pt.py:409), from the mt19937 state alone -- checked against torch itself.
Pure host code of libgraphem_hip.so: no GPU, no oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def _native():
    from graphem_rapids_amd import _native as nat
    from graphem_rapids_amd import build as gra_build
    gra_build.build()
    return nat


@pytest.mark.parametrize("E", [5051, 400_000, 1 << 20, 4_000_000])
def test_ids_and_generator_state_equal_torch_randperm(E):
    nat = _native()
    S = 256
    torch.manual_seed(1234 + E % 7)
    torch.rand(3)                      # a generator part-way through a block, like any live one
    state = torch.get_rng_state().numpy().copy()
    done = 0
    for upto in (1, 2, 50):            # ids and state after 1, 2 and 50 draws
        ref = np.stack([torch.randperm(E)[:S].numpy() for _ in range(upto - done)])
        got = nat.torch_randperm_prefix(state, E, S, upto - done)
        done = upto
        assert got.dtype == np.int32 and np.array_equal(got, ref)
        assert np.array_equal(state, torch.get_rng_state().numpy()), f"generator state differs after {upto} draws"


@pytest.mark.parametrize("E,S", [(1, 1), (2, 1), (2, 2), (7, 3), (300, 299), (300, 300), (1000, 1000), (70_000, 65_536), (623, 5), (624, 5), (625, 5)])
def test_edge_sizes(E, S):
    nat = _native()
    torch.manual_seed(E * 31 + S)
    state = torch.get_rng_state().numpy().copy()
    ref = np.stack([torch.randperm(E)[:S].numpy() for _ in range(4)])
    got = nat.torch_randperm_prefix(state, E, S, 4)
    assert np.array_equal(got, ref)
    assert np.array_equal(state, torch.get_rng_state().numpy())


def test_every_twist_gives_the_same_words():
    """The vector twists (AVX2 / AVX-512, chosen at run time) against torch over many regenerated blocks, at every
    phase of a block: 700 single-entry draws of sizes around the block length."""
    nat = _native()
    assert nat.torch_randperm_isa() in ("avx512", "avx2", "scalar")
    torch.manual_seed(7)
    state = torch.get_rng_state().numpy().copy()
    for t in range(700):
        E = 2 + (t * 37) % 1400
        assert nat.torch_randperm_prefix(state, E, 1, 1)[0, 0] == int(torch.randperm(E)[0])
    assert np.array_equal(state, torch.get_rng_state().numpy())


def test_large_n_branch_against_the_torch_fixture():
    """n >= UINT32_MAX // 20: ATen switches to the inside-out shuffle on 64-bit draws (tests/golden/make_randperm_golden.py)."""
    nat = _native()
    g = load_golden("randperm_large")
    state = g["start_state"].copy()
    got = nat.torch_randperm_prefix(state, int(g["n"]), 256, 1)
    assert np.array_equal(got[0], g["ids"])
    assert np.array_equal(state, g["end_state"])
    # and the last n of the forward branch, against torch directly on a size it does quickly: covered above; the boundary
    # itself (n = UINT32_MAX // 20 - 1) costs torch 15 s and was checked when the fixture was made


def test_rejects_what_is_not_a_generator_state():
    nat = _native()
    with pytest.raises(ValueError):
        nat.torch_randperm_prefix(np.zeros(100, dtype=np.uint8), 10, 2, 1)
    bad = torch.get_rng_state().numpy().copy()
    bad[8:12] = 0     # left = 0 never occurs
    with pytest.raises(ValueError):
        nat.torch_randperm_prefix(bad, 10, 2, 1)
    with pytest.raises(ValueError):
        nat.torch_randperm_prefix(torch.get_rng_state().numpy().copy(), 10, 11, 1)   # S > n
