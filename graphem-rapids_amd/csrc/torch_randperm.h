// torch.randperm(E)[:S] without the other E - S entries (host code; no GPU involved).
//
// What the reference consumes per iteration (pt.py:409, `torch.randperm(n_edges, device='cpu')[:sample_size]`) is ATen's
// CPU randperm: r = arange(n); for i in 0 .. n-2: z = generator.random() % (n - i); swap(r[i], r[i + z])  (a FORWARD
// Fisher-Yates: after step i the entries 0 .. i are final), on the global CPU generator, an MT19937 whose state
// torch.get_rng_state() hands out as a 5056-byte blob.  Entries [:S] are therefore final after S draws; the other
// n - 1 - S draws only have to move the generator on.  This file restates the published algorithms (ATen
// aten/src/ATen/native/TensorFactories.cpp randperm_cpu; ATen/core/MT19937RNGEngine.h; CPUGeneratorImpl.cpp's legacy
// state layout) -- third-party dependency of the reference (torch >= 2.0), pinned by tests/test_torch_sampler.py against
// torch.randperm itself: ids AND the generator state afterwards.
#pragma once
#include <stdint.h>

#define GH_TORCH_RNG_STATE_BYTES 5056

struct gh_mt19937 {
    alignas(64) uint32_t s[624 + 16];   // 64-byte aligned (the AVX-512 twist loads whole lines); + slack: the vector loops read a few words past the last element
    int32_t left;
    uint32_t next;
    int32_t seeded;
    uint64_t seed;
};

// blob: the bytes of torch.get_rng_state() (CPUGeneratorImplState: legacy POD + the float normal cache).  false when the
// blob is not a state this code understands (wrong size fields).
bool gh_mt_load(gh_mt19937 *mt, const uint8_t *blob);
void gh_mt_store(const gh_mt19937 *mt, uint8_t *blob);   // writes seed / left / seeded / next / state back; the rest of the blob is kept
uint32_t gh_mt_draw(gh_mt19937 *mt);                     // at::mt19937::operator()
void gh_mt_skip(gh_mt19937 *mt, uint64_t draws);         // the state after that many operator() calls, none of them tempered
const char *gh_mt_isa();                                 // "avx512" / "avx2" / "scalar": the twist in use on this host

// One torch.randperm(n)[:S] (S <= n): consumes exactly the draws randperm(n) does.  scratch: caller-owned, gh_rp_scratch_words(S) words.
int64_t gh_rp_scratch_words(int64_t S);
void gh_torch_randperm_prefix_one(gh_mt19937 *mt, int64_t n, int64_t S, int32_t *out, int64_t *scratch);
