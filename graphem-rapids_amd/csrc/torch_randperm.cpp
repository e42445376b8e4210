// See torch_randperm.h.  Host-only translation unit of libgraphem_hip.so.
#include "torch_randperm.h"

#include <immintrin.h>
#include <string.h>

namespace {

constexpr int N = 624, M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UMASK = 0x80000000u, LMASK = 0x7fffffffu;

inline uint32_t twist(uint32_t u, uint32_t v) {
    return (((u & UMASK) | (v & LMASK)) >> 1) ^ ((v & 1u) ? MATRIX_A : 0u);
}

// at::mt19937::next_state(): words 0 .. 226 from the old words 397 .. 623, words 227 .. 622 from the new words 0 .. 395,
// word 623 from the new words 396 and 0.  Inside each range an element depends on nothing written fewer than 227
// elements before it, so the ranges vectorise at any width up to that.
void regen_scalar(uint32_t *p) {
    int j = 0;
    for (; j < N - M; ++j) p[j] = p[j + M] ^ twist(p[j], p[j + 1]);
    for (; j < N - 1; ++j) p[j] = p[j + M - N] ^ twist(p[j], p[j + 1]);
    p[N - 1] = p[M - 1] ^ twist(p[N - 1], p[0]);
}

__attribute__((target("avx2"))) inline void blk_avx2(uint32_t *p, int j, int off) {
    const __m256i um = _mm256_set1_epi32((int)UMASK), lm = _mm256_set1_epi32((int)LMASK), ma = _mm256_set1_epi32((int)MATRIX_A),
                  one = _mm256_set1_epi32(1), zero = _mm256_setzero_si256();
    const __m256i a = _mm256_loadu_si256((const __m256i *)(p + j)), b = _mm256_loadu_si256((const __m256i *)(p + j + 1)),
                  m = _mm256_loadu_si256((const __m256i *)(p + j + off));
    const __m256i y = _mm256_srli_epi32(_mm256_or_si256(_mm256_and_si256(a, um), _mm256_and_si256(b, lm)), 1);
    const __m256i mag = _mm256_and_si256(_mm256_sub_epi32(zero, _mm256_and_si256(b, one)), ma);
    _mm256_storeu_si256((__m256i *)(p + j), _mm256_xor_si256(m, _mm256_xor_si256(y, mag)));
}
__attribute__((target("avx2"))) void regen_avx2(uint32_t *p) {
    int j = 0;
    for (; j + 8 <= N - M; j += 8) blk_avx2(p, j, M);
    for (; j < N - M; ++j) p[j] = p[j + M] ^ twist(p[j], p[j + 1]);
    for (; j + 8 <= N - 1; j += 8) blk_avx2(p, j, M - N);
    for (; j < N - 1; ++j) p[j] = p[j + M - N] ^ twist(p[j], p[j + 1]);
    p[N - 1] = p[M - 1] ^ twist(p[N - 1], p[0]);
}

__attribute__((target("avx512f"))) void regen_avx512(uint32_t *p) {
    // 39 full vectors, no scalar remainder: word j takes  m_j = old p[j + 397] (j < 227) or new p[j - 227],  a_j = old p[j],
    // b_j = old p[j + 1] (new p[0] for j = 623).  a is an ALIGNED load carried over from the step before (the state is 64-byte
    // aligned), b = the same register shifted by one word with the next vector's first word (valignd): two loads per step
    // instead of three, none of the a / b loads across a line boundary.  Vector 14 (words 224 .. 239) straddles the two
    // ranges of m: a blend of both loads.  (The slack words behind the state are read, never used.)  The bit select
    // (a & UMASK) | (b & LMASK) is one vpternlogd, the conditional MATRIX_A a test-into-mask + masked xor.  On the GPU box's
    // host: 0.038 ns per word -- 153 us for the 4 M draws torch.randperm(4 M) spends -- against 0.057 with three loads per
    // step (229 us; the five-instead-of-nine ALU operations alone changed nothing: the loads set the pace).
    const __m512i um = _mm512_set1_epi32((int)UMASK), ma = _mm512_set1_epi32((int)MATRIX_A), one = _mm512_set1_epi32(1);
    __m512i a = _mm512_load_si512(p);
    for (int v = 0; v < 39; ++v) {
        const int j = 16 * v;
        __m512i an = _mm512_load_si512(p + j + 16);                       // old words j + 16 .. j + 31 (v = 38: slack)
        if (v == 38) an = _mm512_set1_epi32((int)p[0]);                  // word 623's partner is the NEW word 0
        const __m512i b = _mm512_alignr_epi32(an, a, 1);
        __m512i m;
        if (v < 14) m = _mm512_loadu_si512(p + j + M);
        else if (v > 14) m = _mm512_loadu_si512(p + j + M - N);
        else m = _mm512_mask_blend_epi32((__mmask16)0xFFF8, _mm512_loadu_si512(p + j + M), _mm512_loadu_si512(p + j + M - N + 0));
        const __m512i y = _mm512_ternarylogic_epi32(um, a, b, 0xCA);        // um ? a : b, bit by bit
        __m512i r = _mm512_xor_si512(m, _mm512_srli_epi32(y, 1));
        const __mmask16 odd = _mm512_test_epi32_mask(b, one);
        r = _mm512_mask_xor_epi32(r, odd, r, ma);
        _mm512_store_si512(p + j, r);
        a = an;
    }
}

using regen_fn = void (*)(uint32_t *);
struct isa_choice { regen_fn fn; const char *name; };
isa_choice choose_isa() {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f")) return {regen_avx512, "avx512"};
    if (__builtin_cpu_supports("avx2")) return {regen_avx2, "avx2"};
    return {regen_scalar, "scalar"};
}
const isa_choice &isa() {
    static const isa_choice c = choose_isa();
    return c;
}

inline void next_state(gh_mt19937 *mt) {
    isa().fn(mt->s);
    mt->left = N;
    mt->next = 0;
}

// CPUGeneratorImplStateLegacy (aten/src/ATen/CPUGeneratorImpl.cpp): uint64 the_initial_seed; int left; int seeded;
// uint64 next; uint64 state[624]; double normal_x, normal_y, normal_rho; int normal_is_valid; -- then, in
// CPUGeneratorImplState, float next_float_normal_sample; bool is_next_float_normal_sample_valid.
constexpr int OFF_SEED = 0, OFF_LEFT = 8, OFF_SEEDED = 12, OFF_NEXT = 16, OFF_STATE = 24;

}  // namespace

const char *gh_mt_isa() { return isa().name; }

bool gh_mt_load(gh_mt19937 *mt, const uint8_t *blob) {
    uint64_t next, w;
    memcpy(&mt->seed, blob + OFF_SEED, 8);
    memcpy(&mt->left, blob + OFF_LEFT, 4);
    memcpy(&mt->seeded, blob + OFF_SEEDED, 4);
    memcpy(&next, blob + OFF_NEXT, 8);
    if (mt->left < 1 || mt->left > N || next > (uint64_t)N) return false;
    mt->next = (uint32_t)next;
    for (int i = 0; i < N; ++i) {
        memcpy(&w, blob + OFF_STATE + 8 * i, 8);
        if (w >> 32) return false;
        mt->s[i] = (uint32_t)w;
    }
    memset(mt->s + N, 0, sizeof(mt->s) - N * sizeof(uint32_t));
    return true;
}

void gh_mt_store(const gh_mt19937 *mt, uint8_t *blob) {
    const uint64_t next = mt->next;
    memcpy(blob + OFF_SEED, &mt->seed, 8);
    memcpy(blob + OFF_LEFT, &mt->left, 4);
    memcpy(blob + OFF_SEEDED, &mt->seeded, 4);
    memcpy(blob + OFF_NEXT, &next, 8);
    for (int i = 0; i < N; ++i) {
        const uint64_t w = mt->s[i];
        memcpy(blob + OFF_STATE + 8 * i, &w, 8);
    }
}

uint32_t gh_mt_draw(gh_mt19937 *mt) {
    if (--mt->left == 0) next_state(mt);
    uint32_t y = mt->s[mt->next++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// `left` counts the draws up to and including the one that regenerates the block: left - 1 more words of this block.
void gh_mt_skip(gh_mt19937 *mt, uint64_t k) {
    if (k == 0) return;
    if (k < (uint64_t)mt->left) { mt->left -= (int32_t)k; mt->next += (uint32_t)k; return; }
    k -= (uint64_t)(mt->left - 1);         // the rest of this block; the next draw regenerates
    while (k > (uint64_t)N) { isa().fn(mt->s); k -= N; }
    isa().fn(mt->s);                       // 1 <= k <= 624 words of this block are consumed
    mt->next = (uint32_t)k;
    mt->left = N + 1 - (int32_t)k;
}

// scratch: an open-addressing table for the entries >= S that a swap moved (at most S of them): (key, value) pairs
int64_t gh_rp_scratch_words(int64_t S) {
    int64_t cap = 16;
    while (cap < 4 * S) cap <<= 1;
    return 2 * cap + 1;
}

void gh_torch_randperm_prefix_one(gh_mt19937 *mt, int64_t n, int64_t S, int32_t *out, int64_t *scratch) {
    if (n <= 0) return;
    if (S > n) S = n;
    // ATen: "for small n, preserve old behavior" -- the forward shuffle on 32-bit draws below UINT32_MAX / 20
    if (n >= (int64_t)(UINT32_MAX / 20)) {
        // ... and from there on the inside-out Fisher-Yates on 64-bit draws (two words, the first one the high half):
        // for i in 0 .. n-1: z = random64() % (i + 1); r[i] = r[z]; r[z] = i.  An entry below S is only ever replaced by
        // the current i, and what leaves the prefix never returns: after the first S steps, out[z] = i whenever z < S.
        // Every draw has to be valued here (nothing to skip): ~15 ns each.
        for (int64_t i = 0; i < n; ++i) {
            const uint64_t hi = gh_mt_draw(mt), lo = gh_mt_draw(mt);
            const int64_t z = (int64_t)(((hi << 32) | lo) % (uint64_t)(i + 1));
            if (i < S) { out[i] = out[z]; out[z] = (int32_t)i; }   // (z == i: out[i] = i)
            else if (z < S) out[z] = (int32_t)i;
        }
        return;
    }
    int64_t cap = 16;
    while (cap < 4 * S) cap <<= 1;
    int64_t *keys = scratch, *vals = scratch + cap;
    for (int64_t i = 0; i < cap; ++i) keys[i] = -1;
    for (int64_t i = 0; i < S; ++i) out[i] = (int32_t)i;
    const int64_t swaps = S < n - 1 ? S : n - 1;
    for (int64_t i = 0; i < swaps; ++i) {
        const int64_t j = i + (int64_t)(gh_mt_draw(mt) % (uint64_t)(n - i));
        if (j == i) continue;
        const int32_t vi = out[i];
        if (j < S) { out[i] = out[j]; out[j] = vi; continue; }
        int64_t slot = (int64_t)(((uint64_t)j * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
        while (keys[slot] != -1 && keys[slot] != j) slot = (slot + 1) & (cap - 1);
        out[i] = keys[slot] == j ? (int32_t)vals[slot] : (int32_t)j;
        keys[slot] = j;
        vals[slot] = vi;
    }
    gh_mt_skip(mt, (uint64_t)(n - 1 - swaps));
}
