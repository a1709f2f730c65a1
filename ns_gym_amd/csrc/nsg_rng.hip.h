// nsg_rng.hip.h — NumPy-compatible bit streams on gfx950 (device code).
//
// The reference draws every random number through NumPy: env.np_random (gymnasium seeding,
// called at ns_gym/base.py:377) and one `np.random.default_rng` per stochastic update fn
// (ns_gym/update_functions/single_param.py:76,108,146,342,444), re-seeded by
// SeedSequence(seed).spawn(n) in NSWrapper._seed_update_fns (ns_gym/base.py:412-421).
// Seed-matched trajectory parity therefore needs NumPy's exact algorithms [UPSTREAM numpy,
// uv.lock:1868-1869]: SeedSequence hash-mix -> PCG64 (128-bit LCG, XSL-RR output) ->
// next_double = (u64 >> 11) * 2^-53 -> 256-layer ziggurat normal.
//
// Layout: a stream is one 32-byte record per env ([N][4] u64: state_hi, state_lo, inc_hi, inc_lo).
// The ziggurat tables (6 KiB) are staged in LDS once per workgroup: lookups are per-lane
// random indices, which LDS serves at full rate and HBM/L2 would not.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "nsg_math.hip.h"
#include "nsgym_hip.h"

namespace nsg {

// Global-memory accessors: base pointer wave-uniform (SGPR pair), per-lane 32-bit BYTE offset.
// The buffers reach the kernels through a struct in memory, so without the explicit address
// space the compiler must emit flat_* instructions and 64-bit per-lane address arithmetic;
// with it, every access is `global_load/store vdata, voffset32, s[base:base+1]`.
#define NSG_GLOBAL __attribute__((address_space(1)))
// Write-once outputs are stored non-temporally: they should not displace the state rows, which the next launch
// reads again, from the caches (measured: C1 28.95 -> 28.39 us, Pendulum 21.37 -> 21.19 us).  Non-temporal LOADS of
// the action row were tried too and cost 5 us (C1 33.5 us): the row usually is cache-resident, written just before.
#ifndef NSG_NT_OUTPUTS
#define NSG_NT_OUTPUTS 1
#endif
// readfirstlane declares the row base wave-uniform (it is: every lane computes it from the same
// kernel-uniform values), which keeps it in an SGPR pair instead of 64-bit per-lane arithmetic.
__device__ __forceinline__ uint64_t pin_sgpr(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
template <typename T> __device__ __forceinline__ T ldg(const T* base, uint32_t byte_off) {
  const NSG_GLOBAL char* p = (const NSG_GLOBAL char*)pin_sgpr((uint64_t)base);
  return *(const NSG_GLOBAL T*)(p + byte_off);
}
// NSG_STREAM_STATE 1 (specialised units of batches whose rows cannot live in the 256-MiB Infinity Cache, nsg_specialize): the
// persistent rows leave non-temporally too - nothing written by this launch is still cached when the next one reads it anyway.
#ifndef NSG_STREAM_STATE
#define NSG_STREAM_STATE 0
#endif
template <typename T> __device__ __forceinline__ void stg(T* base, uint32_t byte_off, T v) {
  NSG_GLOBAL char* p = (NSG_GLOBAL char*)pin_sgpr((uint64_t)base);
#if NSG_STREAM_STATE
  __builtin_nontemporal_store(v, (NSG_GLOBAL T*)(p + byte_off));
#else
  *(NSG_GLOBAL T*)(p + byte_off) = v;
#endif
}
// Persistent rows of the grid envs: agent-scope stores (`global_store ... sc1`: written through the XCD's L2 instead of
// sitting there dirty until something evicts them).  Measured on C3 (FrozenLake 8x8, 2^20 envs, two interleaved
// repetitions): 20.7 / 20.1 -> 19.4 / 18.7 us.  The classic-control kernels LOSE with the same policy (C1 +1 %, C2 +2.5 %)
// and keep plain write-back stores.  NSG_GRID_WT=0 restores them here.
#ifndef NSG_GRID_WT
#define NSG_GRID_WT 1
#endif
template <typename T> __device__ __forceinline__ void stg_wt(T* base, uint32_t byte_off, T v) {
#if NSG_GRID_WT
  NSG_GLOBAL char* p = (NSG_GLOBAL char*)pin_sgpr((uint64_t)base);
  __hip_atomic_store((NSG_GLOBAL T*)(p + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  stg(base, byte_off, v);
#endif
}

// write-once outputs (obs, reward, flags, deltas): nothing on the device reads them again
template <typename T> __device__ __forceinline__ void stg_out(T* base, uint32_t byte_off, T v) {
#if NSG_NT_OUTPUTS
  NSG_GLOBAL char* p = (NSG_GLOBAL char*)pin_sgpr((uint64_t)base);
  __builtin_nontemporal_store(v, (NSG_GLOBAL T*)(p + byte_off));
#else
  stg(base, byte_off, v);
#endif
}

// Chunk-blocked row groups: [ceil(N / 256)][R][256] 8-byte elements - the R rows of a workgroup's 256 envs are ONE
// contiguous run instead of R runs megabytes apart (integrator state `phys`: 8 KB for CartPole; grid envs: the four
// env-stream rows and the table probabilities).  Measured on the C1 step: 27.7 -> 26.1 us (2^20 envs), 110 -> 101 us
// (2^22).  Byte offset of element (row k, env i); fits 32 bits for N <= 2^27, R <= 4.
__host__ __device__ inline uint32_t blk_off8(int R, int k, int64_t i) {
  return (uint32_t)((((uint64_t)i >> 8) * (uint64_t)(R * 256) + (uint64_t)k * 256 + ((uint64_t)i & 255)) * 8);
}

struct Pcg {
  uint64_t sh, sl, ih, il;
};

__device__ __forceinline__ void pcg_step(Pcg& r) {
  // state = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc  (mod 2^128), 64-bit limbs
  const uint64_t MH = 2549297995355413924ULL, ML = 4865540595714422341ULL;
  uint64_t lo = r.sl * ML;
  uint64_t hi = __umul64hi(r.sl, ML) + r.sh * ML + r.sl * MH;
  uint64_t lo2 = lo + r.il;
  uint64_t carry = lo2 < lo ? 1ULL : 0ULL;
  r.sl = lo2;
  r.sh = hi + r.ih + carry;
}

__device__ __forceinline__ uint64_t pcg_next64(Pcg& r) {
  pcg_step(r);
  uint64_t x = r.sh ^ r.sl;
  unsigned rot = (unsigned)(r.sh >> 58);
  return (x >> rot) | (x << ((64u - rot) & 63u));
}

// output function alone: the draw whose step has already been taken (pcg_at<2>)
__device__ __forceinline__ uint64_t pcg_out64(const Pcg& r) {
  uint64_t x = r.sh ^ r.sl;
  unsigned rot = (unsigned)(r.sh >> 58);
  return (x >> rot) | (x << ((64u - rot) & 63u));
}
__device__ __forceinline__ double pcg_double_out(const Pcg& r) {
  return (double)(pcg_out64(r) >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double pcg_double(Pcg& r) {
  return (double)(pcg_next64(r) >> 11) * (1.0 / 9007199254740992.0);
}

// ---- SeedSequence(entropy=seed, spawn_key=(child,) or ()).generate_state(4, uint64) ----
__device__ __forceinline__ uint32_t ss_hashmix(uint32_t v, uint32_t& hc) {
  v ^= hc;
  hc *= 0x931e8875u;
  v *= hc;
  v ^= v >> 16;
  return v;
}
__device__ __forceinline__ uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
  r ^= r >> 16;
  return r;
}

// PCG64(SeedSequence(seed[, spawn_key=(child,)])) one step BEFORE its initial state: r.s = T0 = inc + initstate, r.i = inc.
// pcg64_srandom_r is "state = 0; step; state += initstate; step", i.e. S_0 = step(T0), and the state after k draws is
// step^(k+1)(T0): pcg_at folds that last seeding step into its jump.
__device__ inline void pcg_seed_t0(Pcg& r, uint64_t seed, int child) {
  uint32_t e0 = (uint32_t)seed, e1 = (uint32_t)(seed >> 32);
  // entropy words; with a spawn key the entropy is zero-padded to the pool size (4) first
  uint32_t p0, p1, p2, p3, hc = 0x43b0d7e5u;
  p0 = ss_hashmix(e0, hc);
  p1 = ss_hashmix(e1, hc);  // e1 == 0 when seed < 2^32: identical to "word absent"
  p2 = ss_hashmix(0u, hc);
  p3 = ss_hashmix(0u, hc);
  // mix all bits together so late bits can affect earlier bits (src-major order)
  p1 = ss_mix(p1, ss_hashmix(p0, hc)); p2 = ss_mix(p2, ss_hashmix(p0, hc)); p3 = ss_mix(p3, ss_hashmix(p0, hc));
  p0 = ss_mix(p0, ss_hashmix(p1, hc)); p2 = ss_mix(p2, ss_hashmix(p1, hc)); p3 = ss_mix(p3, ss_hashmix(p1, hc));
  p0 = ss_mix(p0, ss_hashmix(p2, hc)); p1 = ss_mix(p1, ss_hashmix(p2, hc)); p3 = ss_mix(p3, ss_hashmix(p2, hc));
  p0 = ss_mix(p0, ss_hashmix(p3, hc)); p1 = ss_mix(p1, ss_hashmix(p3, hc)); p2 = ss_mix(p2, ss_hashmix(p3, hc));
  if (child >= 0) {  // fifth entropy word = the spawn key
    uint32_t k = (uint32_t)child;
    p0 = ss_mix(p0, ss_hashmix(k, hc));
    p1 = ss_mix(p1, ss_hashmix(k, hc));
    p2 = ss_mix(p2, ss_hashmix(k, hc));
    p3 = ss_mix(p3, ss_hashmix(k, hc));
  }
  uint32_t hb = 0x8b51f9ddu, w[8];
  const uint32_t pool[4] = {p0, p1, p2, p3};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t v = pool[i & 3];
    v ^= hb;
    hb *= 0x58f38dedu;
    v *= hb;
    v ^= v >> 16;
    w[i] = v;
  }
  uint64_t s_hi = (uint64_t)w[0] | ((uint64_t)w[1] << 32), s_lo = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
  uint64_t q_hi = (uint64_t)w[4] | ((uint64_t)w[5] << 32), q_lo = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
  // pcg64_srandom_r: state = 0; inc = (initseq << 1) | 1; step (-> state = inc); state += initstate; [step: the caller's]
  r.ih = (q_hi << 1) | (q_lo >> 63);
  r.il = (q_lo << 1) | 1ULL;
  const uint64_t lo = r.il + s_lo;
  r.sh = r.ih + s_hi + (lo < r.il ? 1ULL : 0ULL);
  r.sl = lo;
}

// Seeds PCG64 exactly like np.random.PCG64(SeedSequence(seed[, spawn_key=(child,)])).
__device__ inline void pcg_seed(Pcg& r, uint64_t seed, int child) {
  pcg_seed_t0(r, seed, child);
  pcg_step(r);
}

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// ---- PCG64 jump-ahead: the stream of env i at draw n, without ever storing the stream ---------------------------------
// A PCG64 step is S <- S * M + inc (mod 2^128), so after n steps S_n = A_n * S_0 + inc * G_n with A_n = M^n and
// G_n = 1 + M + ... + M^(n-1) - both independent of the stream.  A_n, G_n for an arbitrary n < 2^40 come from a table over the
// 8-bit digits of n - a digit-0 block of kJumpLow entries (entry e holds the pair for the exponent e itself, a little past
// 255: see LEAD below) followed by 4 blocks of 256 (entry (d, v) holds the pair for the exponent v * 256^d); built on the host
// in 128-bit integer arithmetic, nsgym_hip.hip - by composing one entry per non-zero digit:
//     exponents a then b:   A_(a+b) = A_a * A_b,   G_(a+b) = G_a * A_b + G_b.
// The classic-control envs draw from env.np_random only in reset() (D doubles per reset: CartPole / Acrobot 4, Pendulum 2,
// MountainCar 1), so the initial state of episode e of env i is draws [D*e, D*e + D) of PCG64(SeedSequence(seed_i)): a pure
// function of (seed_i, e).  The step kernels therefore keep NO stream record per env - only the episode count e, in the env's
// dense episode word - and the helper lanes of the reset hand-over rebuild the stream where they need it.  Measured on MI355X
// (C1, specialised kernels; record loads / stores compiled out, everything else in place): 2^20 envs 25.2 -> 22.1 us,
// 2^22 envs 105.9 -> 80.7 us, 2^24 envs 515 -> 423 us: the 32-byte records, touched by the ~5 % of lanes whose env resets,
// cost more than any other part of the step (tools/stream_probe.hip reproduces it on a bare skeleton).
constexpr int kJumpDigits = 5;                       // n < 2^40 draws per stream
constexpr int kJumpLow = 264;                        // digit 0 block: exponents 0 .. 263 (255 + the lead of 1 or 2 steps, padded)
constexpr int kJumpWords = (kJumpLow + (kJumpDigits - 1) * 256) * 4;   // u64 words: (A_hi, A_lo, G_hi, G_lo) per entry
static_assert(kJumpLow == NSG_JUMP_LOW && kJumpWords == NSG_JUMP_TABLE_WORDS, "nsgym_hip.h and the kernels disagree on the jump table");

struct U128 {
  uint64_t hi, lo;
};
__host__ __device__ __forceinline__ U128 mul128(U128 a, U128 b) {  // low 128 bits of a * b
#ifdef __HIP_DEVICE_COMPILE__
  const uint64_t hi = __umul64hi(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi;
#else
  const uint64_t hi = (uint64_t)(((unsigned __int128)a.lo * b.lo) >> 64) + a.hi * b.lo + a.lo * b.hi;
#endif
  return U128{hi, a.lo * b.lo};
}
__host__ __device__ __forceinline__ U128 add128(U128 a, U128 b) {
  const uint64_t lo = a.lo + b.lo;
  return U128{a.hi + b.hi + (lo < a.lo ? 1ULL : 0ULL), lo};
}

// The stream seeded like PCG64(SeedSequence(seed[, spawn_key = (child,)])) at draw n.  LEAD = 1: the state BEFORE draw n (what the
// generator object holds after n draws; pcg_next64 continues from it).  LEAD = 2: the state after draw n's own step - draw n is
// pcg_out64 of it, no further multiply.  Both are step^(n + LEAD)(T0) = A_(n+LEAD) * T0 + inc * G_(n+LEAD) with T0 from
// pcg_seed_t0: the table's digit-0 block is indexed by (n mod 256) + LEAD, so the seeding's last step costs nothing.
// `jump`: the table above (global memory; 41 KB, shared by every env, cache-resident).  The digit loop is wave-uniform (it runs
// while ANY lane has a non-zero digit left); a typical n (a few thousand episodes) takes two digits.
template <int LEAD = 1>
__device__ inline void pcg_at(Pcg& r, uint64_t seed, int child, uint64_t n, const uint64_t* __restrict__ jump) {
  static_assert(LEAD >= 0 && 255 + LEAD < kJumpLow, "digit-0 block too small");
  pcg_seed_t0(r, seed, child);
  U128 A, G;
  {
    const u64x2* e = reinterpret_cast<const u64x2*>(jump + (size_t)((unsigned)(n & 255u) + (unsigned)LEAD) * 4);
    const u64x2 a = e[0], g = e[1];
    A = U128{a.x, a.y};
    G = U128{g.x, g.y};
  }
  // one composition per further non-zero digit
  uint64_t rest = n >> 8;
  for (int d = 1; d < kJumpDigits; d++) {
    if (__ballot(rest != 0) == 0) break;
    const unsigned v = (unsigned)(rest & 255u);
    rest >>= 8;
    if (v != 0) {
      const u64x2* e = reinterpret_cast<const u64x2*>(jump + ((size_t)kJumpLow + (size_t)(d - 1) * 256 + v) * 4);
      const u64x2 a = e[0], g = e[1];
      const U128 Ad = {a.x, a.y}, Gd = {g.x, g.y};
      G = add128(mul128(G, Ad), Gd);
      A = mul128(A, Ad);
    }
  }
  const U128 s = add128(mul128(A, U128{r.sh, r.sl}), mul128(G, U128{r.ih, r.il}));
  r.sh = s.hi;
  r.sl = s.lo;
}

// A stream is one 32-byte record per env, array-of-records [N][4]: state_hi, state_lo, inc_hi,
// inc_lo.  Streams are touched by few, scattered lanes (the ~5 % of envs that reset, the lanes
// whose scheduler fired), so one 32-byte record = one memory sector per touch; four SoA rows
// would cost four sectors.  Two 16-byte accesses per lane; only the state half is written back.
// PCG64 stream storage.  Two layouts, chosen by how a stream is touched:
//   records (default): [N][4] u64, one 32-byte record per env - update-fn / scheduler streams and the
//     env streams of the classic-control envs, which only a few scattered lanes touch per step
//     (resets, fires): one sector per touch instead of four rows;
//   rows (ROWS = true): chunk-blocked [ceil(N/256)][4][256] u64 - the env streams of the grid envs, where EVERY lane
//     draws one uniform per step: four fully coalesced 8-byte rows instead of 16-byte accesses at a 32-byte
//     stride (FrozenLake-shaped skeleton, tools/layout_probe.hip: 19.1 us vs 23.0 us per 2^20 envs).
template <typename T> __device__ __forceinline__ T ldg_nt(const T* base, uint32_t byte_off) {
  const NSG_GLOBAL char* p = (const NSG_GLOBAL char*)pin_sgpr((uint64_t)base);
  return __builtin_nontemporal_load((const NSG_GLOBAL T*)(p + byte_off));
}
template <bool ROWS = false>
__device__ __forceinline__ void pcg_load(const uint64_t* base, int64_t N, int64_t i, Pcg& r) {
  if constexpr (ROWS) {
    r.sh = ldg(base, blk_off8(4, 0, i)); r.sl = ldg(base, blk_off8(4, 1, i));
    r.ih = ldg(base, blk_off8(4, 2, i)); r.il = ldg(base, blk_off8(4, 3, i));
  } else {
    const uint32_t o = (uint32_t)i * 32u;
    const u64x2 a = ldg(reinterpret_cast<const u64x2*>(base), o);
    const u64x2 c = ldg(reinterpret_cast<const u64x2*>(base), o + 16u);
    r.sh = a.x; r.sl = a.y; r.ih = c.x; r.il = c.y;
  }
}
// Stream of env i of a classic-control batch, positioned at draw n (buffers.rng_env, include/nsgym_hip.h): record 0 is the
// batch's descriptor - affine (env i seeded base + i; nothing else is read) or not (record 1 + i holds the env's own seed and
// spawn key: one 16-byte read by the few lanes that reset).
template <int LEAD = 1>
__device__ __forceinline__ void env_stream_at(const uint64_t* rng_env, int64_t i, uint64_t n, const uint64_t* jump, Pcg& r,
                                              const u64x2* desc = nullptr) {
  const u64x2 d = desc ? *desc : *reinterpret_cast<const u64x2*>(rng_env);
  uint64_t seed;
  int key;
  if (d.x & NSG_STREAM_AFFINE) {
    seed = d.y + (uint64_t)i;
    key = (int)(uint32_t)d.x;
  } else {
    const u64x2 rec = ldg(reinterpret_cast<const u64x2*>(rng_env), (uint32_t)(i + 1) * 16u);
    seed = rec.x;
    key = (int)(uint32_t)rec.y;
  }
  pcg_at<LEAD>(r, seed, key, n, jump);
}
__device__ __forceinline__ void env_stream_set_affine(uint64_t* rng_env, uint64_t base, int key) {
  rng_env[0] = NSG_STREAM_AFFINE | (uint64_t)(uint32_t)key;
  rng_env[1] = base;
}
template <bool ROWS = false, bool WT = false>
__device__ __forceinline__ void pcg_store_state(uint64_t* base, int64_t N, int64_t i, const Pcg& r) {
  if constexpr (ROWS && WT) {
    stg_wt(base, blk_off8(4, 0, i), r.sh); stg_wt(base, blk_off8(4, 1, i), r.sl);
  } else if constexpr (ROWS) {
    stg(base, blk_off8(4, 0, i), r.sh); stg(base, blk_off8(4, 1, i), r.sl);  // the increment never changes
  } else {
    stg(reinterpret_cast<u64x2*>(base), (uint32_t)i * 32u, u64x2{r.sh, r.sl});
  }
}
template <bool ROWS = false>
__device__ __forceinline__ void pcg_store_all(uint64_t* base, int64_t N, int64_t i, const Pcg& r) {
  if constexpr (ROWS) {
    stg(base, blk_off8(4, 0, i), r.sh); stg(base, blk_off8(4, 1, i), r.sl);
    stg(base, blk_off8(4, 2, i), r.ih); stg(base, blk_off8(4, 3, i), r.il);
  } else {
    const uint32_t o = (uint32_t)i * 32u;
    stg(reinterpret_cast<u64x2*>(base), o, u64x2{r.sh, r.sl});
    stg(reinterpret_cast<u64x2*>(base), o + 16u, u64x2{r.ih, r.il});
  }
}
// env stream of env type `grid ? rows : records`, for kernels that take the env type at run time
__device__ __forceinline__ void pcg_store_env(uint64_t* base, int64_t N, int64_t i, const Pcg& r, bool grid) {
  if (grid) pcg_store_all<true>(base, N, i, r);
  else pcg_store_all<false>(base, N, i, r);
}

// ---- ziggurat normal (numpy random_standard_normal), tables in LDS --------------------
struct ZigLds {
  const uint64_t* ki;   // normal ziggurat
  const double* wi;
  const double* fi;
  const uint64_t* ke;   // exponential ziggurat (geometric inversion, Dirichlet)
  const double* we;
  const double* fe;
  const uint64_t* jump = nullptr;  // PCG64 jump-ahead table (global memory), pcg_at
  uint64_t sd0 = 0, sd1 = 0;       // classic-control envs: the batch's stream descriptor (buffers.rng_env[0..1]), fetched once per
                                   // workgroup next to the table staging so that the reset hand-over does not wait for it
};

__device__ inline double pcg_std_normal(Pcg& g, const ZigLds& z) {
  const double ZR = 3.6541528853610087963519472518, ZINV = 0.27366123732975827203338247596;
  for (;;) {
    uint64_t r = pcg_next64(g);
    int idx = (int)(r & 0xff);
    r >>= 8;
    int sign = (int)(r & 1);
    uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
    double x = (double)rabs * z.wi[idx];
    if (sign) x = -x;
    if (rabs < z.ki[idx]) return x;  // 99.3 % of draws
    if (idx == 0) {
      for (;;) {
        double xx = -ZINV * nsg_log1p(-pcg_double(g));
        double yy = -nsg_log1p(-pcg_double(g));
        if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(ZR + xx) : ZR + xx;
      }
    } else {
      double f1 = z.fi[idx - 1], f0 = z.fi[idx];
      if ((f1 - f0) * pcg_double(g) + f0 < nsg_exp(-0.5 * x * x)) return x;
    }
  }
}

__device__ __forceinline__ double pcg_normal(Pcg& g, const ZigLds& z, double loc, double scale) {
  return loc + scale * pcg_std_normal(g, z);
}

// numpy random_standard_exponential (256-layer ziggurat), tables in LDS
__device__ inline double pcg_std_exponential(Pcg& g, const ZigLds& z) {
  const double ZER = 7.69711747013104972;
  for (;;) {
    uint64_t ri = pcg_next64(g);
    ri >>= 3;
    const int idx = (int)(ri & 0xFF);
    ri >>= 8;
    const double x = (double)ri * z.we[idx];
    if (ri < z.ke[idx]) return x;  // 98.9 % of draws
    if (idx == 0) return ZER - nsg_log1p(-pcg_double(g));
    const double f1 = z.fe[idx - 1], f0 = z.fe[idx];
    if ((f1 - f0) * pcg_double(g) + f0 < nsg_exp(-x)) return x;
  }
}

// numpy random_geometric: search for p >= 1/3, inversion (ceil(-Exp / log1p(-p))) otherwise
__device__ inline int64_t pcg_geometric(Pcg& g, const ZigLds& z, double p) {
  if (p >= 0.333333333333333333333333) {
    int64_t X = 1;
    double sum = p, prod = p;
    const double q = 1.0 - p, U = pcg_double(g);
    while (U > sum && X < 0x7fffffff) {  // bounded: the kernel must terminate for any p
      prod *= q;
      sum += prod;
      X++;
    }
    return X;
  }
  const double zz = ceil(-pcg_std_exponential(g, z) / nsg_log1p(-p));
  if (zz >= 9.223372036854776e+18) return 0x7fffffffffffffffLL;
  return (int64_t)zz;
}

}  // namespace nsg
