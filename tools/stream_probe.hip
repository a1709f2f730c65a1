// stream_probe.hip — memory skeleton of the C1 CartPole step in the HBM-streaming regime (2^22-2^24 envs per GPU).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip ; run on the GPU box: ./stream_probe [log2N ...]
//
// Every variant moves the algorithmic bytes of one C1 env-step (48 B read + 71-72 B written) with a
// trivial amount of arithmetic, so the numbers bound what a layout can give the real kernel.  Knobs (template flags):
//   STATE  0 = library layout of round 1: phys chunk-blocked, theta / t / status as flat [N] rows
//          1 = every persistent row of a 256-env chunk in ONE contiguous block (phys | theta | t | status)
//   OUT    0 = outputs as flat [N] rows (obs [N][4] f32, reward, delta, term, trunc, flag)
//          1 = outputs of a chunk in one contiguous block
//   PACK   0 = status byte row + three flag byte rows
//          1 = needs-reset bit in t's sign bit, term / trunc / flag as ONE dword row (4-B stores, full 256-B wave stores)
//   CPW    consecutive chunks a workgroup walks before it strides by the grid (longer contiguous runs per stream)
//   NT     non-temporal stores of the write-once outputs
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e = (x);                                                          \
    if (e != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                     \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Bufs {
  unsigned char* state;  // STATE 0: phys [chunk][4][256] f64 | theta [N] f64 | t [N] i32 | status [N] u8 (sub-pointers below)
  double* phys;
  double* theta;
  int32_t* t;
  uint8_t* status;
  const int32_t* action;
  unsigned char* out;  // OUT 1: [chunk][out_stride]
  f32x4* obs;
  float* reward;
  float* delta;
  uint8_t* term;
  uint8_t* trunc;
  uint8_t* flag;
  uint32_t* flags32;
};

template <bool NT, typename T> __device__ __forceinline__ void st_out(T* p, T v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

__device__ __forceinline__ void compute(double s[4], double& th, int& t, unsigned& st, int a, f32x4& o, float& rew,
                                        unsigned& term, unsigned& trunc, unsigned& flag, float& delta) {
  th += 0.1;
  const double f = a ? 10.0 : -10.0;
  s[1] += 0.02 * (f + th * s[2]);
  s[0] += 0.02 * s[1];
  s[3] += 0.02 * (s[2] * 9.8 - f * 0.05);
  s[2] += 0.02 * s[3];
  t += 1;
  rew = 1.f;
  term = (s[0] > 2.4 || s[0] < -2.4 || s[2] > 0.21 || s[2] < -0.21) ? 1u : 0u;
  trunc = t >= 500 ? 1u : 0u;
  if (st) { s[0] = 0.01; s[1] = -0.02; s[2] = 0.03; s[3] = 0.01; t = 0; th = 0.1; term = trunc = 0; }
  st = (term | trunc) ? 1u : 0u;
  o = f32x4{(float)s[0], (float)s[1], (float)s[2], (float)s[3]};
  flag = 1u;
  delta = 0.1f;
}

constexpr int state_stride(int STATE, int PACK) { return 8192 + 2048 + 1024 + (PACK ? 0 : 256); }
constexpr int out_stride(int PACK) { return 4096 + 1024 + 1024 + (PACK ? 1024 : 768); }

template <int STATE, int OUT, int PACK, int CPW, bool NT>
__global__ __launch_bounds__(256) void skel(Bufs r, int64_t N) {
  const int64_t chunks = (N + 255) / 256;
  const int l = threadIdx.x;
  for (int64_t c0 = (int64_t)blockIdx.x * CPW; c0 < chunks; c0 += (int64_t)gridDim.x * CPW) {
#pragma unroll 1
    for (int cc = 0; cc < CPW; cc++) {
      const int64_t c = c0 + cc;
      if (c >= chunks) break;
      const int64_t i = c * 256 + l;
      if (i >= N) continue;
      double* bp;
      double* bth;
      int32_t* bt;
      uint8_t* bst;
      if (STATE) {
        unsigned char* blk = r.state + c * state_stride(STATE, PACK);
        bp = (double*)blk; bth = (double*)(blk + 8192) + l; bt = (int32_t*)(blk + 10240) + l; bst = blk + 11264 + l;
      } else {
        bp = r.phys + c * 1024; bth = r.theta + i; bt = r.t + i; bst = r.status + i;
      }
      unsigned st;
      int t = *bt;
      if (PACK) { st = (unsigned)t >> 31; t &= 0x7fffffff; }
      else st = *bst;
      double s[4];
#pragma unroll
      for (int k = 0; k < 4; k++) s[k] = bp[k * 256 + l];
      const int a = r.action[i];
      double th = *bth;
      f32x4 o;
      float rew, delta;
      unsigned term, trunc, flag;
      compute(s, th, t, st, a, o, rew, term, trunc, flag, delta);
      *bth = th;
#pragma unroll
      for (int k = 0; k < 4; k++) bp[k * 256 + l] = s[k];
      if (PACK) *bt = t | (int)(st << 31);
      else { *bt = t; *bst = (uint8_t)st; }
      if (OUT) {
        unsigned char* ob = r.out + c * out_stride(PACK);
        st_out<NT>((f32x4*)ob + l, o);
        st_out<NT>((float*)(ob + 4096) + l, rew);
        st_out<NT>((float*)(ob + 5120) + l, delta);
        if (PACK) st_out<NT>((uint32_t*)(ob + 6144) + l, term | (trunc << 8) | (flag << 16));
        else {
          st_out<NT>(ob + 6144 + l, (unsigned char)term); st_out<NT>(ob + 6400 + l, (unsigned char)trunc);
          st_out<NT>(ob + 6656 + l, (unsigned char)flag);
        }
      } else {
        st_out<NT>(r.obs + i, o);
        st_out<NT>(r.reward + i, rew);
        st_out<NT>(r.delta + i, delta);
        if (PACK) st_out<NT>(r.flags32 + i, term | (trunc << 8) | (flag << 16));
        else { st_out<NT>(r.term + i, (uint8_t)term); st_out<NT>(r.trunc + i, (uint8_t)trunc); st_out<NT>(r.flag + i, (uint8_t)flag); }
      }
    }
  }
}

// pure copy of the same byte count, 16-B accesses (ceiling of the memory system for this footprint and grid)
__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    f32x4 v = src[i];
    v.x += 1.f;
    dst[i] = v;
  }
}
// read-modify-write in place (the state rows' pattern: every byte read is written back), 16-B accesses
__global__ __launch_bounds__(256) void rmw16_kernel(f32x4* __restrict__ buf, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    f32x4 v = buf[i];
    v.x += 1.f;
    buf[i] = v;
  }
}

template <typename F> float time_it(F&& launch, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int k = 0; k < 5; k++) launch();
  CHECK(hipEventRecord(e0));
  for (int k = 0; k < iters; k++) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return ms * 1000.f / iters;
}

template <int STATE, int OUT, int PACK, int CPW, bool NT> void run(const Bufs& r, int64_t N, int grid, int iters, const char* tag) {
  const int64_t chunks = (N + 255) / 256;
  const int g = (int)((chunks + CPW - 1) / CPW < grid ? (chunks + CPW - 1) / CPW : grid);
  float us = time_it([&] { hipLaunchKernelGGL((skel<STATE, OUT, PACK, CPW, NT>), dim3(g), dim3(256), 0, 0, r, N); }, iters);
  printf("{\"n_log2\": %d, \"variant\": \"%s\", \"state_blocked\": %d, \"out_blocked\": %d, \"pack\": %d, \"cpw\": %d, \"nt\": %d, \"grid\": %d, "
         "\"us\": %.2f, \"algorithmic_GBs\": %.0f}\n",
         (int)__builtin_ctzll((unsigned long long)N), tag, STATE, OUT, PACK, CPW, (int)NT, g, us, 120.0 * N / us / 1e3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  std::vector<int> sizes;
  for (int k = 1; k < argc; k++) sizes.push_back(atoi(argv[k]));
  if (sizes.empty()) sizes = {20, 22, 24};
  for (int lg : sizes) {
    const int64_t N = 1LL << lg;
    const int iters = lg >= 24 ? 40 : lg >= 22 ? 100 : 300;
    const int64_t chunks = (N + 255) / 256;
    Bufs r;
    // one allocation per role so that the flat-row and the blocked variants do not alias (both are kept zero-initialised)
    const size_t state_bytes = (size_t)chunks * 11520, out_bytes = (size_t)chunks * 7168;
    unsigned char *state, *out, *rows, *orow;
    CHECK(hipMalloc(&state, state_bytes));
    CHECK(hipMalloc(&out, out_bytes));
    CHECK(hipMalloc(&rows, (size_t)N * (32 + 8 + 4 + 1) + 4096));
    CHECK(hipMalloc(&orow, (size_t)N * (16 + 4 + 4 + 4 + 3) + 4096));
    CHECK(hipMemset(state, 0, state_bytes));
    CHECK(hipMemset(rows, 0, (size_t)N * 45 + 4096));
    CHECK(hipMalloc((void**)&r.action, N * 4));
    CHECK(hipMemset((void*)r.action, 0, N * 4));
    r.state = state;
    r.out = out;
    r.phys = (double*)rows;
    r.theta = (double*)(rows + (size_t)N * 32);
    r.t = (int32_t*)(rows + (size_t)N * 40);
    r.status = rows + (size_t)N * 44;
    r.obs = (f32x4*)orow;
    r.reward = (float*)(orow + (size_t)N * 16);
    r.delta = (float*)(orow + (size_t)N * 20);
    r.flags32 = (uint32_t*)(orow + (size_t)N * 24);
    r.term = orow + (size_t)N * 28;
    r.trunc = orow + (size_t)N * 29;
    r.flag = orow + (size_t)N * 30;
    const int64_t copy_bytes = 60 * N;
    f32x4 *src, *dst;
    CHECK(hipMalloc(&src, copy_bytes));
    CHECK(hipMalloc(&dst, copy_bytes));
    CHECK(hipMemset(src, 0, copy_bytes));
    for (int rep = 0; rep < 2; rep++) {
      const int G = 4096;
      run<0, 0, 0, 1, true>(r, N, G, iters, "lib: phys blocked, flat rows, nt outputs");
      run<0, 0, 0, 1, false>(r, N, G, iters, "lib layout, plain stores");
      run<1, 0, 0, 1, true>(r, N, G, iters, "state blocked");
      run<1, 1, 0, 1, true>(r, N, G, iters, "state+out blocked");
      run<0, 0, 1, 1, true>(r, N, G, iters, "lib layout, packed flags");
      run<1, 0, 1, 1, true>(r, N, G, iters, "state blocked, packed flags");
      run<1, 1, 1, 1, true>(r, N, G, iters, "state+out blocked, packed flags");
      run<1, 1, 1, 1, false>(r, N, G, iters, "state+out blocked, packed flags, plain stores");
      run<0, 0, 0, 4, true>(r, N, G, iters, "lib layout, 4 chunks per wg");
      run<1, 0, 1, 4, true>(r, N, G, iters, "state blocked, packed, 4 chunks per wg");
      run<1, 1, 1, 4, true>(r, N, G, iters, "state+out blocked, packed, 4 chunks per wg");
      run<1, 1, 1, 2, true>(r, N, G, iters, "state+out blocked, packed, 2 chunks per wg");
      run<1, 1, 1, 1, true>(r, N, 2048, iters, "state+out blocked, packed flags, grid 2048");
      run<1, 1, 1, 1, true>(r, N, 16384, iters, "state+out blocked, packed flags, grid 16384");
      run<0, 0, 0, 1, true>(r, N, 16384, iters, "lib layout, grid 16384");
      float e = time_it([&] { hipLaunchKernelGGL(copy16_kernel, dim3(G), dim3(256), 0, 0, src, dst, copy_bytes / 16); }, iters);
      printf("{\"n_log2\": %d, \"variant\": \"copy16 60 B in + 60 B out per env\", \"us\": %.2f, \"algorithmic_GBs\": %.0f}\n", lg, e, 120.0 * N / e / 1e3);
      float f = time_it([&] { hipLaunchKernelGGL(rmw16_kernel, dim3(G), dim3(256), 0, 0, src, copy_bytes / 16); }, iters);
      printf("{\"n_log2\": %d, \"variant\": \"rmw16 in place, 60 B per env read and written back\", \"us\": %.2f, \"algorithmic_GBs\": %.0f}\n", lg, f, 120.0 * N / f / 1e3);
      fflush(stdout);
    }
    CHECK(hipFree(state)); CHECK(hipFree(out)); CHECK(hipFree(rows)); CHECK(hipFree(orow)); CHECK(hipFree((void*)r.action));
    CHECK(hipFree(src)); CHECK(hipFree(dst));
  }
  return 0;
}
