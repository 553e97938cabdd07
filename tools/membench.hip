// membench.hip — floor measurements for the step kernel's ACCESS PATTERN on one MI355X (diagnostic tool, not product).
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_membench tools/membench.hip && ./gpurun_membench
//
// One "env" = a 16-byte record + up to 8 planes of PS bytes (912 or 1024).  Each variant moves the same bytes the
// C3 op mix moves per env (reads ~0.7 plane, writes ~1.6 planes; table below) or a uniform pattern, with NO grid
// arithmetic — what remains is launch + wave ramp + the dependent memory chain + the store drain.
//   mode 0  one wave per env:   record load -> (dependent) plane loads -> plane stores        (the round-1 structure)
//   mode 1  E envs per wave, interleaved: all E records, then all plane loads, then all stores (fewer waves, more MLP)
//   mode 2  E envs per wave, sequential, next record prefetched                               (persistent-style loop)
//   mode 3  one wave per env, NO record dependency (planes decided from the env index alone)   (single round trip)
// Times: average of K back-to-back launches between two hipEvents on the launch stream.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint32_t U4 __attribute__((ext_vector_type(4)));

struct MB {
  int8_t* plane[8];
  uint8_t* rec;  // [n][16]; byte 0 = pattern id
  int n_envs, PS, E, pattern, salu, valu;
};

// reads / writes per op slot of the C3 mix (35 ops uniform; ELIDE_SELECTED on; bbox => object ops always lift fresh)
__constant__ uint8_t kR[35] = {1,1,1,1,1,1,1,1,1,1, 0,0,0,0,0,0,0,0,0,0, 1,1,1,1,1,1,1,1, 1,1,2,1,0,0,2};
__constant__ uint8_t kW[35] = {1,1,1,1,1,1,1,1,1,1, 0,0,0,0,0,0,0,0,0,0, 5,5,5,5,5,5,5,5, 1,1,1,1,1,1,0};

template <int POLICY>
__device__ __forceinline__ void store16(int8_t* p, const U4& v) {
  if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void burn(const MB& p, U4& acc) {
  for (int i = 0; i < p.valu; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[0]) : "v"(acc[1]));
  uint32_t s = 1;
  for (int i = 0; i < p.salu; i++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
  acc[1] += s;
}

__device__ __forceinline__ void pattern_of(const MB& p, int env, uint32_t recbyte, int& nr, int& nw) {
  if (p.pattern >= 0) {  // uniform: pattern = nr*16 + nw
    nr = p.pattern >> 4;
    nw = p.pattern & 15;
  } else {
    const int slot = (int)(recbyte % 35u);
    nr = kR[slot];
    nw = kW[slot];
  }
  (void)env;
}

template <int POLICY, int MODE, int EMAX>
__global__ __launch_bounds__(256) void mb_kernel(const MB p) {
  const int lane = threadIdx.x & 63;
  const uint32_t nb = gridDim.x, b = blockIdx.x;
  const uint32_t vb = (b & 7u) * (nb >> 3) + (b >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(vb * 4 + (threadIdx.x >> 6)));
  const int E = (MODE == 0 || MODE == 3) ? 1 : p.E;
  const int env0 = wave * E;
  if (env0 >= p.n_envs) return;
  const bool live = 16 * lane < p.PS;
  if (MODE == 0 || MODE == 3) {
    uint32_t rb = (uint32_t)env0 * 2654435761u >> 8;
    if (MODE == 0) {
      U4 r = *reinterpret_cast<const U4*>(p.rec + (size_t)env0 * 16);
      rb = __builtin_amdgcn_readfirstlane(r[0]);
    }
    int nr, nw;
    pattern_of(p, env0, rb, nr, nw);
    const size_t off = (size_t)env0 * p.PS + 16 * lane;
    U4 acc = {rb, 1, 2, 3};
    for (int i = 0; i < nr; i++)
      if (live) acc += *reinterpret_cast<const U4*>(p.plane[i] + off);
    burn(p, acc);
    for (int i = 0; i < nw; i++)
      if (live) store16<POLICY>(p.plane[7 - i] + off, acc);
    if (lane == 0 && nw) *reinterpret_cast<U4*>(p.rec + (size_t)env0 * 16) = U4{rb, acc[1] & 0, 0, 0};
    return;
  }
  if (MODE == 1) {
    uint32_t rb[EMAX];
    int nr[EMAX], nw[EMAX];
    U4 rr[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; e++)
      if (e < E) rr[e] = *reinterpret_cast<const U4*>(p.rec + (size_t)(env0 + e) * 16);
#pragma unroll
    for (int e = 0; e < EMAX; e++)
      if (e < E) {
        rb[e] = __builtin_amdgcn_readfirstlane(rr[e][0]);
        pattern_of(p, env0 + e, rb[e], nr[e], nw[e]);
      }
    U4 acc[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; e++)
      if (e < E) {
        acc[e] = U4{rb[e], 1, 2, 3};
        const size_t off = (size_t)(env0 + e) * p.PS + 16 * lane;
        for (int i = 0; i < nr[e]; i++)
          if (live) acc[e] += *reinterpret_cast<const U4*>(p.plane[i] + off);
      }
#pragma unroll
    for (int e = 0; e < EMAX; e++)
      if (e < E) {
        burn(p, acc[e]);
        const size_t off = (size_t)(env0 + e) * p.PS + 16 * lane;
        for (int i = 0; i < nw[e]; i++)
          if (live) store16<POLICY>(p.plane[7 - i] + off, acc[e]);
        if (lane == 0 && nw[e]) *reinterpret_cast<U4*>(p.rec + (size_t)(env0 + e) * 16) = U4{rb[e], 0, 0, 0};
      }
    return;
  }
  if (MODE == 2) {
    U4 next = *reinterpret_cast<const U4*>(p.rec + (size_t)env0 * 16);
    for (int e = 0; e < E; e++) {
      const int env = env0 + e;
      const uint32_t rb = __builtin_amdgcn_readfirstlane(next[0]);
      if (e + 1 < E) next = *reinterpret_cast<const U4*>(p.rec + (size_t)(env + 1) * 16);
      int nr, nw;
      pattern_of(p, env, rb, nr, nw);
      const size_t off = (size_t)env * p.PS + 16 * lane;
      U4 acc = {rb, 1, 2, 3};
      for (int i = 0; i < nr; i++)
        if (live) acc += *reinterpret_cast<const U4*>(p.plane[i] + off);
      burn(p, acc);
      for (int i = 0; i < nw; i++)
        if (live) store16<POLICY>(p.plane[7 - i] + off, acc);
      if (lane == 0 && nw) *reinterpret_cast<U4*>(p.rec + (size_t)env * 16) = U4{rb, 0, 0, 0};
    }
  }
}

__global__ void mb_empty(const MB p) { (void)p; }

template <int POLICY, int MODE>
static float run(const MB& p, int K, hipStream_t st) {
  const int E = (MODE == 0 || MODE == 3) ? 1 : p.E;
  unsigned waves = (unsigned)((p.n_envs + E - 1) / E);
  unsigned nb = ((waves + 3) / 4 + 7u) & ~7u;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL((mb_kernel<POLICY, MODE, 8>), dim3(nb), dim3(256), 0, st, p);
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int i = 0; i < K; i++) hipLaunchKernelGGL((mb_kernel<POLICY, MODE, 8>), dim3(nb), dim3(256), 0, st, p);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / K;
}

static float run_empty(const MB& p, unsigned nb, int K, hipStream_t st) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(mb_empty, dim3(nb), dim3(256), 0, st, p);
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int i = 0; i < K; i++) hipLaunchKernelGGL(mb_empty, dim3(nb), dim3(256), 0, st, p);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / K;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 300;
  hipStream_t st;
  hipStreamCreate(&st);
  printf("# membench: us per launch (avg of %d back-to-back launches, HIP events)\n", K);
  for (unsigned nb : {256u, 512u, 1024u, 2048u, 4096u}) {
    MB p{};
    printf("empty kernel, %u WGs x 256 threads: %.2f us\n", nb, run_empty(p, nb, K, st));
  }
  const int NMAX = 131072;
  for (int PS : {912, 1024}) {
    MB p{};
    for (int i = 0; i < 8; i++) {
      hipMalloc((void**)&p.plane[i], (size_t)NMAX * PS + 4096);
      hipMemset(p.plane[i], 1, (size_t)NMAX * PS + 4096);
    }
    std::vector<uint8_t> h((size_t)NMAX * 16);
    srand(7);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(rand() % 35);
    hipMalloc((void**)&p.rec, h.size());
    hipMemcpy(p.rec, h.data(), h.size(), hipMemcpyHostToDevice);
    p.PS = PS;
    for (int n : {8192, 32768, 131072}) {
      p.n_envs = n;
      const int Kn = n == 8192 ? K : K / 4;
      struct Pat { const char* name; int pattern; double planes; } pats[] = {
          {"C3mix(R.71,W1.6)", -1, 25.0 / 35 + 56.0 / 35}, {"W1", 0x01, 1}, {"W2", 0x02, 2}, {"W5", 0x05, 5},
          {"R1", 0x10, 1}, {"R1W1", 0x11, 2}, {"R1W5", 0x15, 6}, {"R3W2", 0x32, 5}};
      for (auto& pt : pats) {
        p.pattern = pt.pattern;
        p.E = 1;
        p.salu = p.valu = 0;
        const double mb = pt.planes * PS * n / 1e6;
        float t_plain = run<0, 0>(p, Kn, st), t_sc1 = run<1, 0>(p, Kn, st), t_nt = run<2, 0>(p, Kn, st);
        float t_nodep = run<1, 3>(p, Kn, st);
        printf("PS=%4d N=%6d %-18s %6.1f MB | 1env/wave plain %6.2f sc1 %6.2f nt %6.2f | no-rec-dep sc1 %6.2f", PS, n, pt.name, mb,
               t_plain, t_sc1, t_nt, t_nodep);
        for (int E : {2, 4, 8}) {
          p.E = E;
          printf(" | E=%d inter %6.2f seq %6.2f", E, run<1, 1>(p, Kn, st), run<1, 2>(p, Kn, st));
        }
        printf("  => best sc1 %.0f GB/s\n", mb * 1e3 / t_sc1);
        fflush(stdout);
      }
      // instruction-issue sensitivity on the C3 mix at this N (1 env per wave, sc1)
      p.pattern = -1;
      p.E = 1;
      for (int s : {0, 100, 200, 400}) {
        for (int v : {0, 100, 200, 400}) {
          p.salu = s;
          p.valu = v;
          printf("PS=%4d N=%6d C3mix +%3d SALU +%3d VALU per wave: %6.2f us", PS, n, s, v, run<1, 0>(p, Kn, st));
          p.E = 2;
          printf("  (E=2 interleaved: %6.2f)\n", run<1, 1>(p, Kn, st));
          p.E = 1;
        }
      }
    }
    for (int i = 0; i < 8; i++) hipFree(p.plane[i]);
    hipFree(p.rec);
  }
  return 0;
}
