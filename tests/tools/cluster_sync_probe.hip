// cluster_sync_probe.hip -- what one synchronisation of a cluster of G workgroups costs on gfx950 (flag all-gather through device memory, with a payload),
// for clusters placed on one XCD (workgroup ids 8 apart) or spread over the eight (consecutive ids).  Build: hipcc --offload-arch=gfx950 -O3 cluster_sync_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Slot { unsigned long long flag; double part[7]; double halo[2][704]; unsigned long long gran[2][1440]; };   // one per workgroup; gran: {32 data bits, 32-bit tag} granules

template <int MODE>
__global__ __launch_bounds__(512) void probe(Slot* slots, int G, int same_xcd, int rounds, int payload, double* out, long long* ticks, int* bad) {
  // cluster membership: same_xcd -> members {base + 8 k}; else consecutive
  const int b = blockIdx.x;
  int cl, g;
  if (same_xcd) { const int x = b % 8, r = b / 8; cl = (r / G) * 8 + x; g = r % G; }
  else { cl = b / G; g = b % G; }
  auto member = [&](int k) { return same_xcd ? ((cl / 8) * G + k) * 8 + (cl % 8) : cl * G + k; };
  Slot* mine = slots + b;
  __shared__ double red[32];
  double acc = 0.0;
  const long long t0 = wall_clock64();
  for (int e = 1; e <= rounds; ++e) {
    // local partial (a fake reduction) + payload
    double v = (double)(threadIdx.x + e + g);
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    double s = 0, h = 0;
    const Slot* nb = slots + member((g + 1) % G);
    const int gn = (g + 1) % G;
    if (MODE == 3) {          // no flag, no fence: every 8-byte granule carries its own tag (the round number); a double travels as two granules
      const unsigned tag = (unsigned)e;
      auto put = [&](int i, double x) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        __hip_atomic_store(&mine->gran[e & 1][2 * i], ((b >> 32) << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->gran[e & 1][2 * i + 1], (b << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      auto get = [&](const Slot* o, int i) {
        unsigned long long hi, lo;
        do { hi = __hip_atomic_load(&o->gran[e & 1][2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)hi != tag);
        do { lo = __hip_atomic_load(&o->gran[e & 1][2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)lo != tag);
        return __longlong_as_double((long long)(((hi >> 32) << 32) | (lo >> 32)));
      };
      for (int i = threadIdx.x; i < payload; i += blockDim.x) put(16 + i, (double)(e * 1000 + i + g));
      __syncthreads();                            // red[] complete
      if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < 8; ++i) t += red[i]; put(0, t); }
      if (threadIdx.x < G) red[8 + threadIdx.x] = get(slots + member(threadIdx.x), 0);
      for (int i = threadIdx.x; i < payload; i += blockDim.x) { const double t = get(nb, 16 + i); if (t != (double)(e * 1000 + i + gn)) atomicAdd(bad, 1); h += t; }
      __syncthreads();
      for (int k = 0; k < G; ++k) s += red[8 + k];
      __syncthreads();
    } else if (MODE == 2) {          // release / acquire at agent scope
      for (int i = threadIdx.x; i < payload; i += blockDim.x) mine->halo[e & 1][i] = (double)(e * 1000 + i + g);
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0; for (int i = 0; i < 8; ++i) t += red[i];
        mine->part[e & 1] = t;
        __hip_atomic_store(&mine->flag, (unsigned long long)e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (threadIdx.x < G) {
        const Slot* o = slots + member(threadIdx.x);
        while (__hip_atomic_load(&o->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)e) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int k = 0; k < G; ++k) s += slots[member(k)].part[e & 1];
      for (int i = threadIdx.x; i < payload; i += blockDim.x) { const double t = nb->halo[e & 1][i]; if (t != (double)(e * 1000 + i + gn)) atomicAdd(bad, 1); h += t; }
    } else if (MODE == 0) {
      for (int i = threadIdx.x; i < payload; i += blockDim.x) mine->halo[e & 1][i] = (double)(e * 1000 + i + g);
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0; for (int i = 0; i < 8; ++i) t += red[i];
        mine->part[e & 1] = t;
        __atomic_store_n(&mine->flag, (unsigned long long)e, __ATOMIC_RELEASE);       // agent scope by default in HIP
      }
      if (threadIdx.x < G) {
        const Slot* o = slots + member(threadIdx.x);
        while (__atomic_load_n(&o->flag, __ATOMIC_ACQUIRE) < (unsigned long long)e) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      __atomic_thread_fence(__ATOMIC_ACQUIRE);      // the other wavefronts read the payload too
      for (int k = 0; k < G; ++k) s += slots[member(k)].part[e & 1];
      for (int i = threadIdx.x; i < payload; i += blockDim.x) { const double t = nb->halo[e & 1][i]; if (t != (double)(e * 1000 + i + gn)) atomicAdd(bad, 1); h += t; }
    } else {
      for (int i = threadIdx.x; i < payload; i += blockDim.x) __hip_atomic_store(&mine->halo[e & 1][i], (double)(e * 1000 + i + g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0) {
        double t = 0; for (int i = 0; i < 8; ++i) t += red[i];
        __hip_atomic_store(&mine->part[e & 1], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // (orders the compiler; emits NO s_waitcnt for device memory on this target)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the write-through stores above have been acknowledged
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(&mine->flag, (unsigned long long)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x < G) {
        const Slot* o = slots + member(threadIdx.x);
        while (__hip_atomic_load(&o->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)e) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __syncthreads();
      for (int k = 0; k < G; ++k) s += __hip_atomic_load(&slots[member(k)].part[e & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = threadIdx.x; i < payload; i += blockDim.x) { const double t = __hip_atomic_load(&nb->halo[e & 1][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (t != (double)(e * 1000 + i + gn)) atomicAdd(bad, 1); h += t; }
    }
    acc = acc * 0.5 + s * 1e-9 + h * 1e-12;
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[b] = acc; ticks[b] = t1 - t0; }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  printf("CUs %d wall clock %d kHz\n", ncu, rate);
  Slot* d; double* out; long long* ticks;
  hipMalloc(&d, sizeof(Slot) * 1024); hipMalloc(&out, 8 * 1024); hipMalloc(&ticks, 8 * 1024);
  int* bad; hipMalloc(&bad, 4);
  for (int mode = 1; mode < 4; mode += 2)
  for (int same = 0; same < 2; ++same)
    for (int G : {1, 8, 16})
      for (int payload : {0, 424}) {
        const int grid = (ncu / (8 * G)) * 8 * G > 0 ? (ncu / (8 * G)) * 8 * G : 8 * G;
        hipMemset(d, 0, sizeof(Slot) * 1024);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipMemset(bad, 0, 4);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, d, G, same, rounds, payload, out, ticks, bad);
        else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, d, G, same, rounds, payload, out, ticks, bad);
        else if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(512), 0, 0, d, G, same, rounds, payload, out, ticks, bad);
        else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, d, G, same, rounds, payload, out, ticks, bad);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        int nbad = 0; hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
        printf("mode %d same_xcd %d G %2d payload %3d grid %3d: %.3f us per sync, %d wrong payload values (%s)\n", mode, same, G, payload, grid, 1e3 * ms / rounds, nbad, hipGetErrorString(hipGetLastError()));
      }
  return 0;
}
