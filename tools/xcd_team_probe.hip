// xcd_team_probe.hip -- go / no-go microbenchmark for an XCD-team persistent step kernel.
//
// Questions, each answered with a number on MI355X:
//   A. census: do the workgroups of one 768-block launch spread evenly over the eight XCCs (HW_REG_XCC_ID)?
//   B. what does a returning device-scope atomicAdd (the ticket dispenser) cost, idle and with every workgroup pulling?
//   C. an XCD-LOCAL barrier of a team's workgroups WITHOUT atomics: every workgroup plain-stores its epoch into its own
//      word (the line stays in the XCD's L2), one wave polls the team's words with L1-bypassing (sc1) loads -- against the
//      memory-side form (agent atomic counter).
//   D. a same-XCD producer -> consumer hand-off of 16 KB slabs (37 of them = one 614 KB activation of the C2 UNet's first
//      level): plain payload stores -> vmcnt(0) -> plain flag store | sc1 poll -> sc1 loads, every word verified with the
//      consumer's L1 warm from the previous epoch; against the write-through / agent-scope form the chained pair of round 4
//      used, and against a cross-XCD pairing.  Reported: flag latency (producer drained -> consumer saw it, 100 MHz wall
//      clock) and the consumer's read bandwidth.
// Every spin is bounded (20 ms); a give-up sets a flag and the run is reported as failed, never a hang.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/xcd_team_probe tools/xcd_team_probe.hip && tools/bin/xcd_team_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CHECK(x)                                                                            \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
      exit(1);                                                                              \
    }                                                                                       \
  } while (0)

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 0xf; }
__device__ __forceinline__ unsigned load_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store16_plain(char* p, const u32x4_t& v) { *reinterpret_cast<u32x4_t*>(p) = v; }
__device__ __forceinline__ void store16_wt(char* p, const u32x4_t& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void load16_sc1(u32x4_t& d, const char* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(d) : "v"(p) : "memory"); }
// a PLAIN dword store the compiler can neither drop nor promote (a volatile store compiles to `sc0 sc1`: system scope)
__device__ __forceinline__ void store_plain_u32(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store16_plain_asm(char* p, const u32x4_t& v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct Ctl {
  unsigned team_count[8 * 16];   // registration counters, one line each
  unsigned total;                // workgroups registered
  unsigned pad0[15];
  unsigned fail;                 // a bounded spin gave up
  unsigned pad1[15];
  unsigned head[8 * 16];         // ticket heads per XCC (test B)
  unsigned ghead[16];            // one global head
  unsigned bar_cnt[8 * 16];      // atomic barrier counters per XCC (test C, memory-side form)
};

static constexpr unsigned long long kSpinTicks = 2000000ull;   // 20 ms at 100 MHz

// wait until *p (sc1 load) >= want; false on time-out or when another workgroup already failed
__device__ __forceinline__ bool poll_ge(const unsigned* p, unsigned want, unsigned* fail) {
  const unsigned long long t0 = wall_clock64();
  for (int it = 0;; ++it) {
    if (load_sc1(p) >= want) return true;
    __builtin_amdgcn_s_sleep(2);
    if ((it & 63) == 63) {
      if (load_sc1(fail)) return false;
      if (wall_clock64() - t0 > kSpinTicks) {
        __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

// every workgroup: XCC id, rank inside its team; then wait for the whole launch to have registered (team sizes known)
__device__ __forceinline__ bool team_join(Ctl* ctl, unsigned& xcc, unsigned& rank, unsigned& tsize, unsigned* sh) {
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id() & 7;
    sh[0] = x;
    sh[1] = __hip_atomic_fetch_add(&ctl->team_count[x * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&ctl->total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[3] = poll_ge(&ctl->total, gridDim.x, &ctl->fail) ? 1u : 0u;
    sh[2] = load_sc1(&ctl->team_count[x * 16]);
  }
  __syncthreads();
  xcc = sh[0]; rank = sh[1]; tsize = sh[2];
  const bool ok = sh[3] != 0;
  __syncthreads();
  return ok;
}

// ---- A. census ---------------------------------------------------------------------------------------------------
__global__ void census_kernel(unsigned* out) {
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = xcc_id();
    out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11));   // HW_ID
  }
}

// ---- B. ticket dispenser -----------------------------------------------------------------------------------------
// mode 0: per-XCC heads, mode 1: one global head.  out[block] = mean wall ticks (x100) per returning atomicAdd
__global__ void ticket_kernel(Ctl* ctl, int mode, int iters, unsigned* out) {
  __shared__ unsigned sh[8];
  unsigned xcc, rank, tsize;
  if (!team_join(ctl, xcc, rank, tsize, sh)) return;
  if (threadIdx.x == 0) {
    unsigned* head = mode == 0 ? &ctl->head[xcc * 16] : &ctl->ghead[0];
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      acc += __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the next pull depends on the previous value, as a ticket loop's does
      if (acc == 0xffffffffu) head += 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x] = (unsigned)((t1 - t0) / (unsigned long long)iters);
  }
}

// ---- C. XCD-local barrier ----------------------------------------------------------------------------------------
// mode 0: flag words (plain store into the own word, sc1 polls of the team's words); mode 1: agent atomic counter + sc1 poll
// flags: [8 teams][256 words]
__global__ void barrier_kernel(Ctl* ctl, unsigned* flags, int mode, int iters, unsigned* out) {
  __shared__ unsigned sh[8];
  unsigned xcc, rank, tsize;
  if (!team_join(ctl, xcc, rank, tsize, sh)) return;
  unsigned* mine = flags + xcc * 256 + rank;
  const unsigned* team = flags + xcc * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  bool ok = true;
  for (int e = 1; e <= iters && ok; ++e) {
    __syncthreads();   // (the phase's work would be here)
    if (mode == 0) {
      if (threadIdx.x == 0) {
        store_plain_u32(mine, (unsigned)e);   // plain store: stays in this XCD's L2
        wait_vm0();
      }
      if (wave == 0) {
        const unsigned long long ts = wall_clock64();
        for (int it = 0;; ++it) {
          bool all = true;
          for (unsigned w0 = 0; w0 < tsize; w0 += 64) {
            const unsigned w = w0 + lane;
            const unsigned v = w < tsize ? load_sc1(team + w) : (unsigned)e;
            all = all && v >= (unsigned)e;
          }
          if (__all(all)) break;
          __builtin_amdgcn_s_sleep(1);
          if ((it & 255) == 255 && (load_sc1(&ctl->fail) || wall_clock64() - ts > kSpinTicks)) {
            if (lane == 0) __hip_atomic_store(&ctl->fail, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = false;
            break;
          }
        }
        if (lane == 0) sh[4] = ok ? 1u : 0u;
      }
    } else {
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl->bar_cnt[xcc * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh[4] = poll_ge(&ctl->bar_cnt[xcc * 16], tsize * (unsigned)e, &ctl->fail) ? 1u : 0u;
      }
    }
    __syncthreads();
    ok = sh[4] != 0;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = ok ? (unsigned)((t1 - t0) / (unsigned long long)iters) : 0xffffffffu;
}

// ---- D. producer -> consumer hand-off ----------------------------------------------------------------------------
// mode 0: XCD-local form (plain stores, plain flag, sc1 poll, sc1 loads)
// mode 1: write-through form (sc1 stores, agent atomic flag, sc1 poll, sc1 loads)
// mode 2: hazard: mode 0 with PLAIN consumer loads (expected stale: the consumer's L1 holds the previous epoch)
// mode 3: mode 0 across XCDs (producer on team x, consumer on team x ^ 1): expected stale
// mode 4: mode 1 across XCDs: expected correct (placement-independent)
// slabs: [pairs][kSlab bytes]; flags: [pairs][32 words] (flag @0, ack @16: separate 64-byte lines)
static constexpr int kSlab = 16384;
struct HandStats { unsigned long long lat_sum, read_sum, iter_sum; unsigned bad, n; };

__global__ void handoff_kernel(Ctl* ctl, char* slabs, unsigned* flags, unsigned long long* stamps, int mode, int iters, int active_pairs_per_team,
                               HandStats* out) {
  __shared__ unsigned sh[8];
  unsigned xcc, rank, tsize;
  if (!team_join(ctl, xcc, rank, tsize, sh)) return;
  const bool cross = mode == 3 || mode == 4;
  const unsigned pair_local = rank >> 1;
  const bool producer = (rank & 1) == 0;
  // pair id: the producer's team owns it; a cross pairing takes the consumer from the neighbouring team
  const unsigned owner = producer ? xcc : (cross ? (xcc ^ 1u) : xcc);
  const unsigned pair = owner * 128 + pair_local;
  const bool active = pair_local < (unsigned)active_pairs_per_team && (rank | 1u) < (tsize & ~1u);
  if (!active) return;
  char* slab = slabs + (size_t)pair * kSlab;
  unsigned* flag = flags + (size_t)pair * 32;
  unsigned* ack = flag + 16;
  unsigned long long* st = stamps + (size_t)pair * 2;
  const int tid = threadIdx.x;
  const bool wt = mode == 1 || mode == 4;
  unsigned bad = 0;
  unsigned long long lat_sum = 0, read_sum = 0;
  const unsigned long long t_begin = wall_clock64();
  bool ok = true;
  for (int e = 1; e <= iters && ok; ++e) {
    if (producer) {
      // 16 KB = 256 threads x 4 x 16 B; the value encodes (pair, epoch, position)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned idx = (unsigned)(q * 256 + tid);
        const unsigned base = (pair << 20) ^ ((unsigned)e << 12) ^ idx;
        const u32x4_t v = {base, base + 0x1000000u, base + 0x2000000u, base + 0x3000000u};
        if (wt) store16_wt(slab + (size_t)idx * 16, v);
        else store16_plain_asm(slab + (size_t)idx * 16, v);
      }
      wait_vm0();
      __syncthreads();
      if (tid == 0) {
        if (wt) __hip_atomic_store(st, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else st[0] = wall_clock64();                  // (plain store; read by the consumer only after the flag)
        wait_vm0();
        if (wt) __hip_atomic_store(flag, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else store_plain_u32(flag, (unsigned)e);
        wait_vm0();
        sh[4] = poll_ge(ack, (unsigned)e, &ctl->fail) ? 1u : 0u;
      }
      __syncthreads();
      ok = sh[4] != 0;
    } else {
      if (tid == 0) {
        const bool got = poll_ge(flag, (unsigned)e, &ctl->fail);
        const unsigned long long now = wall_clock64();
        sh[4] = got ? 1u : 0u;
        const unsigned long long t_prod = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got && now > t_prod) lat_sum += now - t_prod;
      }
      __syncthreads();
      ok = sh[4] != 0;
      if (!ok) break;
      const unsigned long long r0 = wall_clock64();
      u32x4_t v[4];
      if (mode == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const u32x4_t*>(slab + (size_t)(q * 256 + tid) * 16);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) load16_sc1(v[q], slab + (size_t)(q * 256 + tid) * 16);
        wait_vm0();
      }
      unsigned b = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned idx = (unsigned)(q * 256 + tid);
        const unsigned base = (pair << 20) ^ ((unsigned)e << 12) ^ idx;
        b += (v[q][0] != base) + (v[q][1] != base + 0x1000000u) + (v[q][2] != base + 0x2000000u) + (v[q][3] != base + 0x3000000u);
      }
      bad += b;
      __syncthreads();
      if (tid == 0) {
        read_sum += wall_clock64() - r0;
        if (wt) __hip_atomic_store(ack, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else store_plain_u32(ack, (unsigned)e);
        wait_vm0();
      }
    }
  }
  const unsigned long long t_end = wall_clock64();
  // reduce `bad` over the workgroup
  __shared__ unsigned sbad;
  if (tid == 0) sbad = 0;
  __syncthreads();
  atomicAdd(&sbad, bad);
  __syncthreads();
  if (tid == 0 && !producer) {
    HandStats* o = out + pair;
    o->lat_sum = lat_sum; o->read_sum = read_sum; o->iter_sum = t_end - t_begin; o->bad = sbad; o->n = ok ? (unsigned)iters : 0u;
  }
}

static double clock_ghz() {
  int khz = 0;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  return khz > 0 ? khz * 1e-6 : 2.4;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 768;
  const int iters = argc > 2 ? atoi(argv[2]) : 200;
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("# xcd_team_probe on %s (%s), %d CUs, grid %d x 256 threads, %d iterations\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, grid, iters);
  const double ghz = clock_ghz();
  Ctl* ctl;
  CHECK(hipMalloc(&ctl, sizeof(Ctl)));
  unsigned* out;
  CHECK(hipMalloc(&out, (size_t)grid * 2 * 4));
  std::vector<unsigned> h(grid * 2);

  // A. census
  census_kernel<<<grid, 256>>>(out);
  CHECK(hipMemcpy(h.data(), out, (size_t)grid * 2 * 4, hipMemcpyDeviceToHost));
  {
    int per[16] = {0};
    int map_ok = 1, first[8];
    for (int i = 0; i < 8; ++i) first[i] = -1;
    for (int b = 0; b < grid; ++b) {
      const int x = h[b * 2] & 15;
      per[x]++;
      if (first[b & 7] < 0) first[b & 7] = x;
      else if (first[b & 7] != x) map_ok = 0;
    }
    printf("A. census: workgroups per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
    printf("; blockIdx %% 8 -> XCC is %s (class 0..7 ->", map_ok ? "a fixed map" : "NOT a fixed map");
    for (int i = 0; i < 8; ++i) printf(" %d", first[i]);
    printf(")\n");
  }

  // B. tickets
  for (int mode = 0; mode < 2; ++mode)
    for (int g : {8, 64, grid}) {
      CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
      ticket_kernel<<<g, 256>>>(ctl, mode, iters, out);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h.data(), out, (size_t)g * 4, hipMemcpyDeviceToHost));
      double s = 0;
      unsigned mx = 0;
      for (int b = 0; b < g; ++b) { s += h[b]; mx = std::max(mx, h[b]); }
      printf("B. returning agent atomicAdd, %s, %d pullers: mean %.0f cycles = %.2f us per pull (max %.2f us)\n", mode == 0 ? "per-XCC heads" : "one head", g,
             s / g, s / g / ghz * 1e-3, mx / ghz * 1e-3);
    }

  // C. barrier
  unsigned* flags;
  CHECK(hipMalloc(&flags, 8 * 256 * 4 + 1024 * 32 * 4));
  for (int mode = 0; mode < 2; ++mode)
    for (int g : {256, 512, grid}) {
      CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
      CHECK(hipMemset(flags, 0, 8 * 256 * 4));
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      barrier_kernel<<<g, 256>>>(ctl, flags, mode, iters, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(h.data(), out, (size_t)g * 4, hipMemcpyDeviceToHost));
      double s = 0;
      int failed = 0;
      for (int b = 0; b < g; ++b) { if (h[b] == 0xffffffffu) failed++; else s += h[b]; }
      unsigned fl = 0;
      CHECK(hipMemcpy(&fl, &ctl->fail, 4, hipMemcpyDeviceToHost));
      printf("C. XCD-local barrier, %s, %d workgroups (%d per XCC): %.2f us per barrier (device cycles), %.2f us (host-paired wall / iterations)%s\n",
             mode == 0 ? "flag words in L2 (no atomics)" : "agent atomic counter", g, g / 8, s / std::max(1, g - failed) / ghz * 1e-3, ms * 1e3 / iters,
             (failed || fl) ? "  [FAILED: a spin gave up]" : "");
    }

  // D. hand-off
  char* slabs;
  const int npairs = 8 * 128;
  CHECK(hipMalloc(&slabs, (size_t)npairs * kSlab));
  unsigned* hflags;
  CHECK(hipMalloc(&hflags, (size_t)npairs * 32 * 4));
  unsigned long long* stamps;
  CHECK(hipMalloc(&stamps, (size_t)npairs * 2 * 8));
  HandStats* hs;
  CHECK(hipMalloc(&hs, (size_t)npairs * sizeof(HandStats)));
  std::vector<HandStats> hh(npairs);
  const char* names[5] = {"XCD-local: plain stores + plain flag | sc1 poll + sc1 loads", "write-through: sc1 stores + agent flag | sc1 poll + sc1 loads",
                          "HAZARD: XCD-local stores, PLAIN consumer loads (L1 warm)", "XCD-local form ACROSS XCDs (expected stale)", "write-through form across XCDs"};
  for (int mode = 0; mode < 5; ++mode)
    for (int active : {1, 37, 48}) {
      CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
      CHECK(hipMemset(hflags, 0, (size_t)npairs * 32 * 4));
      CHECK(hipMemset(stamps, 0, (size_t)npairs * 2 * 8));
      CHECK(hipMemset(hs, 0, (size_t)npairs * sizeof(HandStats)));
      CHECK(hipMemset(slabs, 0, (size_t)npairs * kSlab));
      handoff_kernel<<<grid, 256>>>(ctl, slabs, hflags, stamps, mode, iters, active, hs);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(hh.data(), hs, (size_t)npairs * sizeof(HandStats), hipMemcpyDeviceToHost));
      unsigned fl = 0;
      CHECK(hipMemcpy(&fl, &ctl->fail, 4, hipMemcpyDeviceToHost));
      double lat = 0, rd = 0, it = 0;
      unsigned long long bad = 0, n = 0;
      int pairs = 0;
      for (int p = 0; p < npairs; ++p)
        if (hh[p].n) { lat += (double)hh[p].lat_sum / hh[p].n; rd += (double)hh[p].read_sum / hh[p].n; it += (double)hh[p].iter_sum / hh[p].n; bad += hh[p].bad; n += hh[p].n; pairs++; }
      if (!pairs) { printf("D. %s, %d pairs per XCC: no pair finished%s\n", names[mode], active, fl ? " [a spin gave up]" : ""); continue; }
      printf("D. %s, %d pairs per XCC (%d pairs): flag latency %.2f us, consumer read of 16 KB %.2f us (%.1f GB/s per workgroup), round trip %.2f us; stale words %llu of %llu%s\n",
             names[mode], active, pairs, lat / pairs * 0.01, rd / pairs * 0.01, kSlab / (rd / pairs * 0.01) * 1e-3, it / pairs * 0.01, bad,
             n * (unsigned long long)(kSlab / 4), fl ? "  [a spin gave up]" : "");
    }
  return 0;
}
