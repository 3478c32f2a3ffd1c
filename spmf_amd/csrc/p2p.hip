// p2p.hip -- the step's one collective as a hand-written kernel over peer pointers (gfx950).
//
// SURVEY 5 (last row) / 8e: row shards live on different GPUs of one node; between the column pass and the
// finish every rank needs the SUM of the packed fp32 accumulators (5.2 MB on C3).  xGMI is a point-to-point
// mesh (7 links per GPU), so a ring serialises 2(N-1) hops on single links; this is the direct two-shot
// form instead -- reduce-scatter + all-gather with every pair of ranks talking over its own link:
//
//   A  push      rank r sends slice q of its buffer to rank q's inbox slot r            (q != r, N-1 links at once)
//   B  reduce    rank q adds the N contributions to slice q IN RANK ORDER 0..N-1 (its own from its buffer),
//                keeps the result in its buffer and pushes it to every peer's gather inbox
//   C  gather    rank r copies the N-1 reduced slices out of its gather inbox into its buffer
//
// One kernel launch per rank on the step's stream, no host synchronisation, no other library.  Every slice
// is reduced by exactly ONE rank in a fixed order and then copied, so all ranks end with the SAME bits
// (a ring all-reduce does not guarantee that; the replicas of a row-sharded job cannot drift through it).
//
// Memory.  Every byte a peer writes lives in a region this library allocates fine-grained
// (hipExtMallocWithFlags: coherent across agents inside a kernel) and exports with hipIpcGetMemHandle; the
// caller's buffer (coarse-grained torch memory) is only touched by the local kernel.  Region of a rank:
//   rs[2][N][slice_cap]   reduce-scatter inbox, slot j = rank j's contribution to MY slice, two parities
//   ag[2][N][slice_cap]   gather inbox, slot q = the reduced slice of owner q
//   flags[2 phases][2 parities][N][kMaxChunks]  uint64 sequence numbers, written by the peer
//   seq, ticket           this rank's call counter (device resident: a hipGraph replay advances it)
// Peers are "the same GPU seen through IPC" on a one-GPU box (tests) and other GPUs over xGMI on a node:
// the kernel is the same.
//
// Protocol.  Workgroup c of every rank owns chunk c of every slice: it pushes chunk c (A), waits for the
// N-1 peers' chunk-c flags, reduces chunk c of its own slice and pushes the result (B), waits for the N-1
// owners' chunk-c flags and copies (C).  A workgroup only ever waits for the SAME workgroup index on the
// other ranks, and phase A waits for nothing, so no workgroup waits for one that has not been
// dispatched on its own rank; the grid is kept far below the resident capacity anyway and every spin
// is bounded (kWaitTicks of wall clock, then the call's status word is set and the workgroup leaves).
// Release / acquire: data stores, s_waitcnt vmcnt(0) of every wave + workgroup barrier, then system-scope
// release stores of the flag (one lane per peer); the consumer polls with relaxed system-scope loads, issues
// one system-scope acquire fence after the match, then the workgroup barrier, then plain loads of
// fine-grained memory.
// Reuse without an exit barrier: inboxes and flags alternate with the call parity.  A rank can be at most
// one call ahead of a peer (call t+1 cannot finish without the peer's phase B of t+1, which follows the
// peer's whole call t in stream order), so parity (t+1) is never written while parity (t+1) of call t-1
// is still being read.  Flags carry the call number (monotonic): nothing is ever reset.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace spmf {

// A workgroup gives up waiting for a peer after this long (wall_clock64 ticks of 10 ns): ranks are separate
// processes whose launches can be skewed by whole host-side pauses, but a lost peer must not hang the GPU
constexpr long long kWaitTicks = 20LL * 100000000LL;   // 20 s

struct P2PArgs {
  float* buf;              // the caller's accumulators [n]
  int64_t n;
  int rank, world, nchunk;
  int64_t slice_cap;       // floats per inbox slot
  // this rank's region
  float* rs;               // [2][world][slice_cap]
  float* ag;               // [2][world][slice_cap]
  uint64_t* flags;         // [2][2][world][kP2PMaxChunks]
  uint64_t* seq;           // [0] calls completed, [1] ticket, [2] status (0 ok, else first failure)
  // the peers' regions as mapped here (index = rank; own entry = own region)
  float* peer_rs[kP2PMaxWorld];
  float* peer_ag[kP2PMaxWorld];
  uint64_t* peer_flags[kP2PMaxWorld];
};

__device__ __forceinline__ uint64_t* flag_at(uint64_t* flags, int phase, int par, int world_slot, int chunk) {
  return flags + (((size_t)phase * 2 + par) * kP2PMaxWorld + world_slot) * kP2PMaxChunks + chunk;
}

// all waves of the workgroup have their stores out; lane 0 then publishes `val` to every peer's flag
__device__ __forceinline__ void publish(const P2PArgs& a, int phase, int par, int chunk, uint64_t val) {
  __builtin_amdgcn_s_waitcnt(0);        // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's stores have left
  __syncthreads();
  if (threadIdx.x < (unsigned)a.world && (int)threadIdx.x != a.rank) {
    uint64_t* f = flag_at(a.peer_flags[threadIdx.x], phase, par, a.rank, chunk);
    __hip_atomic_store(f, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// wait until every peer's flag of (phase, parity, chunk) has reached `val`; false = gave up.
// The poll is a relaxed system-scope load (it bypasses the caches; an acquire load would invalidate them
// on every poll); ONE system-scope acquire fence follows the matched poll, then the workgroup barrier,
// then the other waves' plain loads.
__device__ __forceinline__ bool await(const P2PArgs& a, int phase, int par, int chunk, uint64_t val) {
  __shared__ int ok_s;
  if (threadIdx.x == 0) ok_s = 1;
  __syncthreads();
  if (threadIdx.x < 64) {
    if (threadIdx.x < (unsigned)a.world && (int)threadIdx.x != a.rank) {
      uint64_t* f = flag_at(a.flags, phase, par, threadIdx.x, chunk);
      long long t0 = 0;
      int spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < val) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023) == 0) {              // look at the clock every ~1000 polls
          const long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > kWaitTicks) {
            ok_s = 0;
            break;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  __syncthreads();
  return ok_s != 0;
}

__global__ __launch_bounds__(512) void p2p_allreduce_kernel(const P2PArgs a) {
  const int c = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const int N = a.world, r = a.rank;
  const uint64_t call = a.seq[0] + 1;                 // stream order: the previous call has stored it
  const int par = (int)(call & 1);
  // slice q = elements [q * per, min(n, (q+1) * per)), per a multiple of 4 floats; chunk c of a slice likewise
  const int64_t per = ((a.n + N - 1) / N + 3) & ~(int64_t)3;
  const int64_t cper = ((per + a.nchunk - 1) / a.nchunk + 3) & ~(int64_t)3;
  const int64_t c0 = (int64_t)c * cper;               // chunk range inside a slice
  bool alive = true;
  // ---- A: push chunk c of slice q to rank q ------------------------------------------
  for (int dq = 1; dq < N; ++dq) {
    const int q = (r + dq) % N;                       // every rank starts with a different peer
    const int64_t s0 = (int64_t)q * per;
    const int64_t len = max((int64_t)0, min(a.n, s0 + per) - s0);
    const int64_t lo = min(c0, len), hi = min(c0 + cper, len);
    float* dst = a.peer_rs[q] + ((size_t)par * kP2PMaxWorld + r) * a.slice_cap;
    const float* src = a.buf + s0;
    const int64_t lo4 = lo, hi4 = lo + ((hi - lo) & ~(int64_t)3);
    for (int64_t i = lo4 + 4 * t; i < hi4; i += 4 * (int64_t)nt)
      *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
    for (int64_t i = hi4 + t; i < hi; i += nt) dst[i] = src[i];
  }
  publish(a, 0, par, c, call);
  // ---- B: reduce chunk c of my slice in rank order, push the result -----------------------
  {
    alive = await(a, 0, par, c, call);
    const int64_t s0 = (int64_t)r * per;
    const int64_t len = max((int64_t)0, min(a.n, s0 + per) - s0);
    const int64_t lo = min(c0, len), hi = min(c0 + cper, len);
    const float* mine = a.buf + s0;
    const float* inbox = a.rs + (size_t)par * kP2PMaxWorld * a.slice_cap;
    const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
    if (alive) {
      for (int64_t i = lo + 4 * t; i < hi4; i += 4 * (int64_t)nt) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < N; ++j) {
          const float4 v = j == r ? *reinterpret_cast<const float4*>(mine + i)
                                  : *reinterpret_cast<const float4*>(inbox + (size_t)j * a.slice_cap + i);
          if (j == 0) s = v;
          else { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        }
        *reinterpret_cast<float4*>(a.buf + s0 + i) = s;
        for (int dq = 1; dq < N; ++dq) {
          const int q = (r + dq) % N;
          *reinterpret_cast<float4*>(a.peer_ag[q] + ((size_t)par * kP2PMaxWorld + r) * a.slice_cap + i) = s;
        }
      }
      for (int64_t i = hi4 + t; i < hi; i += nt) {
        float s = 0.f;
        for (int j = 0; j < N; ++j) {
          const float v = j == r ? mine[i] : inbox[(size_t)j * a.slice_cap + i];
          s = j == 0 ? v : s + v;
        }
        a.buf[s0 + i] = s;
        for (int dq = 1; dq < N; ++dq) {
          const int q = (r + dq) % N;
          a.peer_ag[q][((size_t)par * kP2PMaxWorld + r) * a.slice_cap + i] = s;
        }
      }
    }
    publish(a, 1, par, c, call);      // (also when this rank gave up: the peers must not hang; seq[2] tells)
  }
  // ---- C: copy the reduced slices of the other owners ------------------------------------
  {
    const bool ok = await(a, 1, par, c, call);
    alive = alive && ok;
    if (alive) {
      for (int dq = 1; dq < N; ++dq) {
        const int q = (r + dq) % N;
        const int64_t s0 = (int64_t)q * per;
        const int64_t len = max((int64_t)0, min(a.n, s0 + per) - s0);
        const int64_t lo = min(c0, len), hi = min(c0 + cper, len);
        const float* src = a.ag + ((size_t)par * kP2PMaxWorld + q) * a.slice_cap;
        float* dst = a.buf + s0;
        const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
        for (int64_t i = lo + 4 * t; i < hi4; i += 4 * (int64_t)nt)
          *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
        for (int64_t i = hi4 + t; i < hi; i += nt) dst[i] = src[i];
      }
    }
  }
  // ---- the last workgroup to leave advances the call counter -------------------------------
  __syncthreads();
  if (t == 0) {
    if (!alive) atomicCAS((unsigned long long*)&a.seq[2], 0ull, (unsigned long long)call);
    __threadfence();
    const unsigned long long k = atomicAdd((unsigned long long*)&a.seq[1], 1ull);
    if (k == (unsigned long long)gridDim.x - 1) {
      a.seq[1] = 0;
      __hip_atomic_store(&a.seq[0], call, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

void launch_p2p_allreduce(const P2PLaunch& L, hipStream_t st) {
  P2PArgs a{};
  a.buf = L.buf;
  a.n = L.n;
  a.rank = L.rank;
  a.world = L.world;
  a.nchunk = L.nchunk;
  a.slice_cap = L.slice_cap;
  a.rs = L.rs;
  a.ag = L.ag;
  a.flags = L.flags;
  a.seq = L.seq;
  for (int i = 0; i < L.world; ++i) {
    a.peer_rs[i] = L.peer_rs[i];
    a.peer_ag[i] = L.peer_ag[i];
    a.peer_flags[i] = L.peer_flags[i];
  }
  hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(L.nchunk), dim3(512), 0, st, a);
}

}  // namespace spmf
