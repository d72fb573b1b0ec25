// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path: GRU recurrence (cluster kernels)
// (one translation unit per kernel family; shared device helpers in ou_dev.h, cross-file launchers in ou_internal.h)
#include "ou_kernels.h"
#include "ou_internal.h"
#include "ou_dev.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ou {

// =========================================================================================================
// GRU recurrence (torch.nn.GRU semantics; score.py:83-89,116 / condition.py:173-179,212)
//   W_hh of one direction is 3H x H fp32 (786 KB at H = 256): larger than one CU's LDS and VGPR file, so a
//   cluster of HB = H/64 workgroups (512 threads each, one per CU) keeps it resident in registers:
//   workgroup g owns hidden units [64g, 64g+64) = 192 gate rows; thread (rg, cg) holds rows of units
//   {2rg, 2rg+1} x columns {4cg + 64i + 0..3}.  Per time step: 96..144 FMAs per thread, a 16-lane DPP
//   row reduction, the gate math, then the 64 new hidden values are published to the other workgroups as
//   8-byte {step tag, value} granules with relaxed agent-scope stores (write-through to L2) and gathered
//   by one polling wave -- no fence, no flag (the tag is the flag).  Spins are bounded; a timeout raises
//   the status word instead of hanging the GPU.
// =========================================================================================================
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(t);
}
// sum over the 8 lanes of an aligned 8-lane group (every lane ends with the total)
__device__ __forceinline__ float row8_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror: lane i <-> 7-i within each 8
  return v;
}
// sum over the 16 lanes of a DPP row (every lane ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v = row8_sum(v);
  v = dpp_add<0x140>(v);  // row_mirror: lane i <-> 15-i
  return v;
}
// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): the gate math is
// a serial chain on the critical path of every GRU time step, libm-grade expf/tanhf/division cost ~100 dependent
// instructions there.  Absolute error ~2e-7, two orders of magnitude inside the parity gate.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(1 + e^{2x});  e^{2x} -> inf gives 1, -> 0 gives -1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}
constexpr unsigned GRU_SPIN_LIMIT = 4000000u;

#ifdef OU_EXPERIMENTS  // round 1's polling-wave kernel (OU_GRU_V=1): built with `make EXPERIMENTS=1` only
// Thread mapping: a direction's H hidden units are split over NWG = H/UPW workgroups of NT threads;
// LPU = NT/UPW lanes share one unit: thread (u, cg) = (tid/LPU, tid%LPU) holds, for unit UPW*g + u, the three gate
// rows (r, z, n) x columns {4cg + 4*LPU*i + 0..3, i < NI = H/(4*LPU)} in registers (gathered from the canonical
// row-major W_hh at kernel start).  Per step: NI ds_read_b128 of h, 6*NI v_pk_fma_f32, a DPP reduction of the 3 gate
// sums over the LPU lanes of the unit, the gate math in lane cg == 0, publish.
// Variants (chosen by launch_gru from the batch size): <64 units, 512 thr> = 4 workgroups per direction at H = 256
// (throughput: 8 CUs per utterance), <32, 512> = 8, <16, 256> = 16 (latency: one wave per SIMD, least work per step).
template <int HB, int UPW, int NT>
__global__ __launch_bounds__(NT) void gru_cluster_kernel(GruArgs p, int nclusters) {
  constexpr int H = 64 * HB, LPU = NT / UPW, NI = H / (4 * LPU), NR = 12 * NI, NWG = H / UPW;
  static_assert(LPU == 8 || LPU == 16, "8 or 16 lanes per hidden unit");
  static_assert(H % (4 * LPU) == 0, "column blocks");
  __shared__ __attribute__((aligned(16))) float hbuf[2][H];
  __shared__ int abort_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  // Workgroup -> (cluster, member): the dispatcher places block i on XCD i % 8 (observed, speed only), so the
  // NWG members of a cluster are given ids that are congruent mod 8 and share one L2.  Correctness does not
  // depend on it: the exchange below is agent-scope.
  const int bid = blockIdx.x;
  // clusters are dealt to XCDs by block id (block i runs on XCD (i + k) % 8, k fixed per process); `xcd_rot` turns the deal so
  // that the few clusters of a small batch land on DIFFERENT XCDs in every lane of a multi-lane process (ou_set_lanes)
  const int xcd = (bid - p.xcd_rot) & 7, slot = bid >> 3;
  const int cluster = xcd + 8 * (slot / NWG);
  const int g = slot % NWG;
  if (cluster >= nclusters) return;
  const int dir = cluster & 1, b = cluster >> 1;
  const int ul = tid / LPU, cg = tid % LPU;
  const int unit = g * UPW + ul;
  const int T = p.T;

  f32x2 w[NR / 2];
  {
    const float* wd = p.whh + (size_t)dir * 3 * H * H;
#pragma unroll
    for (int gt = 0; gt < 3; gt++)
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const float4 v = *reinterpret_cast<const float4*>(wd + (size_t)(gt * H + unit) * H + cg * 4 + 4 * LPU * i);
        w[(gt * NI + i) * 2] = f32x2{v.x, v.y};
        w[(gt * NI + i) * 2 + 1] = f32x2{v.z, v.w};
      }
  }
  for (int i = tid; i < 2 * H; i += NT) (&hbuf[0][0])[i] = 0.f;
  if (tid == 0) abort_flag = 0;

  const bool fin = cg == 0;
  const float bhn = p.bhn[dir * H + unit];
  const float* gxb = p.gx + ((size_t)b * 6 * H + (size_t)dir * 3 * H) * T;
  const float* gx_r = gxb + (size_t)unit * T;
  const float* gx_z = gxb + (size_t)(H + unit) * T;
  const float* gx_n = gxb + (size_t)(2 * H + unit) * T;
  const size_t orow = ((size_t)b * 2 * H + (size_t)dir * H + unit) * T;
  unsigned long long* xq = p.xchg + ((size_t)(b * 2 + dir) * 2) * H;
  const bool has_res = p.res != nullptr;

  // One-time rendezvous: every member posts the id of the XCD it runs on (granule g*UPW of buffer 0, which is not
  // written again before all members have passed step 0).  When the whole cluster shares one XCD -- the normal case,
  // see the block mapping above -- the per-step publishes can be ordinary stores: the vector L1 is write-through, so
  // they land in the L2 that serves every poller's sc1 (agent-scope) load, without the write-through to the memory
  // side that an agent-scope store adds (measured: -6 % per GRU launch, and finer splits stop losing to store
  // traffic).  Anything else keeps agent-scope stores.
  __shared__ int plain_flag;
  if (NWG > 1 && tid < 64) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xFu;
    constexpr unsigned RTAG = 0x80000000u;
    if (lane == 0)
      __hip_atomic_store(xq + (size_t)g * UPW, ((unsigned long long)RTAG << 32) | xcc, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    bool same = true, fail = false;
    if (lane < NWG) {
      unsigned spins = 0;
      unsigned long long v;
      while (true) {
        v = __hip_atomic_load(xq + (size_t)lane * UPW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == RTAG) break;
        if (++spins > GRU_SPIN_LIMIT) { fail = true; break; }
      }
      same = !fail && (unsigned)v == xcc;
    }
    const bool all_same = __builtin_amdgcn_ballot_w64(!same) == 0ull;
    if (lane == 0) plain_flag = (all_same && !p.agent_stores) ? 1 : 0;
    if (fail) atomicOr(p.err, 1u);
  }

  int t = dir ? T - 1 : 0;
  const int dt = dir ? -1 : 1;
  float xr = 0.f, xz = 0.f, xn = 0.f, rs = 0.f;
  if (fin) {
    xr = gx_r[t]; xz = gx_z[t]; xn = gx_n[t];
    if (has_res) rs = p.res[orow + t];
  }
  // Everything loaded so far (weights, bhn, first-step inputs) is first USED inside the loop; without this the
  // compiler places a vmcnt(0) wait at that first use -- in every iteration, right behind the prefetch loads
  // issued there, which exposes a full memory latency per time step.  vmcnt(0), expcnt/lgkmcnt untouched:
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  const bool plain = NWG > 1 && __builtin_amdgcn_readfirstlane(plain_flag) != 0;

  long long c_comp = 0, c_poll = 0, c_bar = 0, c_mv = 0, c_red = 0, c_gate = 0;
  const bool ts_on = p.tstamps != nullptr;
  for (int step = 0; step < T; step++, t += dt) {
    const int cur = step & 1;
    long long q0 = 0, q1 = 0, q2 = 0;
    if (ts_on) q0 = __builtin_readcyclecounter();
    const float hp = hbuf[cur][unit];  // read with the matvec operands, off the gate chain
    f32x2 acc[3][2];
#pragma unroll
    for (int gt = 0; gt < 3; gt++) { acc[gt][0] = 0.f; acc[gt][1] = 0.f; }
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const float4 hv = *reinterpret_cast<const float4*>(&hbuf[cur][cg * 4 + 4 * LPU * i]);
      const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
#pragma unroll
      for (int gt = 0; gt < 3; gt++) {
        const int r = (gt * NI + i) * 2;
        acc[gt][0] = __builtin_elementwise_fma(w[r], h01, acc[gt][0]);
        acc[gt][1] = __builtin_elementwise_fma(w[r + 1], h23, acc[gt][1]);
      }
    }
    long long qa = 0, qb = 0;
    if (ts_on) qa = __builtin_readcyclecounter();
    float hs[3];
#pragma unroll
    for (int gt = 0; gt < 3; gt++) {
      const float part = (acc[gt][0].x + acc[gt][0].y) + (acc[gt][1].x + acc[gt][1].y);
      hs[gt] = LPU == 8 ? row8_sum(part) : row16_sum(part);
    }

    if (ts_on) qb = __builtin_readcyclecounter();
    // next step's input-projection / residual values: issued now, consumed one iteration later, so that no
    // global-load latency ever sits between the gate math and the publish below
    float nxr = 0.f, nxz = 0.f, nxn = 0.f, nrs = 0.f;
    if (fin && step + 1 < T) {
      nxr = gx_r[t + dt]; nxz = gx_z[t + dt]; nxn = gx_n[t + dt];
      if (has_res) nrs = p.res[orow + t + dt];
    }

    if (fin) {
      const float r = sigmoidf_(xr + hs[0]);
      const float z = sigmoidf_(xz + hs[1]);
      const float n = tanhf_(xn + r * (hs[2] + bhn));
      const float hnew = (hp - n) * z + n;
      if (NWG > 1) {  // publish first: the other workgroups are waiting on this
        unsigned long long gran = ((unsigned long long)(unsigned)(step + 1) << 32) | (unsigned)__float_as_int(hnew);
        unsigned long long* dst = xq + (size_t)(cur ^ 1) * H + unit;
        if (plain) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(gran) : "memory");
        else __hip_atomic_store(dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      hbuf[cur ^ 1][unit] = hnew;
      p.out[orow + t] = has_res ? (hnew + rs) * p.res_scale : hnew;
    }
    xr = nxr; xz = nxz; xn = nxn; rs = nrs;
    if (ts_on) { q1 = __builtin_readcyclecounter(); c_mv += qa - q0; c_red += qb - qa; c_gate += q1 - qb; }

    if (NWG > 1 && tid < 64) {
      // gather the other workgroups' slices: lane l polls granules l, l+64, ...; all polls in flight together
      const unsigned tag = (unsigned)(step + 1);
      unsigned long long* src = xq + (size_t)(cur ^ 1) * H + lane;
      unsigned long long v[HB];
      unsigned spins = 0;
      if (p.poll_backoff > 0) __builtin_amdgcn_s_sleep(8);   // ~512 cycles: nothing can have arrived yet
      if (p.poll_backoff > 1) __builtin_amdgcn_s_sleep(8);
      while (true) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < HB; k++) {
          const bool own = (k * 64 + lane) / UPW == g;
          v[k] = own ? ((unsigned long long)tag << 32)
                     : __hip_atomic_load(src + k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((unsigned)(v[k] >> 32) == tag);
        }
        if (ok) break;
        if (++spins > GRU_SPIN_LIMIT) { abort_flag = 1; break; }
        // safety net as in gru_ring_kernel: a plain publish has no visibility deadline; after a long wait (~1 ms; the gate
        // lanes have written this step's values to LDS long before) repeat this workgroup's own granules as system-scope
        // write-through stores
        if (plain && (spins & 1023u) == 1023u) {
#pragma unroll
          for (int k = 0; k < HB; k++)
            if ((k * 64 + lane) / UPW == g) {
              const float hv = *reinterpret_cast<volatile float*>(&hbuf[cur ^ 1][k * 64 + lane]);
              const unsigned long long gran = ((unsigned long long)tag << 32) | (unsigned)__float_as_int(hv);
              unsigned long long* dst = src + k * 64;
              asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(gran) : "memory");
            }
        }
      }
#pragma unroll
      for (int k = 0; k < HB; k++)
        if ((k * 64 + lane) / UPW != g) hbuf[cur ^ 1][k * 64 + lane] = __int_as_float((int)(unsigned)v[k]);
    }
    if (ts_on) q2 = __builtin_readcyclecounter();
    __syncthreads();
    if (ts_on) { long long q3 = __builtin_readcyclecounter(); c_comp += q1 - q0; c_poll += q2 - q1; c_bar += q3 - q2; }
    if (NWG > 1 && abort_flag) {
      if (tid == 0) atomicOr(p.err, 1u);
      break;
    }
  }
  if (ts_on && lane == 0) {
    long long* o = p.tstamps + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 8;
    o[0] = c_comp; o[1] = c_poll; o[2] = c_bar; o[3] = T; o[4] = c_mv; o[5] = c_red; o[6] = c_gate;
  }
}

#endif  // OU_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------------
// GRU recurrence, second generation ("ring"): same cluster decomposition, different exchange.
//   * every WAVE gathers the h columns its lanes need straight from L2 into registers (volatile agent-scope 16-byte
//     buffer loads = two {value, tag} granules each): no polling wave, no LDS hop, no workgroup barrier -- the four
//     waves of a workgroup run unsynchronised, each ordered only by the tags it reads;
//   * tags never repeat: tag(step s of this launch) = epoch + s with a device-side epoch that the last block of a
//     launch advances by T + 1 (graph-replay safe, no per-launch memset).  Stale granules therefore always carry
//     SMALLER tags, so "all tags arrived" is one v_min3 tree + one compare;
//   * granule = {h, tag}: the matvec multiplies a (W_r, W_z) row pair by the granule's low half with one v_pk_fma_f32
//     (op_sel_hi = 0 on the h operand) and W_n by a scalar FMA -- the loaded registers are the FMA operands, no
//     repacking;
//   * the one-time XCD rendezvous uses the cluster's own rendezvous granules with tag = epoch.
// Double buffering by step parity is enough without barriers: h_{s+2} overwrites h_s only after its writer has read all of
// h_{s+1}, and every wave publishes its part of h_{s+1} only after it has finished reading h_s.
// ---------------------------------------------------------------------------------------------------------
// s_waitcnt vmcnt(0) that the register allocator sees as the producer of the gathered registers (the loads themselves are
// inline asm, invisible to the compiler's own wait-count insertion)
template <int N>
__device__ __forceinline__ void gather_wait(u32x4 (&hv)[N]) {
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]));
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]));
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]));
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]));
  else if constexpr (N == 8)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]),
                 "+v"(hv[6]), "+v"(hv[7]));
  else if constexpr (N == 12)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]),
                 "+v"(hv[6]), "+v"(hv[7]), "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]));
  else if constexpr (N == 16)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]),
                 "+v"(hv[6]), "+v"(hv[7]), "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]), "+v"(hv[12]),
                 "+v"(hv[13]), "+v"(hv[14]), "+v"(hv[15]));
  else static_assert(N == 2, "gather_wait: unsupported register count");
}

// First-recovery record, out of line so that the hot loop does not change: when a hand-off has not arrived after 256 poll
// rounds, look at the first stale granule of this lane's columns once more with three kinds of loads and leave what they
// return -- and where this wave runs now vs. at the rendezvous -- in status words 21..29 (first event of a workspace only).
// Reading the record: sc1 == want            -> the publish was only late (a member was not scheduled / not resident);
//                     sc1 != want, atomic == want (or sc0 sc1 == want) -> the line sits where an L2-served agent-scope load of
//                                               THIS CU does not see it: the writer or the reader is not on the cluster's XCD
//                                               any more (xcc now != xcc at the rendezvous), e.g. after a context save / restore;
//                     nothing == want         -> the writer has not stored it: look at that member's own wait record.
__device__ __attribute__((noinline)) void gru_stale_probe(const unsigned long long* buf, int col0, int ncol, int lstride,
                                                          unsigned want, unsigned* err, unsigned who, unsigned xcc_then,
                                                          unsigned step, int grp) {
  for (int i = 0; i < ncol; i++) {
    const int goff = col0 + (i / grp) * lstride + (i % grp);
    const unsigned long long* g = buf + goff;
    u32x2 a, b, c;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(g) : "memory");
    if (a.y == want) continue;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(b) : "v"(g) : "memory");
    const unsigned long long v = __hip_atomic_fetch_or(const_cast<unsigned long long*>(g), 0ull, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
    c = u32x2{(unsigned)v, (unsigned)(v >> 32)};
    // status word 33: events where the granule HAD arrived for a system-scope load or an atomic but not for the agent-scope
    // load the gather uses -- the cheap publish form really failed (what the host switches the publish mode on).  Everything
    // else counted in word 20 is a member that was late (descheduled / starved by the kernels of other streams).
    // (the granule may simply ARRIVE between the first load and the other two: only if an agent-scope load issued after them
    // still misses it was it invisible to the gather)
    if (b.y == want || c.y == want) {
      u32x2 a2;
      asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a2) : "v"(g) : "memory");
      if (a2.y != want) atomicAdd(err + 33, 1u);
    }
    if (atomicAdd(err + 21, 1u) == 0u) {
      unsigned now;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(now));
      err[22] = who; err[23] = (unsigned)goff; err[24] = want;
      err[25] = a.y; err[26] = b.y; err[27] = c.y;
      asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(g) : "memory");
      err[28] = a.y;
      err[29] = (step << 16) | ((xcc_then & 0xFFu) << 8) | (now & 0xFFu);
    }
    return;
  }
}

// Rare-path bookkeeping of the ring kernel, out of line and with few arguments: inlined, the diagnostics below cost the hot
// loop 10 SGPRs (106 -> spills to VGPR lanes, 17 v_readlane in the step loop) and 20 VGPRs, +25 us per 401-frame pass.
// A workgroup that has been waiting for ~2 ms leaves its position in its rendezvous slot: {epoch, step << 8 | XCC now << 4 | XCC
// at the rendezvous} -- the tag stays the epoch, late members still pass the rendezvous.
__device__ __attribute__((noinline)) void gru_note_long_wait(unsigned long long* slot, unsigned epoch, unsigned xcc_then,
                                                             unsigned step) {
  unsigned now;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(now));
  const unsigned long long pos = ((unsigned long long)epoch << 32) | ((now & 0xFu) << 4) | (xcc_then & 0xFu) | (step << 8);
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(slot), "v"(pos) : "memory");
}
// Time-out report (first reporter of a workspace only): who waited for what, and where every member of the cluster is.
//   ids = cluster << 16 | member << 8 | plain-publish flag, pos = step << 8 | XCC at the rendezvous
__device__ __attribute__((noinline)) void gru_timeout_report(unsigned* err, const unsigned long long* slots, int nwg,
                                                             unsigned epoch, unsigned ids, unsigned pos, unsigned m,
                                                             unsigned mx, unsigned want) {
  if ((atomicOr(err, 4u) & 4u) != 0u) return;
  unsigned now;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(now));
  err[11] = 0x100u | (now & 0xFu);  // has the workgroup been moved since the rendezvous? (context save / restore)
  err[12] = ids >> 16; err[13] = (ids >> 8) & 0xFFu; err[14] = pos >> 8; err[15] = m; err[16] = want;
  err[17] = pos & 0xFFu; err[18] = ids & 1u; err[19] = (unsigned)blockIdx.x;
  err[32] = mx;
  for (int i = 0; i < nwg && i < 24; i++) {  // step << 8 | xcc of every member that ever waited ~2 ms
    const unsigned long long v = __hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    err[36 + i] = (unsigned)(v >> 32) == epoch ? (unsigned)v : 0xFFFFFFFFu;
  }
}

constexpr unsigned GRU_EPOCH_WRAP = 0x7F000000u;  // past this the last block of a launch clears the exchange area

// WIDE: the gather layout with no redundant loads.  In the default layout a unit's 16 lanes hold all H columns, so the four
// units of a wave each fetch the same 2 KB (H = 256) per poll round: 8 dwordx4 loads per lane, 32 wave-loads of 16 TA cycles
// per workgroup and round -- the round is bound by the CU's address path (~500 cycles), not by the L2 latency (~200).  WIDE:
// lane l holds columns 2l, 2l + 1 (+ 128 i) for ALL four units of its wave: HB / 2 loads per lane and round; the 12 (unit,
// gate) partial sums are folded 64 -> 16 lanes by two swap levels (v_permlane32_swap / v_permlane16_swap halve the value
// count as they halve the lane count: 6 + 3 swaps), after which row r of the wave holds unit r's sums exactly as in the
// default layout.
// (Tried on top of WIDE and dropped: the four waves of a workgroup sharing the gather -- wave w polls granules [w H / 4,
// (w + 1) H / 4), 32 instead of 128 requests per line and step, and passes them on through a tag-checked LDS copy of the
// exchange buffer.  Gather 1 225 instead of 738 cycles per step: the L2 request rate is not what a poll round waits for.
// Measured for the record (OU_GRU_BACKOFF=10..14, tools/gru_ts.py): the first poll round succeeds on 92-95 % of the steps; a
// wave's two stores are acknowledged after ~230 cycles; one isolated 8-byte load, sc1 or plain, quiet or just-written line,
// takes ~320 cycles; every cycle a wave spends between its publish and its poll comes back one-to-one in everybody's step
// time -- the clusters run in lock step, the step is compute + one store latency + one load latency + the skew of 64-128 waves.)
// Gate pre-activations are carried PRE-SCALED through the ring kernel: S (a) for the sigmoids, T (y) for the tanh, with
//   sigmoid(a) = 1 / (1 + 2^(S a)),  S = -log2(e);    tanh(y) = 1 - 2 / (1 + 2^(T y)),  T = 2 log2(e)
// -- the recurrent weights and b_hn are scaled once per launch when they are gathered into registers, the input projections when
// a chunk is staged into LDS, and x_r / x_z / b_hn enter as the INITIAL VALUE of the fin lane's partial sums: the serial
// chain of a step behind the gather (it is instruction latency, one wave per SIMD: 505 cycles in round 5) loses the scale
// multiplies, the bias adds and -- with explicit fmas -- three more dependent operations.
constexpr float kGateS = -1.44269504088896341f, kGateT = 2.88539008177792681f;
template <int HB, int UPW, bool WIDE = false>
__global__ __launch_bounds__(256) void gru_ring_kernel(GruArgs p, int nclusters) {
  constexpr int H = 64 * HB, NT = 256, LPU = NT / UPW, NC = WIDE ? HB : H / LPU, NI = WIDE ? HB / 2 : NC / 4, NWG = H / UPW;
  constexpr int WU = UPW / 4;  // units per wave
  static_assert(!WIDE || ((UPW == 16 || UPW == 8) && HB % 2 == 0), "wide gather: 4 / 2 units per wave, whole granule pairs");
  constexpr int CSTRIDE = 2 * H + 64;  // granules per cluster: two parity buffers + rendezvous slots
  static_assert(LPU == 8 || LPU == 16 || LPU == 32, "8, 16 or 32 lanes per hidden unit");
  static_assert((WIDE || NC % 4 == 0) && NWG > 1 && NWG < 64, "column blocks / rendezvous slots (slot 63 = the mode flag)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int bid = blockIdx.x;
  // clusters are dealt to XCDs by block id (block i runs on XCD (i + k) % 8, k fixed per process); `xcd_rot` turns the deal so
  // that the few clusters of a small batch land on DIFFERENT XCDs in every lane of a multi-lane process (ou_set_lanes)
  const int xcd = (bid - p.xcd_rot) & 7, slot = bid >> 3;
  const int cluster = xcd + 8 * (slot / NWG);
  const int g = slot % NWG;
  // tags of this launch: epoch (rendezvous), epoch + s (h after s steps).  The stored counter starts at 0 in a freshly
  // cleared workspace, whose granules carry tag 0 -> + 1
  const unsigned epoch = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int T = p.T;
  const bool ts_on = p.tstamps != nullptr;
  long long c_poll = 0, c_comp = 0, r_start = 0, r_loop = 0;
  unsigned c_rounds = 0;
  long long c_ack = 0;
  if (ts_on) r_start = (long long)__builtin_amdgcn_s_memrealtime();
  if (cluster < nclusters) {
    const int dir = cluster & 1, b = cluster >> 1;
    const int ul = tid / LPU, cg = tid % LPU;
    const int unit = g * UPW + ul;

    // this lane's weights: (W_r, W_z) row pairs and W_n for columns 4cg + 4 LPU i + {0..3}
    // (WIDE: for the wave's four units u and columns 2 lane + 128 i + {0, 1}: index u * HB + 2 i + {0, 1})
    constexpr int NW = WIDE ? WU * HB : NC;
    f32x2 wrz[NW];
    float wn[NW];
    if constexpr (WIDE) {
#pragma unroll
      for (int u = 0; u < WU; u++) {
        const float* wd = p.whh + (size_t)dir * 3 * H * H + (size_t)(g * UPW + (tid >> 6) * WU + u) * H + 2 * lane;
#pragma unroll
        for (int i = 0; i < NI; i++) {
          const float2 vr = *reinterpret_cast<const float2*>(wd + 128 * i);
          const float2 vz = *reinterpret_cast<const float2*>(wd + (size_t)H * H + 128 * i);
          const float2 vn = *reinterpret_cast<const float2*>(wd + (size_t)2 * H * H + 128 * i);
          wrz[u * HB + 2 * i] = f32x2{vr.x, vz.x} * kGateS; wrz[u * HB + 2 * i + 1] = f32x2{vr.y, vz.y} * kGateS;
          wn[u * HB + 2 * i] = vn.x * kGateT; wn[u * HB + 2 * i + 1] = vn.y * kGateT;
        }
      }
    } else {
      const float* wd = p.whh + (size_t)dir * 3 * H * H + (size_t)unit * H + cg * 4;
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const float4 vr = *reinterpret_cast<const float4*>(wd + 4 * LPU * i);
        const float4 vz = *reinterpret_cast<const float4*>(wd + (size_t)H * H + 4 * LPU * i);
        const float4 vn = *reinterpret_cast<const float4*>(wd + (size_t)2 * H * H + 4 * LPU * i);
        wrz[4 * i + 0] = f32x2{vr.x, vz.x} * kGateS; wrz[4 * i + 1] = f32x2{vr.y, vz.y} * kGateS;
        wrz[4 * i + 2] = f32x2{vr.z, vz.z} * kGateS; wrz[4 * i + 3] = f32x2{vr.w, vz.w} * kGateS;
        wn[4 * i + 0] = vn.x * kGateT; wn[4 * i + 1] = vn.y * kGateT; wn[4 * i + 2] = vn.z * kGateT; wn[4 * i + 3] = vn.w * kGateT;
      }
    }
    // the lane that ends up with the unit's gate sums: any lane of a 8 / 16-lane group (all-reduce), the upper row of a
    // 32-lane group (row_bcast:15 adds the lower row's total into the upper row only)
    const bool fin = cg == (LPU == 32 ? 16 : 0);
    const float bhn = p.bhn[dir * H + unit] * kGateT;
    const float* gxb = p.gx + ((size_t)b * 6 * H + (size_t)dir * 3 * H) * T;
    const size_t orow = ((size_t)b * 2 * H + (size_t)dir * H + unit) * T;
    unsigned long long* xq = p.xchg + (size_t)cluster * CSTRIDE;
    const bool has_res = p.res != nullptr;

    // one-time rendezvous (also proves that every member of the cluster is resident): member g posts {xcc, epoch}
    bool sysmode = (p.dbg & 2) != 0;  // system-scope publishes for the rest of this launch, see the safety net
    bool plain = false;
    unsigned xcc;
    {
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 0xFu;
      if (tid == 0)
        __hip_atomic_store(xq + 2 * H + g, ((unsigned long long)epoch << 32) | xcc, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      bool same = true, fail = false;
      if (lane < NWG) {
        unsigned spins = 0;
        unsigned long long v;
        while (true) {
          v = __hip_atomic_load(xq + 2 * H + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(v >> 32) == epoch) break;
          if (++spins > GRU_SPIN_LIMIT) { fail = true; break; }
        }
        same = !fail && (unsigned)v == xcc;
      }
      plain = __builtin_amdgcn_ballot_w64(!same) == 0ull && !p.agent_stores;
      if (!plain && tid == 0) atomicAdd(p.err + 30, 1u);  // diagnostics: workgroups whose cluster spans XCDs
      if (fail) { atomicOr(p.err, 2u); p.err[8] = (unsigned)cluster; p.err[9] = (unsigned)g; p.err[10] = epoch; }
    }

    // Input projections (gx rows r, z, n of this wave's units) and the residual row are staged through a wave-private
    // LDS ring in chunks of CH time steps: the global loads of chunk c + 2 are issued at the start of chunk c (coalesced
    // along time, a whole chunk of steps to land) and parked in registers, moved to LDS one chunk later and read from
    // there by the gate lanes -- the per-step loop touches global memory only for the exchange and the output store.
    constexpr int CH = 32, UW = UPW / 4;           // steps per chunk, units per wave
    constexpr int ROWS = 4 * UW, PER = ROWS * CH / 64;  // staged rows per wave (unit x {r, z, n, res}), floats per lane
    static_assert(ROWS * CH % 64 == 0 && 64 % ROWS == 0, "staging map");
    constexpr int LPR = 64 / ROWS;                 // lanes per staged row
    __shared__ float stage[4][2][ROWS][CH];
    const int wv = tid >> 6;
    const int srow = lane / LPR, sq = lane % LPR;  // this lane loads steps sq * PER .. + PER of staged row srow
    const int s_unit = g * UPW + wv * UW + (srow >> 2), s_kind = srow & 3;
    const float* s_src = s_kind == 3 ? (has_res ? p.res + ((size_t)b * 2 * H + (size_t)dir * H + s_unit) * T : nullptr)
                                     : gxb + (size_t)(s_kind * H + s_unit) * T;
    float park[PER];
    auto fetch_chunk = [&](int c) {  // -> park[]: steps c*CH + sq*PER + j
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const int sidx = c * CH + sq * PER + j;
        const int tt = dir ? T - 1 - sidx : sidx;
        park[j] = (s_src && sidx < T) ? s_src[tt] : 0.f;
      }
    };
    // (the pre-activations are staged PRE-SCALED for the exp2-based gates -- off the per-step chain: see the gate math below)
    const float s_scale = s_kind == 3 ? 1.0f : (s_kind == 2 ? kGateT : kGateS);
    auto park_to_lds = [&](int c) {
#pragma unroll
      for (int j = 0; j < PER; j++) stage[wv][c & 1][srow][sq * PER + j] = park[j] * s_scale;
    };
    fetch_chunk(0);
    park_to_lds(0);
    fetch_chunk(1);
    const int ulw = ul - wv * UW;                  // this lane's unit within its wave
    // 1 on the fin lane of unit u of this wave, 0 on every other lane -- and T b_hn there: what the lane's partial sums start from
    float sel[WIDE ? UW : 1], inn[WIDE ? UW : 1];
#pragma unroll
    for (int u = 0; u < (WIDE ? UW : 1); u++) {
      sel[u] = (fin && (!WIDE || ulw == u)) ? 1.0f : 0.0f;
      inn[u] = sel[u] * bhn;
    }
    float hprev = 0.f;
    int t = dir ? T - 1 : 0;
    const int dt = dir ? -1 : 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // weights landed: no vmcnt(0) inside the loop on their account

    if (ts_on) r_loop = (long long)__builtin_amdgcn_s_memrealtime();
    bool fast_pub = plain && !sysmode;                                   // plain-store publishes
    const int inject_step = ((p.dbg & 4) && g == 1) ? 50 : -1;           // fault injection (tests)
    for (int step = 0; step < T; step++, t += dt) {
      long long q0 = 0, q1 = 0;
      if (ts_on) q0 = __builtin_readcyclecounter();
      const int cidx = step / CH, soff = step % CH;
      if (soff == 0 && step > 0) {  // chunk boundary: park -> LDS (chunk cidx), start fetching chunk cidx + 1
        park_to_lds(cidx);
        fetch_chunk(cidx + 1);
      }
      float xr = 0.f, xz = 0.f, xn = 0.f, rs = 0.f;
      if (fin) {
        const float* sp = &stage[wv][cidx & 1][ulw * 4][soff];
        xr = sp[0]; xz = sp[CH]; xn = sp[2 * CH]; rs = sp[3 * CH];  // (pre-scaled: S x_r, S x_z, T x_n)
      }
      // initial values of this lane's partial sums (computed here, under the gather): every lane's partial for unit u ends up
      // in unit u's total, so the fin lane of unit u contributes S x_r, S x_z and T b_hn through its own
      // The lane's partial sums start from these (x 1 on the fin lane of unit u, x 0 elsewhere: exact).  Formed BEHIND the issue
      // of the gather's loads, under their latency: in front of them the wait for the LDS reads above sits between this wave's
      // publish and its poll -- measured: +80 cycles per step, the whole gain of the shorter chain and more (round 6, A / B).
      f32x2 irz[WIDE ? WU : 1];
      auto form_init = [&]() {
        if constexpr (WIDE) {
#pragma unroll
          for (int u = 0; u < WU; u++) irz[u] = f32x2{xr, xz} * sel[u];
        } else {
          irz[0] = f32x2{xr, xz};  // (0 on the lanes that are not fin)
        }
#pragma unroll
        for (int u = 0; u < (WIDE ? WU : 1); u++) asm volatile("" : "+v"(irz[u]));  // pinned between the loads and their wait
      };
      // ---- h_step: zero at step 0, else gathered from the parity buffer (all granules must carry tag epoch + step)
      u32x4 hv[NC / 2];
      if (step == 0) {
#pragma unroll
        for (int k = 0; k < NC / 2; k++) hv[k] = u32x4{0u, 0u, 0u, 0u};
        form_init();
      } else {
        const unsigned want = epoch + (unsigned)step;
        const unsigned long long* src = xq + (size_t)(step & 1) * H + (WIDE ? 2 * lane : cg * 4);
        unsigned spins = 0;
        // experiment (OU_GRU_BACKOFF = 6..9): nothing can have arrived right after this wave's own publish -- the first poll
        // rounds only load the L2 channels that the other members' stores have to get through
        if (p.poll_backoff >= 6) {
          if (p.poll_backoff == 6) __builtin_amdgcn_s_sleep(1);
          else if (p.poll_backoff == 7) __builtin_amdgcn_s_sleep(2);
          else if (p.poll_backoff == 8) __builtin_amdgcn_s_sleep(3);
          else __builtin_amdgcn_s_sleep(5);
        }
        while (true) {
          // 16-byte loads = two granules each; asm: the compiler must neither cache the values nor pick the scope.
          // sc1 = agent scope.  (sc0 -- workgroup scope -- polls were tried for clusters that share an XCD: they never
          // observe the other CUs' publishes; kept behind OU_GRU_BACKOFF=3 for the record.)
          if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NI; i++)
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(hv[i]) : "v"(src), "n"(1024 * i) : "memory");
          } else if (plain && p.poll_backoff == 3) {
#pragma unroll
            for (int i = 0; i < NI; i++) {
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0"
                           : "=v"(hv[2 * i]) : "v"(src), "n"(4 * LPU * i * 8) : "memory");
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0"
                           : "=v"(hv[2 * i + 1]) : "v"(src), "n"(4 * LPU * i * 8 + 16) : "memory");
            }
          } else if (p.poll_backoff == 4) {  // experiment: system-scope polls
#pragma unroll
            for (int i = 0; i < NI; i++) {
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1"
                           : "=v"(hv[2 * i]) : "v"(src), "n"(4 * LPU * i * 8) : "memory");
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1"
                           : "=v"(hv[2 * i + 1]) : "v"(src), "n"(4 * LPU * i * 8 + 16) : "memory");
            }
          } else {
            if (p.poll_backoff == 5 && (spins & 15u) == 15u) asm volatile("buffer_inv sc1" ::: "memory");  // experiment
#pragma unroll
            for (int i = 0; i < NI; i++) {
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1"
                           : "=v"(hv[2 * i]) : "v"(src), "n"(4 * LPU * i * 8) : "memory");
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1"
                           : "=v"(hv[2 * i + 1]) : "v"(src), "n"(4 * LPU * i * 8 + 16) : "memory");
            }
          }
          form_init();
          gather_wait(hv);
          // all tags arrived <=> the smallest one is the wanted one: stale granules always carry SMALLER tags (tags are
          // monotonic per exchange area; a larger one could only come from a buffer that ou_workspace_init has not prepared,
          // which the C ABI refuses, and an epoch wrap clears the whole area).  An explicit min == max == want test costs a
          // second reduction tree on the critical path of every step: +50 cycles per step, measured.
          unsigned m = hv[0].y < hv[0].w ? hv[0].y : hv[0].w;
#pragma unroll
          for (int k = 1; k < NC / 2; k++) {
            const unsigned a = hv[k].y < hv[k].w ? hv[k].y : hv[k].w;
            m = a < m ? a : m;
          }
          if (__builtin_amdgcn_ballot_w64(m != want) == 0ull) break;  // wave-uniform: the wave needs all H values anyway
          // every wave polls all H granules: 32 line requests per wave and round -- a few per cent of the L2 request
          // rate for the two clusters of a batch-1 call; with dozens of clusters the polling-wave kernel (one poller per
          // workgroup) is ahead again, see the version rule in ou_api.cpp.  Back off if a wait gets long.
          ++spins;
          if ((spins & 63u) == 0u) __builtin_amdgcn_s_sleep(4);
          // Safety net: a plain store carries no visibility deadline.  Under load -- a second process on the device
          // (tests/test_gpu_distributed.py: 1 run in 4 timed out), or dozens of clusters (OR16, B = 16: multi-second stalls)
          // -- a publish was seen to stay invisible to the other CUs for good.  Everybody ends up waiting then, the wave
          // whose store is missing too: after ~0.1 ms of waiting (256 poll rounds; a healthy wait is 2-4) every wave repeats
          // its last publish (tag epoch + step) as a system-scope write-through store.  Never taken in a healthy run.
          if (__builtin_expect((spins & 15u) != 15u, 1)) continue;
          // a wait that long is unusual (a healthy one takes 2-4 rounds): has somebody raised the cluster's mode flag?
          bool flagged = false;
          if (!sysmode && !(p.dbg & 1)) {
            u32x2 mflag;
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=v"(mflag) : "v"(xq + 2 * H + 63) : "memory");
            flagged = __builtin_amdgcn_readfirstlane(mflag.y) == epoch;
          }
          if (((spins & 255u) == 255u || flagged) && !(p.dbg & 1)) {
            if (!flagged && !sysmode)  // first trigger of this wave: leave a record of what the stale granule looks like
              gru_stale_probe(xq + (size_t)(step & 1) * H, WIDE ? 2 * lane : cg * 4, NC, WIDE ? 128 : 4 * LPU, want, p.err,
                              ((unsigned)cluster << 16) | ((unsigned)g << 8) | (unsigned)(tid >> 6), xcc, (unsigned)step,
                              WIDE ? 2 : 4);
            if (fin) {
              const unsigned long long gran = ((unsigned long long)want << 32) | (unsigned)__float_as_int(hprev);
              unsigned long long* dst = xq + (size_t)(step & 1) * H + unit;
              asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(gran) : "memory");
            }
            // ... and for the rest of this launch the wave publishes that way in the first place: when the effect shows
            // up it lasts (one GRU pass out of 574 in a profiled run needed a recovery on almost every step: 46 ms
            // instead of 0.28), whereas a system-scope publish costs about one more hop per step (0.39 ms per pass).
            // The cluster's flag granule (slot 63 of the rendezvous area, = this launch's epoch) makes every other wave
            // -- they are all waiting, and look at the flag every 16 rounds -- switch right away instead of after a
            // 256-round wait of its own (64 waves x 0.1 ms otherwise).  Status word 31 counts the triggers.
            if (!sysmode) {
              sysmode = true;
              fast_pub = false;
              if (lane == 0 && !flagged) {  // tell the rest of the cluster: flag granule = this launch's epoch
                const unsigned long long fl = ((unsigned long long)epoch << 32) | 1u;
                asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(xq + 2 * H + 63), "v"(fl) : "memory");
                atomicAdd(p.err + 31, 1u);
              }
            }
            // status word 20 counts the recoveries (one per wave and event)
            if (lane == 0) atomicAdd(p.err + 20, 1u);
          }
          // a long wait (~2 ms): leave this workgroup's position in its rendezvous slot for whoever reports a time-out
          if (spins == 4096u && tid == 0) gru_note_long_wait(xq + 2 * H + g, epoch, xcc, (unsigned)step);
          if (spins > GRU_SPIN_LIMIT) {
            if (lane == 0)
              gru_timeout_report(p.err, xq + 2 * H, NWG, epoch, ((unsigned)cluster << 16) | ((unsigned)g << 8) | (plain ? 1u : 0u),
                                 ((unsigned)step << 8) | (xcc & 0xFFu), m, 0u, want);
            // (which granule is stale: the first-event record of gru_stale_probe, written at the first recovery)
            step = T;
            break;
          }
        }
        if (step >= T) break;
        if (ts_on) c_rounds += spins + 1u;
      }
      if (ts_on) q1 = __builtin_readcyclecounter();
      // ---- matvec: (r, z) as packed row pairs against the granule's value half, n as scalar FMAs
      float hs[3];
      if constexpr (WIDE) {
        // 3 WU partial sums (unit u of the wave x gate) over this lane's HB columns
        constexpr int HU = WU / 2;
        f32x2 arz[WU];
        f32x2 an2[HU];  // the n sums of units u and u + HU as one packed pair: half the FMA instructions
#pragma unroll
        for (int u = 0; u < WU; u++) arz[u] = irz[u];
#pragma unroll
        for (int u = 0; u < HU; u++) an2[u] = f32x2{inn[u], inn[u + HU]};
#pragma unroll
        for (int k = 0; k < NC / 2; k++) {
          const float h0 = __uint_as_float(hv[k].x), h1 = __uint_as_float(hv[k].z);
#pragma unroll
          for (int u = 0; u < WU; u++) {
            arz[u] = __builtin_elementwise_fma(wrz[u * HB + 2 * k], f32x2{h0, h0}, arz[u]);
            arz[u] = __builtin_elementwise_fma(wrz[u * HB + 2 * k + 1], f32x2{h1, h1}, arz[u]);
          }
#pragma unroll
          for (int u = 0; u < HU; u++) {
            an2[u] = __builtin_elementwise_fma(f32x2{wn[u * HB + 2 * k], wn[(u + HU) * HB + 2 * k]}, f32x2{h0, h0}, an2[u]);
            an2[u] = __builtin_elementwise_fma(f32x2{wn[u * HB + 2 * k + 1], wn[(u + HU) * HB + 2 * k + 1]}, f32x2{h1, h1}, an2[u]);
          }
        }
        // fold 64 -> 32 lanes: units u and u + WU / 2 trade halves; lanes < 32 keep the lower units, lanes >= 32 the upper ones
        f32x2 rz01[HU];
        float n01[HU];
#pragma unroll
        for (int u = 0; u < HU; u++) {
          const auto sr = __builtin_amdgcn_permlane32_swap(__float_as_uint(arz[u].x), __float_as_uint(arz[u + HU].x), false, false);
          const auto sz = __builtin_amdgcn_permlane32_swap(__float_as_uint(arz[u].y), __float_as_uint(arz[u + HU].y), false, false);
          const auto sn = __builtin_amdgcn_permlane32_swap(__float_as_uint(an2[u].x), __float_as_uint(an2[u].y), false, false);
          rz01[u] = f32x2{__uint_as_float(sr[0]), __uint_as_float(sz[0])} + f32x2{__uint_as_float(sr[1]), __uint_as_float(sz[1])};
          n01[u] = __uint_as_float(sn[0]) + __uint_as_float(sn[1]);
        }
        if constexpr (WU == 4) {
          // fold 32 -> 16 lanes: the two remaining units trade rows; row r of the wave ends up with unit r
          const auto sr = __builtin_amdgcn_permlane16_swap(__float_as_uint(rz01[0].x), __float_as_uint(rz01[1].x), false, false);
          const auto sz = __builtin_amdgcn_permlane16_swap(__float_as_uint(rz01[0].y), __float_as_uint(rz01[1].y), false, false);
          const auto sn = __builtin_amdgcn_permlane16_swap(__float_as_uint(n01[0]), __float_as_uint(n01[1]), false, false);
          const f32x2 rz = f32x2{__uint_as_float(sr[0]), __uint_as_float(sz[0])} + f32x2{__uint_as_float(sr[1]), __uint_as_float(sz[1])};
          hs[0] = rz.x; hs[1] = rz.y;
          hs[2] = __uint_as_float(sn[0]) + __uint_as_float(sn[1]);
        } else {  // two units per wave: each half of the wave goes on as one 32-lane group (row sums + row_bcast:15 below)
          hs[0] = rz01[0].x; hs[1] = rz01[0].y; hs[2] = n01[0];
        }
      } else {
        f32x2 arz0 = irz[0], arz1 = {0.f, 0.f};
        float an0 = inn[0], an1 = 0.f;
#pragma unroll
        for (int k = 0; k < NC / 2; k++) {
          const float h0 = __uint_as_float(hv[k].x), h1 = __uint_as_float(hv[k].z);
          arz0 = __builtin_elementwise_fma(wrz[2 * k], f32x2{h0, h0}, arz0);
          arz1 = __builtin_elementwise_fma(wrz[2 * k + 1], f32x2{h1, h1}, arz1);
          an0 = fmaf(wn[2 * k], h0, an0);
          an1 = fmaf(wn[2 * k + 1], h1, an1);
        }
        hs[0] = arz0.x + arz1.x; hs[1] = arz0.y + arz1.y; hs[2] = an0 + an1;
      }
#pragma unroll
      for (int gt = 0; gt < 3; gt++) {
        hs[gt] = LPU == 8 ? row8_sum(hs[gt]) : row16_sum(hs[gt]);
        if (LPU == 32) {  // rows 1 / 3 += total of rows 0 / 2
          const int lo = __builtin_amdgcn_update_dpp(0, __float_as_int(hs[gt]), 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
          hs[gt] += __int_as_float(lo);
        }
      }

      if (fin) {
        // hs[0] = S (x_r + W_hr h), hs[1] = S (x_z + W_hz h), hs[2] = T (W_hn h + b_hn), xn = T x_n   (see kGateS / kGateT)
        const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(hs[0]));
        const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(hs[1]));
        const float n = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(r, hs[2], xn))), 1.0f);
        const float hnew = fmaf(hprev - n, z, n);
        hprev = hnew;
        {  // publish first: everybody is waiting on this
          const unsigned long long gran =
              ((unsigned long long)(epoch + (unsigned)step + 1u) << 32) | (unsigned)__float_as_int(hnew);
          unsigned long long* dst = xq + (size_t)((step + 1) & 1) * H + unit;
          // (one branch on the fast path: this store is on the critical path of every step.)  Plain store = the cluster
          // shares one XCD (proved by the rendezvous): the line stays in the L2 that every poller's sc1 load is served from.
          // Otherwise ONE agent-scope (sc1, write-through) 8-byte store per granule, the documented form of a data-tagged
          // hand-off on gfx950 -- it drops the line from the XCD's L2 (+0.11 ms per pass), see DESIGN.md 4.4.
          if (__builtin_expect(fast_pub && step != inject_step, 1)) {
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(gran) : "memory");
          } else if (step == inject_step) {
            // fault injection (tests): workgroup 1 "loses" its publishes of step 50 -- the safety net has to bring them back
          } else if (sysmode) {
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(gran) : "memory");
          } else {
            asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(gran) : "memory");
          }
        }
        p.out[orow + t] = has_res ? (hnew + rs) * p.res_scale : hnew;
      }
      if (ts_on) {
        const long long q2 = __builtin_readcyclecounter();
        c_poll += q1 - q0; c_comp += q2 - q1;
        if (p.poll_backoff == 10) {  // experiment: how long until this wave's two stores (publish, output) are acknowledged?
          __builtin_amdgcn_s_waitcnt(0x0F70);
          c_ack += __builtin_readcyclecounter() - q2;
        } else if (p.poll_backoff >= 11 && p.poll_backoff <= 14) {
          // experiment: latency of ONE 8-byte load per lane once this wave's stores are acknowledged --
          // 11: sc1, quiet lines (the rendezvous slots), 12: sc1, the buffer that was gathered in this step (nobody writes
          // it now), 13: the same without sc1, 14: sc1, the buffer everybody is publishing into right now
          __builtin_amdgcn_s_waitcnt(0x0F70);
          const unsigned long long* a = p.poll_backoff == 11 ? xq + 2 * H + (lane & 31)
                                        : xq + (size_t)((step + (p.poll_backoff == 14 ? 1 : 0)) & 1) * H + (tid >> 6) * (H / 4) + lane % (H / 4);
          u32x2 d;
          const long long q3 = __builtin_readcyclecounter();
          if (p.poll_backoff == 13) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(a) : "memory");
          else asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(a) : "memory");
          c_ack += __builtin_readcyclecounter() - q3;
        }
      }
    }
    if (ts_on && lane == 0) {
      long long* o = p.tstamps + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 8;
      // 10 ns ticks: kernel entry -> first step (weights, rendezvous, first chunks), the T steps
      o[0] = c_comp; o[1] = c_poll; o[2] = c_rounds; o[3] = T; o[4] = r_loop - r_start;
      o[5] = (long long)__builtin_amdgcn_s_memrealtime() - r_loop; o[6] = r_start; o[7] = c_ack;
    }
  }
  // ---- epoch hand-over: the last block to finish advances the epoch for the next launch on this exchange area
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned done = atomicAdd(p.epoch + 1, 1u);
    if (done == gridDim.x - 1) {
      p.epoch[1] = 0u;
      unsigned next = epoch + (unsigned)T;  // stored counter = last tag used
      if (next >= GRU_EPOCH_WRAP) {  // tags must stay monotonic: clear the area and restart (every block is done)
        // the whole area of this GRU layer, not just this launch's clusters: the sub-launches of a chunked batch share
        // it, and a stale high tag left behind a smaller remainder launch would outlive the restart
        const size_t n = p.xchg_granules ? p.xchg_granules : (size_t)nclusters * CSTRIDE;
        for (size_t i = 0; i < n; i++) p.xchg_base[i] = 0ull;
        next = 0u;
      }
      __threadfence();
      __hip_atomic_store(p.epoch, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int HB>
static void (*gru_ring_entry(int upw, int wide))(GruArgs, int) {
  if constexpr (HB <= 4) {
    if (upw == 32) return gru_ring_kernel<HB, 32>;
  }
  if constexpr (HB % 2 == 0) {  // 8 units per workgroup: half the matvec per wave, twice the workgroups
    if (upw == 8 && wide) return gru_ring_kernel<HB, 8, true>;
  }
  if constexpr (HB >= 2 && HB <= 4) {
    if (upw == 8) return gru_ring_kernel<HB, 8>;
  }
  if constexpr (HB % 2 == 0) {
    if (wide) return gru_ring_kernel<HB, 16, true>;
  }
  return gru_ring_kernel<HB, 16>;
}
// Workgroups of the ring kernel that can be resident per CU (every member of a cluster spins on the others: the whole
// grid has to be on the machine at once).  The occupancy query can be one block high where SGPRs are the limit (guide:
// admitted = min(API, 8, 800 / (ceil(sgpr / 16) * 16 + 16)): API 8 -> 7 at 81-96 SGPRs, 7 -> 6 at 97-112), which only
// concerns answers >= 7: one block of margin is taken off those.  At most TWO per CU are relied upon (two workgroups = two
// waves per SIMD, each waiting on its gather most of the time).
template <int HB>
static int gru_ring_resident_per_cu(int upw, int wide) {
  static int cache[5] = {0, 0, 0, 0, 0};
  const int slot = upw == 32 ? 2 : (upw == 8 ? (wide ? 4 : 0) : (wide ? 3 : 1));
  if (cache[slot] == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(gru_ring_entry<HB>(upw, wide)), 256,
                                                     0) != hipSuccess)
      nb = 1;
    if (nb >= 7) nb -= 1;
    cache[slot] = nb >= 2 ? 2 : 1;
  }
  return cache[slot];
}
// OU_GRU_UPW (force_upw): 0 auto | 16 / 8 units per workgroup, wide layout | 17 / 9 the same in the default (round-2)
// layout | 32 units, default layout
static int gru_ring_wide(int H, int force_upw) {
  return ((H / 64) % 2 == 0 && force_upw != 17 && force_upw != 32 && force_upw != 9) ? 1 : 0;
}
// hidden units per workgroup.  16 by default; 8 (wide layout: two units per wave -- half the matvec and one fold level less
// on the critical path of every step, 252 -> 240 us per 401-frame pass) while every workgroup of the launch still gets a CU
// of its own: measured 7.59 -> 7.43 ms per enhance at B = 1, 16.80 -> 16.71 at B = 4, but 29.0 -> 29.5 at B = 8 (two
// workgroups per CU).  B = 0: the choice for the smallest batch.  `lanes`: enhance calls in flight side by side in this
// process (each with its own launches), `share`: GRU launches that may be resident on one XCD at the same time.
static int gru_ring_per_cu(int H, int upw, int wide) {
  switch (H / 64) {
    case 1: return gru_ring_resident_per_cu<1>(upw, wide);
    case 2: return gru_ring_resident_per_cu<2>(upw, wide);
    case 4: return gru_ring_resident_per_cu<4>(upw, wide);
    case 6: return gru_ring_resident_per_cu<6>(upw, wide);
    default: return 0;
  }
}
static int gru_ring_upw(int H, int force_upw, int B, int num_cu, int lanes, int share) {
  if (force_upw == 32 && H <= 256) return 32;
  if (force_upw == 9 && H >= 128 && H <= 256) return 8;
  if (force_upw == 8 && (H / 64) % 2 == 0 && H <= 384) return 8;
  (void)lanes;
  if (force_upw == 0 && (H / 64) % 2 == 0 && H <= 256 && 2 * (B > 0 ? B : 1) * (H / 8) <= num_cu) {
    // ... and the clusters that may meet on one XCD must all fit there (every member of a cluster spins on the others).  The
    // lanes of a process do NOT enter the one-CU-per-workgroup preference above: a lane keeps the split of the single call as
    // long as everything is resident (two workgroups per CU at worst), so that its results stay bit-identical to that call's.
    const int per_cu = gru_ring_per_cu(H, 8, gru_ring_wide(H, force_upw));
    if (share <= 1 || (num_cu / 8) * per_cu / (H / 8) >= share) return 8;
  }
  return 16;
}
// utterances one ring-kernel launch may carry: whole groups of 8 clusters (one per XCD), two clusters per utterance, the
// whole grid resident -- with 1 / share of every XCD's capacity when `share` GRU launches may run side by side there (the
// conditioner's layer beside the first score pass'; the launches of other lanes)
int gru_ring_batch_cap(int H, int num_cu, int share, int force_upw, int B, int lanes) {
  if (H % 64) return 0;
  if (share < 1) share = 1;
  const int upw = gru_ring_upw(H, force_upw, B, num_cu, lanes, share), nwg = H / upw;
  const int wide = gru_ring_wide(H, force_upw);
  const int per_cu = gru_ring_per_cu(H, upw, wide);
  if (per_cu <= 0) return 0;
  const int cpx = (num_cu / 8) * per_cu / nwg;  // clusters one XCD can hold
  return (cpx / share) * 8 / 2;
}
template <int HB>
static hipError_t launch_gru_ring(const GruArgs& c, int upw, int nclusters, hipStream_t st) {
  constexpr int H = 64 * HB;
  const int nwg = H / upw;
  dim3 grid(8 * nwg * ((nclusters + 7) / 8));
  hipLaunchKernelGGL(gru_ring_entry<HB>(upw, gru_ring_wide(H, c.force_upw)), grid, dim3(256), 0, st, c, nclusters);
  return hipGetLastError();
}

#ifdef OU_EXPERIMENTS
template <int HB>
static hipError_t launch_gru_variant(const GruArgs& c, int upw, int nclusters, hipStream_t st) {
  constexpr int H = 64 * HB;
  const int nwg = H / upw;
  dim3 grid(8 * nwg * ((nclusters + 7) / 8));
  if (upw == 64) hipLaunchKernelGGL((gru_cluster_kernel<HB, 64, 512>), grid, dim3(512), 0, st, c, nclusters);
  else if (upw == 32) hipLaunchKernelGGL((gru_cluster_kernel<HB, 32, 512>), grid, dim3(512), 0, st, c, nclusters);
  else hipLaunchKernelGGL((gru_cluster_kernel<HB, 16, 256>), grid, dim3(256), 0, st, c, nclusters);
  return hipGetLastError();
}

#endif  // OU_EXPERIMENTS

// Measurement only (GruArgs.prof): the pass is timed by two one-thread kernels on the same stream, right before and right
// behind it -- the recurrence kernel itself carries no stamps (its step loop has no SGPR to spare: `make check` fails the
// build on a spill there).  What is read back is the pass plus the two dispatch gaps around it (~2 us of ~240).
__global__ void gru_stamp_kernel(unsigned long long* slot, int end) {
  const unsigned long long t = __builtin_amdgcn_s_memrealtime();
  if (end) atomicMin(slot + 16, ~t);
  else atomicMin(slot, t);
}

hipError_t launch_gru(const GruArgs& a, int num_cu, hipStream_t st) {
  if (a.prof) hipLaunchKernelGGL(gru_stamp_kernel, dim3(1), dim3(1), 0, st, a.prof, 0);
  struct EndStamp {
    const GruArgs& a; hipStream_t st;
    ~EndStamp() { if (a.prof) hipLaunchKernelGGL(gru_stamp_kernel, dim3(1), dim3(1), 0, st, a.prof, 1); }
  } end_stamp{a, st};
  if (a.H % 64) return hipErrorInvalidValue;
  const int HB = a.H / 64;
  // Every workgroup of a cluster has to be resident at once; clusters are dealt to XCDs in groups of 8 and a launch
  // is sized to at most half the CUs (the conditioner's and the score net's GRUs may overlap).  Take the finest split
  // -- least work per time step; measured 12.1 / 12.6 / 13.1 ms per PP16 enhance for 16 / 32 / 64 units per
  // workgroup, the same order at B = 2, 4, 8 -- that still runs the whole batch in ONE launch.
  auto batch_cap = [&](int u) { return (num_cu / (8 * (a.H / u))) * 8 / 2; };
  int upw = 64;
  if (a.force_upw) upw = a.force_upw;
  else if (a.H % 16 == 0 && batch_cap(16) >= a.B) upw = 16;
  else if (a.H % 32 == 0 && batch_cap(32) >= a.B) upw = 32;
  // the ring kernel runs 256-thread workgroups of 16 units (32 on request, H <= 256)
  if (a.version == 2) upw = gru_ring_upw(a.H, a.force_upw, a.B, num_cu, a.lanes, a.share);
  const int nwg = a.H / upw;
  (void)nwg;  // (used by the EXPERIMENTS-only polling-wave path)
  int bmax = batch_cap(upw);
  // residency of the ring kernel: every member of a cluster spins on the others, so a launch is sized to what can be on
  // the machine at once (half of it when a second GRU layer may run beside this one: conditioner / first score pass);
  // a batch that does not fit is split into sub-launches, never enqueued oversized
  if (a.version == 2) bmax = gru_ring_batch_cap(a.H, num_cu, a.share, a.force_upw, a.B, a.lanes);
  if (a.force_bmax > 0 && a.force_bmax < bmax) bmax = a.force_bmax;
  if (bmax < 1) return hipErrorInvalidConfiguration;
  for (int b0 = 0; b0 < a.B; b0 += bmax) {
    GruArgs c = a;
    c.B = (a.B - b0 < bmax) ? a.B - b0 : bmax;
    c.gx = a.gx + (size_t)b0 * 6 * a.H * a.T;
    c.out = a.out + (size_t)b0 * 2 * a.H * a.T;
    if (a.res) c.res = a.res + (size_t)b0 * 2 * a.H * a.T;
    if (a.version == 2) {
      if (!a.epoch) return hipErrorInvalidValue;
      c.xchg_base = a.xchg;
      c.xchg_granules = gru_granules(a.B, a.H);
      hipError_t e;
      switch (HB) {
        case 1: e = launch_gru_ring<1>(c, upw, 2 * c.B, st); break;
        case 2: e = launch_gru_ring<2>(c, upw, 2 * c.B, st); break;
        case 4: e = launch_gru_ring<4>(c, upw, 2 * c.B, st); break;
        case 6: e = launch_gru_ring<6>(c, upw, 2 * c.B, st); break;
        default: return hipErrorInvalidConfiguration;
      }
      if (e != hipSuccess) return e;
      continue;
    }
#ifndef OU_EXPERIMENTS
    return hipErrorInvalidConfiguration;  // OU_GRU_V=1 needs a library built with `make EXPERIMENTS=1`
#else
    if (nwg > 1) {
      hipError_t e = hipMemsetAsync(c.xchg, 0, (size_t)c.B * 4 * a.H * sizeof(unsigned long long), st);
      if (e != hipSuccess) return e;
    }
    hipError_t e;
    switch (HB) {
      case 1: e = launch_gru_variant<1>(c, upw, 2 * c.B, st); break;
      case 2: e = launch_gru_variant<2>(c, upw, 2 * c.B, st); break;
      case 4: e = launch_gru_variant<4>(c, upw, 2 * c.B, st); break;
      case 6: e = launch_gru_variant<6>(c, upw, 2 * c.B, st); break;
      default: return hipErrorInvalidConfiguration;
    }
    if (e != hipSuccess) return e;
#endif  // OU_EXPERIMENTS
  }
  return hipSuccess;
}


}  // namespace ou
