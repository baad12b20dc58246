// GPU test program (built and run by tests/test_gpu_se_set.py): the wave-wide DPP scans of abm_device.hpp
// (wave_incl_sum / wave_excl_sum / wave_incl_max) and the seed passes' segment location (locate128) against
// serial host code, on random inputs.  Prints "OK <cases>" or the first mismatch.
#include "../../abismal_amd/csrc/abm_kernels.hip"
#include <cstdio>
#include <vector>
using namespace abm;

__global__ __launch_bounds__(64) void scans(const u32 *in, u32 *sum_out, u32 *max_out, u32 *tot_out) {
  const u32 x = in[blockIdx.x * 64 + threadIdx.x];
  u32 total;
  sum_out[blockIdx.x * 64 + threadIdx.x] = wave_excl_sum(x, total);
  max_out[blockIdx.x * 64 + threadIdx.x] = wave_incl_max(x);
  if (threadIdx.x == 0) tot_out[blockIdx.x] = total;
}

// one flattened block per workgroup: na / nb per lane; every candidate's (segment, entry index) through locate128
__global__ __launch_bounds__(64) void locate(const u32 *na, const u32 *nb, const u32 *lo2, const u32 *lo3, u32 *seg_out,
                                             u32 *entry_out, u32 cap, u32 rounds) {
  __shared__ u32 smark[128], sdelta[128];
  WaveLds lds{};
  lds.smark = smark; lds.sdelta = sdelta;
  const int lane = threadIdx.x;
  smark[lane] = 0; smark[64 + lane] = 0;
  u32 epoch = 0;
  for (u32 rep = 0; rep < rounds; ++rep) {  // (repeated: stale marks of earlier calls must lose)
    const u32 b = (blockIdx.x * rounds + rep);
    Segs sg;
    sg.na = na[b * 64 + lane]; sg.nb = nb[b * 64 + lane]; sg.lo2 = lo2[b * 64 + lane]; sg.lo3 = lo3[b * 64 + lane];
    u32 total;
    sg.start_a = wave_excl_sum(sg.na + sg.nb, total);
    publish_segs(lds, sg);
    u32 carry = 0;
    for (u32 c0 = 0; c0 < total; c0 += 128) {
      bool va, vb;
      u32 sa, sb, ea, eb;
      locate128(lds, epoch, sg, c0, total, carry, va, vb, sa, sb, ea, eb);
      if (va && c0 + lane < cap) { seg_out[b * cap + c0 + lane] = sa; entry_out[b * cap + c0 + lane] = ea; }
      if (vb && c0 + 64 + lane < cap) { seg_out[b * cap + c0 + 64 + lane] = sb; entry_out[b * cap + c0 + 64 + lane] = eb; }
    }
  }
}


// wavefront_rows<true> against wavefront<true>: one random job per block -- read and window words, length, band --
// run through both; every cell the anti-diagonal run writes must hold the same arrow byte in the row-by-row run's
// table, and the per-lane best value and row must agree.
__global__ __launch_bounds__(64) void rows_vs_antidiagonal(const u64 *qwords, const u64 *gwords, const int *Ls, const int *bws,
                                                           const int *t0s, u32 W, u32 GW, u32 *bad) {
  extern __shared__ unsigned char smem[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int L = Ls[b], bw = bws[b];
  const size_t tb_bytes = static_cast<size_t>(L + bw) * bw;
  u64 *q = reinterpret_cast<u64 *>(smem);
  u64 *g = q + W;
  u8 *tb_a = reinterpret_cast<u8 *>(g + GW);
  u8 *tb_b = tb_a + ((tb_bytes + 15) & ~size_t(15));
  for (u32 k = lane; k < W; k += 64) q[k] = qwords[b * W + k];
  for (u32 k = lane; k < GW; k += 64) g[k] = gwords[b * GW + k];
  for (size_t k = lane; k < tb_bytes; k += 64) { tb_a[k] = 0xEE; tb_b[k] = 0xEE; }
  __syncthreads();
  WaveLds lds{};
  lds.qpk = q; lds.gwin = g; lds.W = W; lds.GW = GW;
  AlnJob job = {0, 0, 0, 0, 0};
  if (lane < bw) { job.bw = bw; job.jl = lane; job.t0nib = t0s[b]; }
  int va, ra, vb, rb;
  lds.tb = tb_a;
  wavefront<true>(lds, job, L, bw, bw, va, ra);
  __syncthreads();
  lds.tb = tb_b;
  wavefront_rows<true>(lds, job, L, bw, vb, rb);
  __syncthreads();
  u32 wrong = 0;
  if (lane < bw && (va != vb || ra != rb)) ++wrong;
  for (size_t k = lane; k < tb_bytes; k += 64)
    if (tb_a[k] != 0xEE && tb_a[k] != tb_b[k]) ++wrong;
  if (wrong) atomicAdd(&bad[b], wrong);
}

// score_round_quad (four jobs per slot, packed 16-bit cells) against score_round (two per slot): random job lists --
// positions on a random genome, Hamming counts that give every band width, all four read encodings -- scored by
// both, job by job.
__global__ __launch_bounds__(64) void quad_vs_pair(const u64 *genome, const u64 *qwords, const u32 *jpos_in, const u32 *jdf_in,
                                                   const int *n_jobs_in, const int *Ls, u32 W, u32 GW, int md, int *score_pair,
                                                   int *score_quad) {
  extern __shared__ unsigned char smem[];
  const int b = blockIdx.x, lane = threadIdx.x;
  u64 *q = reinterpret_cast<u64 *>(smem);            // [4 * W]
  u64 *gwin = q + 4 * W;                              // [kMaxJobs * GW]
  u32 *jpos = reinterpret_cast<u32 *>(gwin + kMaxJobs * GW);
  u32 *jdf = jpos + 64;
  int *lbest = reinterpret_cast<int *>(jdf + 64);
  const int n = n_jobs_in[b], L = Ls[b];
  for (u32 k = lane; k < 4 * W; k += 64) q[k] = qwords[static_cast<size_t>(b) * 4 * W + k];
  if (lane < n) { jpos[lane] = jpos_in[b * 64 + lane]; jdf[lane] = jdf_in[b * 64 + lane]; }
  __syncthreads();
  DevIndex ix{};
  ix.genome = genome;
  WaveLds lds{};
  lds.qpk = q; lds.gwin = gwin; lds.jpos = jpos; lds.jdf = jdf; lds.lbest = lbest; lds.W = W; lds.GW = GW; lds.max_jobs = kMaxJobs;
  for (int s = 0; s < n;) {
    const int first = s;
    s = score_round<false>(ix, lds, first, n, L, md, 0);
    if (lane < s - first) score_pair[b * 64 + first + lane] = lbest[lane];
    __syncthreads();
  }
  for (int s = 0; s < n;) {
    const int first = s;
    s = score_round_quad(ix, lds, first, n, L, md, 0);
    if (lane < s - first) score_quad[b * 64 + first + lane] = lbest[lane];
    __syncthreads();
  }
}

static u32 rnd(u32 &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

int main() {
  const int B = 512;
  std::vector<u32> in(B * 64), es(B * 64), em(B * 64), et(B);
  u32 seed = 12345;
  for (int b = 0; b < B; ++b) {
    u32 run = 0, mx = 0;
    for (int l = 0; l < 64; ++l) {
      const u32 r = rnd(seed);
      const u32 v = (b % 4 == 0) ? (r % 3 == 0 ? r >> 8 : 0) : (b % 4 == 1 ? r % 5 : (b % 4 == 2 ? r : (l == (b % 64) ? 7u : 0u)));
      in[b * 64 + l] = v;
      es[b * 64 + l] = run; run += v;
      mx = v > mx ? v : mx; em[b * 64 + l] = mx;
    }
    et[b] = run;
  }
  u32 *d_in, *d_s, *d_m, *d_t;
  hipMalloc(&d_in, in.size() * 4); hipMalloc(&d_s, in.size() * 4); hipMalloc(&d_m, in.size() * 4); hipMalloc(&d_t, B * 4);
  hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(scans, dim3(B), dim3(64), 0, 0, d_in, d_s, d_m, d_t);
  std::vector<u32> gs(B * 64), gm(B * 64), gt(B);
  hipMemcpy(gs.data(), d_s, gs.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(gm.data(), d_m, gm.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(gt.data(), d_t, gt.size() * 4, hipMemcpyDeviceToHost);
  for (int k = 0; k < B * 64; ++k)
    if (gs[k] != es[k] || gm[k] != em[k]) { std::printf("scan mismatch block %d lane %d: sum %u/%u max %u/%u\n", k / 64, k % 64, gs[k], es[k], gm[k], em[k]); return 1; }
  for (int b = 0; b < B; ++b) if (gt[b] != et[b]) { std::printf("total mismatch block %d\n", b); return 1; }

  // locate128
  const u32 rounds = 8, NB = 64 * rounds, cap = 8192;
  std::vector<u32> na(NB * 64), nb(NB * 64), l2(NB * 64), l3(NB * 64), eseg(static_cast<size_t>(NB) * cap, 0xFFFFFFFFu), eent(static_cast<size_t>(NB) * cap, 0xFFFFFFFFu);
  for (u32 b = 0; b < NB; ++b) {
    u32 c = 0;
    for (int l = 0; l < 64; ++l) {
      const u32 r = rnd(seed), shape = b % 5;
      u32 a = 0, bb = 0;
      if (shape == 0) { a = r % 4 == 0 ? (r >> 8) % 101 : 0; bb = (r >> 4) % 7 == 0 ? (r >> 16) % 101 : 0; }
      else if (shape == 1) { a = (r >> 3) % 101; bb = (r >> 12) % 101; }
      else if (shape == 2) { a = l == 5 ? 100 : 0; bb = l == 60 ? 3 : 0; }
      else if (shape == 3) { a = (r & 1); bb = (r >> 1) & 1; }
      else { a = l < 3 ? 1 : 0; bb = 0; }
      if (c + a + bb > cap) { a = 0; bb = 0; }
      na[b * 64 + l] = a; nb[b * 64 + l] = bb;
      l2[b * 64 + l] = rnd(seed) >> 4; l3[b * 64 + l] = rnd(seed) >> 4;
      for (u32 k = 0; k < a; ++k) { eseg[static_cast<size_t>(b) * cap + c] = 2 * l; eent[static_cast<size_t>(b) * cap + c] = l2[b * 64 + l] + k; ++c; }
      for (u32 k = 0; k < bb; ++k) { eseg[static_cast<size_t>(b) * cap + c] = 2 * l + 1; eent[static_cast<size_t>(b) * cap + c] = l3[b * 64 + l] + k; ++c; }
    }
  }
  u32 *d_na, *d_nb, *d_l2, *d_l3, *d_seg, *d_ent;
  hipMalloc(&d_na, na.size() * 4); hipMalloc(&d_nb, na.size() * 4); hipMalloc(&d_l2, na.size() * 4); hipMalloc(&d_l3, na.size() * 4);
  hipMalloc(&d_seg, eseg.size() * 4); hipMalloc(&d_ent, eseg.size() * 4);
  hipMemcpy(d_na, na.data(), na.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_nb, nb.data(), na.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_l2, l2.data(), na.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_l3, l3.data(), na.size() * 4, hipMemcpyHostToDevice);
  hipMemset(d_seg, 0xFF, eseg.size() * 4); hipMemset(d_ent, 0xFF, eseg.size() * 4);
  hipLaunchKernelGGL(locate, dim3(NB / rounds), dim3(64), 0, 0, d_na, d_nb, d_l2, d_l3, d_seg, d_ent, cap, rounds);
  std::vector<u32> gseg(eseg.size()), gent(eseg.size());
  hipMemcpy(gseg.data(), d_seg, gseg.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(gent.data(), d_ent, gent.size() * 4, hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 1; }
  for (size_t k = 0; k < eseg.size(); ++k)
    if (gseg[k] != eseg[k] || gent[k] != eent[k]) {
      std::printf("locate mismatch block %zu candidate %zu: segment %u/%u entry %u/%u\n", k / cap, k % cap, gseg[k], eseg[k], gent[k], eent[k]);
      return 1;
    }
  // wavefront_rows vs wavefront
  const int NJ = 3000;
  const u32 QW = 10, QGW = 16;  // up to 160 read bases, windows of 256 bases
  std::vector<u64> qv(static_cast<size_t>(NJ) * QW), gv(static_cast<size_t>(NJ) * QGW);
  std::vector<int> jl(NJ), jb(NJ), jt(NJ);
  const u32 enc[4] = {1u, 2u, 4u, 10u};  // T-rich read letters (A, C, G, T->T|C... as bisulfite nibbles)
  for (int j = 0; j < NJ; ++j) {
    const int L = 40 + static_cast<int>(rnd(seed) % 121);
    const int bw = 2 * static_cast<int>(rnd(seed) % (j % 3 == 0 ? 31 : 6)) + 1;
    jl[j] = L; jb[j] = bw; jt[j] = static_cast<int>(rnd(seed) % 16);
    // the window: random genome letters (one-hot nibbles); the read: the window's diagonal with mutations and an indel
    std::vector<u32> gl(QGW * 16), ql(QW * 16, 0u);
    for (auto &x : gl) x = 1u << (rnd(seed) % 4);
    int shift = jt[j] + (bw - 1) / 2, at = 0;
    const int indel_at = static_cast<int>(rnd(seed) % static_cast<u32>(L)), indel = static_cast<int>(rnd(seed) % 5) - 2;
    for (int k = 0; k < L; ++k) {
      if (k == indel_at) at += indel;
      const int gi = shift + k + at;
      u32 letter = (gi >= 0 && gi < static_cast<int>(gl.size())) ? gl[gi] : 1u;
      if (rnd(seed) % 12 == 0) letter = 1u << (rnd(seed) % 4);
      ql[k] = letter == 8u ? 10u : letter;  // (a read T matches genome T and C)
    }
    for (u32 w = 0; w < QW; ++w) { u64 x = 0; for (int k = 0; k < 16; ++k) x |= static_cast<u64>(ql[w * 16 + k]) << (4 * k); qv[static_cast<size_t>(j) * QW + w] = x; }
    for (u32 w = 0; w < QGW; ++w) { u64 x = 0; for (int k = 0; k < 16; ++k) x |= static_cast<u64>(gl[w * 16 + k]) << (4 * k); gv[static_cast<size_t>(j) * QGW + w] = x; }
  }
  (void)enc;
  u64 *d_q, *d_g;
  int *d_L, *d_bw, *d_t0;
  u32 *d_bad;
  hipMalloc(&d_q, qv.size() * 8); hipMalloc(&d_g, gv.size() * 8); hipMalloc(&d_L, NJ * 4); hipMalloc(&d_bw, NJ * 4); hipMalloc(&d_t0, NJ * 4); hipMalloc(&d_bad, NJ * 4);
  hipMemcpy(d_q, qv.data(), qv.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_g, gv.data(), gv.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_L, jl.data(), NJ * 4, hipMemcpyHostToDevice); hipMemcpy(d_bw, jb.data(), NJ * 4, hipMemcpyHostToDevice); hipMemcpy(d_t0, jt.data(), NJ * 4, hipMemcpyHostToDevice);
  hipMemset(d_bad, 0, NJ * 4);
  const size_t smem = (QW + QGW) * 8 + 2 * (((160 + 61) * 61 + 15) & ~15);
  hipLaunchKernelGGL(rows_vs_antidiagonal, dim3(NJ), dim3(64), smem, 0, d_q, d_g, d_L, d_bw, d_t0, QW, QGW, d_bad);
  std::vector<u32> gbad(NJ);
  hipMemcpy(gbad.data(), d_bad, NJ * 4, hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) { std::printf("rows kernel failed\n"); return 1; }
  for (int j = 0; j < NJ; ++j)
    if (gbad[j]) { std::printf("rows mismatch job %d (L %d band %d t0 %d): %u cells/lanes differ\n", j, jl[j], jb[j], jt[j], gbad[j]); return 1; }
  // score_round_quad vs score_round
  const int NQ = 2000, md = 15;
  const u32 SW = 10, SGW = 14;   // reads of up to 160 bases; windows of 150 + 61 + 15 bases
  const size_t GWORDS = 1u << 16;  // a genome of 1 M bases (one-hot nibbles)
  std::vector<u64> gen(GWORDS), sq(static_cast<size_t>(NQ) * 4 * SW);
  for (auto &x : gen) { u64 v = 0; for (int k = 0; k < 16; ++k) v |= static_cast<u64>(1u << (rnd(seed) % 4)) << (4 * k); x = v; }
  std::vector<u32> qpos(NQ * 64, 0), qdf(NQ * 64, 0);
  std::vector<int> qn(NQ), qL(NQ);
  auto gn = [&](u64 k) { return static_cast<u32>(gen[k >> 4] >> ((k & 15) << 2)) & 15u; };
  for (int b = 0; b < NQ; ++b) {
    const int L = 60 + static_cast<int>(rnd(seed) % 91), n = 1 + static_cast<int>(rnd(seed) % 50);
    qn[b] = n; qL[b] = L;
    // four encodings of "the read": each a mutated copy of a genome stretch (so that scores are far from trivial)
    const u64 origin = 1000 + rnd(seed) % (GWORDS * 16 - 4000);
    for (int e = 0; e < 4; ++e) {
      std::vector<u32> ql(SW * 16, 0u);
      int drift = 0;
      for (int k = 0; k < L; ++k) {
        if (rnd(seed) % 40 == 0) drift += static_cast<int>(rnd(seed) % 3) - 1;
        u32 letter = gn(origin + k + drift);
        if (rnd(seed) % 10 == 0) letter = 1u << (rnd(seed) % 4);
        ql[k] = (e & 1) ? (letter == 1u ? 5u : letter) : (letter == 8u ? 10u : letter);
      }
      for (u32 w = 0; w < SW; ++w) { u64 x = 0; for (int k = 0; k < 16; ++k) x |= static_cast<u64>(ql[w * 16 + k]) << (4 * k); sq[(static_cast<size_t>(b) * 4 + e) * SW + w] = x; }
    }
    for (int j = 0; j < n; ++j) {
      const int shape = b % 4;
      const u32 d = shape == 0 ? rnd(seed) % 40 : (shape == 1 ? 1 + rnd(seed) % 3 : (shape == 2 ? 15 + rnd(seed) % 20 : rnd(seed) % 16));
      qpos[b * 64 + j] = static_cast<u32>(origin + static_cast<int>(rnd(seed) % 9) - 4);
      qdf[b * 64 + j] = (d << 16) | ((rnd(seed) & 1) ? kFlagRC : 0u) | ((rnd(seed) & 1) ? kFlagARich : 0u);
    }
  }
  u64 *d_gen, *d_sq;
  u32 *d_qpos, *d_qdf;
  int *d_qn, *d_qL, *d_sp, *d_sqd;
  hipMalloc(&d_gen, gen.size() * 8); hipMalloc(&d_sq, sq.size() * 8); hipMalloc(&d_qpos, qpos.size() * 4); hipMalloc(&d_qdf, qdf.size() * 4);
  hipMalloc(&d_qn, NQ * 4); hipMalloc(&d_qL, NQ * 4); hipMalloc(&d_sp, NQ * 64 * 4); hipMalloc(&d_sqd, NQ * 64 * 4);
  hipMemcpy(d_gen, gen.data(), gen.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_sq, sq.data(), sq.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_qpos, qpos.data(), qpos.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_qdf, qdf.data(), qdf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_qn, qn.data(), NQ * 4, hipMemcpyHostToDevice); hipMemcpy(d_qL, qL.data(), NQ * 4, hipMemcpyHostToDevice);
  hipMemset(d_sp, 0xFF, NQ * 64 * 4); hipMemset(d_sqd, 0xFE, NQ * 64 * 4);
  const size_t smem2 = (4 * SW + kMaxJobs * SGW) * 8 + 3 * 64 * 4;
  hipLaunchKernelGGL(quad_vs_pair, dim3(NQ), dim3(64), smem2, 0, d_gen, d_sq, d_qpos, d_qdf, d_qn, d_qL, SW, SGW, md, d_sp, d_sqd);
  std::vector<int> sp(NQ * 64), sqd(NQ * 64);
  hipMemcpy(sp.data(), d_sp, sp.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(sqd.data(), d_sqd, sqd.size() * 4, hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) { std::printf("quad kernel failed\n"); return 1; }
  long long jobs_checked = 0, positive = 0;
  for (int b = 0; b < NQ; ++b)
    for (int j = 0; j < qn[b]; ++j) {
      if (sp[b * 64 + j] != sqd[b * 64 + j]) {
        std::printf("quad mismatch list %d job %d of %d (L %d, diffs %u): pair %d quad %d\n", b, j, qn[b], qL[b], qdf[b * 64 + j] >> 16, sp[b * 64 + j], sqd[b * 64 + j]);
        return 1;
      }
      ++jobs_checked; positive += sp[b * 64 + j] > 20;
    }
  if (positive * 4 < jobs_checked) { std::printf("quad check degenerate: %lld of %lld scores above 20\n", positive, jobs_checked); return 1; }
  std::printf("OK %d scan blocks, %u flattened blocks, %d traceback tables, %lld scoring jobs\n", B, NB, NJ, jobs_checked);
  return 0;
}
