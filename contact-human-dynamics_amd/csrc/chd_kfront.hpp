// chd_kfront.hpp -- frontal L D L^T of the bordered band with the front in the ACCUMULATOR REGISTERS of the workgroup.
// Included by chd_kernels.hpp inside namespace chd (after kfactor_rl / kfactor_ll; chd_config.factorisation = 2).
//
// Why: the right-looking factorisation read-modify-writes a panel's trailing window (100-200 active rows) in HBM / L2 every 32
// columns, the left-looking one re-reads the finished factor -- per panel a chain of dependent memory round trips for a few
// matrix-core tiles, and 7-9 MB of traffic per factorisation for a 1.2 MB factor.  The window fits neither LDS (as a
// dense triangle: 111-173 KB beside the panel) nor -- as a window that slides with the row index -- the register file's static
// indexing.  What does fit is a front addressed by SLOT: every row that is coupled to an eliminated column but not yet eliminated
// itself (the "active rows" of the other two versions, <= 208 at 16-column panels) holds one of C = 208 slots for as long as it is
// in the front; the symmetric C x C front matrix lives as 91 lower-triangular 16 x 16 fp64 tiles in the MFMA accumulator registers
// of the eight wavefronts (12 tiles = 96 VGPRs per wavefront, owned statically: register indices are compile-time constants, the
// slots are data).  Per 16-column panel:
//   extract   the pivot rows' columns of the front -> LDS (PT[j][slot]), by the lanes that hold them
//   diagonal  the 16 x 16 pivot block, one wavefront (while another builds the next panel's list of entering rows and gives them
//             slots)
//   rows      every other row of the front is solved against it in LDS and stored to the factor storage (written once)
//   update    front -= L_panel D L_panel^T on the matrix cores, operands from LDS, accumulators stay where they are; the
//             entries of the rows that enter at the next panel are fetched from K0 (read-only, requested before the MFMA loop)
// so nothing on the panel's critical path waits for HBM, K0 is read once and the factor written once.  The border rows are
// front rows that never leave; what remains in the front after the last band panel is the border's Schur complement, which
// goes to LDS for the dense L D L^T as before.
// Falls back (returns false, the caller runs kfactor_rl) when the front needs more than C slots (the 600-frame configuration:
// 700 border rows), the band is wider than the slot ring or the workgroup is not eight wavefronts.
//
// STATUS (profiles/r03b_register_front/, MI355X): correct on the first GPU run (1e-15 of the right-looking solution on every stage) and
// NOT faster.  One workgroup on an idle GPU, stage 2.2 of a 90-frame sequence (N = 2 125): 3.2 ms against 1.78 ms right-looking (2.2 ms
// left-looking); the whole bench workload: 368 against 500 sequences/s.  129 panels of 16 columns at ~25 us each: entry + extraction 7.4 us,
// diagonal block 4.1 us (slot bookkeeping 3.2 us beside it), row solve 3.4 us, update 9.3 us -- of which the tile wavefronts' matrix-core
// work is 2.8 us since it became branch-free straight-line code (8.5 us with a skip test per tile); the rest of that phase is the service
// wavefronts gathering the entering rows' K0 values, ~3 500 scattered loads per panel.  The panel's critical path has no window
// read-modify-write any more, but it is instruction-bound instead: slot indirection (every register <-> LDS move is predicated on table
// look-ups), 15-18 statically owned tiles per wavefront, loop invariants spilled to scratch beside 120-144 accumulator registers, and a
// 16 x 16 pivot block + row solve that are serial chains repeated twice as often as with 32-column panels.  What it would take: entry and
// extraction as straight-line code too (slots handed out by whole 16-slot blocks so that both are plain tile copies), the entering rows'
// K0 segments staged as contiguous rows, 32-column panels, the row solve as a matrix-core triangular solve.  Kept selectable
// (chd_config.factorisation = 2) and tested (tests/test_gpu_parity.py, tests/test_host_emu.py).
#define CHD_RF_CMAX 224           // slots of the largest instantiation (sizes the LDS tables); instantiated for 208 (stages without duration
                                 // variables: fronts <= 170 rows on 90-frame sequences) and 224 (duration stage: <= 210)
#define CHD_RF_NBKMAX (CHD_RF_CMAX / 16)
#define CHD_RF_RING 512
#define CHD_RF_BMAX 256
#define CHD_RF_NB 16

struct RfLds {                     // LDS layout of the factorisation (after the reduction scratch)
  LdsD* dv; LdsD* DL; LdsD* PT;
  LdsI* rowslot;                   // [RING + BMAX]: slot of band row i at i & (RING - 1), of border row r at RING + r; -1 = not in the front
  LdsI* slotrow;                   // [CMAX]: KKT position held by the slot, -1 = free
  LdsI* freel;                     // [CMAX]: stack of free slots
  LdsI* pjA; LdsI* pjB;            // [CMAX] each: pivot index (0..15) of a slot in the current / next panel, -1 otherwise
  LdsI* isnew;                     // [CMAX]: 1 = the slot's row enters at the NEXT panel (not part of the current one)
  LdsI* psA; LdsI* psB;            // [16] each: slots of the current / next panel's pivots
  LdsI* misc;                      // [0] free slots, [1] overflow; block bit masks: [2] blocks with rows of the current panel's front, [3] with rows entering at
                                   // the next panel, [4] / [5] with pivots of the panel whose tables are pjA / pjB
  LdsI* pcA; LdsI* pcB;            // [NBKMAX] each: pivots of the current / next panel per block
  LdsI* newl;                      // [EMAX + 1]: [0] = rows entering at the next panel, then their slots (isnew[slot] = 1 + index in this list)
  LdsD* KST;                       // [EMAX][C]: K0 (+ shift) of the entering rows against every slot, staged by the service wavefronts
  LdsD* PTD;                       // [NB][C]: -d_j L(t, j): the update's second operand, scaled once by the row solve
};
#define CHD_RF_EMAX 32
#define CHD_RF_LDS_INTS (CHD_RF_RING + CHD_RF_BMAX + 5 * CHD_RF_CMAX + 32 + 2 + 2 * CHD_RF_NBKMAX + 2 + 2 * CHD_RF_NBKMAX + CHD_RF_EMAX + 2)
#define CHD_RF_LDS_DOUBLES_TABLES (64 + CHD_RF_NB * CHD_RF_NB + CHD_RF_NB * CHD_RF_CMAX + (CHD_RF_LDS_INTS + 1) / 2 + 2)
#define CHD_RF_LDS_DOUBLES (CHD_RF_LDS_DOUBLES_TABLES + CHD_RF_EMAX * CHD_RF_CMAX + CHD_RF_NB * CHD_RF_CMAX)
CHD_DEV RfLds rf_layout(LdsD* base) {
  RfLds L;
  L.dv = base; L.DL = L.dv + 64; L.PT = L.DL + CHD_RF_NB * CHD_RF_NB;
  LdsI* ip = (LdsI*)(L.PT + CHD_RF_NB * CHD_RF_CMAX);
  L.rowslot = ip; ip += CHD_RF_RING + CHD_RF_BMAX;
  L.slotrow = ip; ip += CHD_RF_CMAX;
  L.freel = ip; ip += CHD_RF_CMAX;
  L.pjA = ip; ip += CHD_RF_CMAX; L.pjB = ip; ip += CHD_RF_CMAX;
  L.isnew = ip; ip += CHD_RF_CMAX;
  L.psA = ip; ip += 16; L.psB = ip; ip += 16;
  L.misc = ip; ip += 2 + 2 * CHD_RF_NBKMAX + 2;
  L.pcA = ip; ip += CHD_RF_NBKMAX; L.pcB = ip; ip += CHD_RF_NBKMAX;
  L.newl = ip; ip += CHD_RF_EMAX + 2;
  L.KST = base + CHD_RF_LDS_DOUBLES_TABLES;
  L.PTD = L.KST + CHD_RF_EMAX * CHD_RF_CMAX;
  return L;
}

CHD_DEV int rf_ring_index(const LCtx& c, int p) { return p < c.Nb ? (p & (CHD_RF_RING - 1)) : CHD_RF_RING + (p - c.Nb); }

// ---- Between two panels (one wavefront, while another factors the diagonal block): the pivots of the panel being eliminated, [cp, cp + jp)
// with slots psp[], give their slots back (their columns of the front have been extracted: the register entries are dead), then the rows that
// enter the front at the panel [c0, c0 + jb) take slots, and that panel's pivot tables (pj, ps) are filled.  jp = 0: nothing to free.
#ifdef CHD_HOST_EMU
CHD_DEV void rf_enter(LCtx& c, const RfLds& L, const int C, LdsI* pj, LdsI* ps, LdsI* pc, const int c0, const int jb, const LdsI* psp, const int cp, const int jp) {
  const int Nb = c.Nb, w = c.w, bc = c.bc, last = c0 + jb - 1;
  (void)pc;
  for (int a = 0; a < jp; ++a) { const int s = psp[a]; L.slotrow[s] = -1; L.rowslot[rf_ring_index(c, cp + a)] = -1; L.misc[0] += 1; }
  for (int t = 0; t < C; ++t) { pj[t] = -1; L.isnew[t] = 0; }
  L.misc[3] = 0;
  if (jb <= 0) return;
  const int iend = c0 + jb + w < Nb ? c0 + jb + w : Nb;
  for (int pass = 0; pass < 2; ++pass) {
    const int n = pass == 0 ? iend - c0 : bc;
    for (int u = 0; u < n; ++u) {
      const int p = pass == 0 ? c0 + u : Nb + u;
      if (c.env[2 * p] > last || L.rowslot[rf_ring_index(c, p)] >= 0) continue;
      if (L.misc[0] <= 0) { L.misc[1] = 1; return; }
      int s;
      if (pass == 0) { s = L.misc[6] % C; while (L.slotrow[s] >= 0) s = (s + 1) % C; L.misc[6] = (s + 1) % C; }      // band rows: next free slot after the cursor
      else { s = C - 1; while (L.slotrow[s] >= 0) --s; }                                                        // border rows: highest free slot
      L.misc[0] -= 1;
      L.slotrow[s] = p; L.rowslot[rf_ring_index(c, p)] = s; L.isnew[s] = 1; L.misc[3] |= 1 << (s >> 4);
    }
  }
  for (int a = 0; a < 16; ++a) { ps[a] = a < jb ? L.rowslot[rf_ring_index(c, c0 + a)] : -1; if (a < jb) pj[ps[a]] = a; }
}
#else
CHD_DEV int rf_select64(const unsigned long long m, int k) {      // position of the k-th (0-based) set bit
  int pos = 0;
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) { const int cn = __popcll((m >> pos) & ((1ull << sft) - 1ull)); if (k >= cn) { k -= cn; pos += sft; } }
  return pos;
}
// Slots are given out with locality in mind -- the tile wavefronts skip whole 16-slot blocks by bit masks: band rows take the next free slot
// after a cursor (they enter and leave roughly in order, so a panel's pivots and its entering rows each sit in one or two blocks), border rows,
// which never leave, take the highest free slot.
CHD_DEV void rf_enter(LCtx& c, const RfLds& L, const int C, LdsI* pj, LdsI* ps, LdsI* pc, const int c0, const int jb, const LdsI* psp, const int cp, const int jp) {
  const int Nb = c.Nb, w = c.w, bc = c.bc, last = c0 + jb - 1;
  const int ln = threadIdx.x & 63;
  // candidates: band rows c0 .. c0 + jb + w - 1, then the border rows; their envelope starts are requested first, all together
  const int nband = jb > 0 ? (c0 + jb + w < Nb ? c0 + jb + w : Nb) - c0 : 0;
  const int wr = jb > 0 ? nband + bc : 0;
  constexpr int MR = 12;                       // 64 MR >= RING + BMAX
  int ef[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    const int u = 64 * r + ln;
    ef[r] = (64 * r < wr) ? c.env[u < wr ? 2 * (u < nband ? c0 + u : Nb + (u - nband)) : 0] : 0;
  }
  int nfree = L.misc[0], cursor = L.misc[6];
  if (ln < jp) { const int s = psp[ln]; L.slotrow[s] = -1; L.rowslot[rf_ring_index(c, cp + ln)] = -1; }
  nfree += jp;
  for (int t = ln; t < C; t += 64) { pj[t] = -1; L.isnew[t] = 0; }
  if (ln == 0) { L.misc[3] = 0; pc[0] = 0; }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
  bool over = false;
  int nnew = 0;
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    if (64 * r >= wr || over) continue;
    const int u = 64 * r + ln;
    const int p = u < nband ? c0 + u : Nb + (u - nband);
    const bool on = u < wr && ef[r] <= last && L.rowslot[rf_ring_index(c, u < wr ? p : c0)] < 0;
    const unsigned long long m = __ballot(on);
    const int cnt = __popcll(m);
    if (cnt == 0) continue;
    if (cnt > nfree) { over = true; continue; }
    const unsigned long long mband = __ballot(on && u < nband), mbord = m & ~mband;
    const int e = nnew + __popcll(m & ((1ull << ln) - 1ull));
    auto commit = [&](const bool mine, const int slot) {
      if (mine) {
        L.slotrow[slot] = p; L.rowslot[rf_ring_index(c, p)] = slot; L.isnew[slot] = 1 + e;
        if (e < CHD_RF_EMAX) L.newl[1 + e] = slot;
        __hip_atomic_fetch_or(&L.misc[3], 1 << (slot >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
    };
    // band rows: the next free slots after the cursor (cyclic), in row order
    if (mband) {
      const int need = __popcll(mband), myk = __popcll(mband & ((1ull << ln) - 1ull));
      const bool mine = on && u < nband;
      int served = 0, slot = 0;
      for (int scanned = 0; served < need && scanned < C; scanned += 64) {
        int st = cursor + ln; st -= st >= C ? C : 0;
        const unsigned long long fm = __ballot(scanned + ln < C && L.slotrow[st] < 0);
        const int nf = __popcll(fm);
        if (mine && myk >= served && myk < served + nf) { int q = cursor + rf_select64(fm, myk - served); q -= q >= C ? C : 0; slot = q; }
        const int take = need - served < nf ? need - served : nf;
        served += take;
        cursor += (served < need || take == 0) ? 64 : rf_select64(fm, take - 1) + 1;          // past the last slot taken (or past an exhausted window)
        cursor -= cursor >= C ? C : 0;
      }
      commit(mine, slot);
    }
    // border rows: the highest free slots
    if (mbord) {
      const int need = __popcll(mbord), myk = __popcll(mbord & ((1ull << ln) - 1ull));
      const bool mine = on && u >= nband;
      int served = 0, slot = 0;
      for (int top = C - 1; served < need && top >= 0; top -= 64) {
        const int st = top - ln;
        const unsigned long long fm = __ballot(st >= 0 && L.slotrow[st >= 0 ? st : 0] < 0);
        const int nf = __popcll(fm);
        if (mine && myk >= served && myk < served + nf) slot = top - rf_select64(fm, myk - served);
        served += need - served < nf ? need - served : nf;
      }
      commit(mine, slot);
    }
    nfree -= cnt; nnew += cnt;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
  if (ln == 0) { L.misc[0] = nfree; L.misc[6] = cursor; if (over) L.misc[1] = 1; L.newl[0] = nnew; }
  if (ln < 16 && jb > 0) {
    const int s = (ln < jb && !over) ? L.rowslot[rf_ring_index(c, c0 + ln)] : -1;
    ps[ln] = s;
    if (s >= 0) { pj[s] = ln; __hip_atomic_fetch_or(&pc[0], 1 << (s >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  }
}
#endif

// value of the unfactored matrix (+ diagonal shift) for two KKT positions
CHD_DEV double rf_k0(const LCtx& c, const GD* diag, const int p, const int q) {
  return kget(c, p, q) + (p == q ? diag[p] : 0.0);
}
// slots the front of this stage may need (the duration stage's fronts are the large ones)
CHD_DEV int rf_capacity(const LCtx& c) { return c.S->opt_dur ? 224 : 208; }

#ifdef CHD_HOST_EMU
// ------------------------------------------------------------------------------------------
// host emulation: the same slot bookkeeping, the front as a plain C x C array
// ------------------------------------------------------------------------------------------
static long long chd_rf_completed = 0, chd_rf_refused = 0;      // (test instrumentation: factorisations that ran here / fell back)
CHD_DEV bool kfactor_rf_impl(LCtx& c, const GD* diag, const GI* sign);
CHD_DEV bool kfactor_rf(LCtx& c, const GD* diag, const GI* sign) { const bool ok = kfactor_rf_impl(c, diag, sign); if (ok) ++chd_rf_completed; else ++chd_rf_refused; return ok; }
CHD_DEV bool kfactor_rf_impl(LCtx& c, const GD* diag, const GI* sign) {
  const int Nb = c.Nb, w = c.w, W1 = c.w + 1, LD = c.LD, bc = c.bc;
  constexpr int NB = CHD_RF_NB;
  const int C = rf_capacity(c);
  if (w + 2 * NB >= CHD_RF_RING || bc > CHD_RF_BMAX || c.lds_cap - LDS_RED < CHD_RF_LDS_DOUBLES) return false;
  c.n_bad_pivots = 0;
  static std::vector<int> ibuf; ibuf.assign(CHD_RF_LDS_INTS + 16, 0);
  static std::vector<double> dbuf; dbuf.assign(64 + NB * NB + NB * CHD_RF_CMAX, 0.0);
  RfLds L;
  L.dv = dbuf.data(); L.DL = L.dv + 64; L.PT = L.DL + NB * NB;
  int* ip = ibuf.data();
  L.rowslot = ip; ip += CHD_RF_RING + CHD_RF_BMAX; L.slotrow = ip; ip += CHD_RF_CMAX; L.freel = ip; ip += CHD_RF_CMAX; L.pjA = ip; ip += CHD_RF_CMAX; L.pjB = ip; ip += CHD_RF_CMAX;
  L.isnew = ip; ip += CHD_RF_CMAX; L.psA = ip; ip += 16; L.psB = ip; ip += 16; L.misc = ip;
  for (int i = 0; i < CHD_RF_RING + CHD_RF_BMAX; ++i) L.rowslot[i] = -1;
  for (int t = 0; t < C; ++t) { L.slotrow[t] = -1; L.freel[t] = C - 1 - t; }
  L.misc[0] = C; L.misc[1] = 0; L.misc[6] = 0;
  static std::vector<double> Fm; Fm.assign((size_t)C * C, 0.0);
  auto F = [&](int a, int b) -> double& { return Fm[(size_t)(a > b ? a : b) * C + (a > b ? b : a)]; };
  auto init_new = [&]() {
    for (int a = 0; a < C; ++a) for (int b = 0; b <= a; ++b)
      if ((L.isnew[a] || L.isnew[b]) && L.slotrow[a] >= 0 && L.slotrow[b] >= 0) F(a, b) = rf_k0(c, diag, L.slotrow[a], L.slotrow[b]);
  };
  LdsI* pj = L.pjA; LdsI* ps = L.psA; LdsI* pjn = L.pjB; LdsI* psn = L.psB;
  rf_enter(c, L, C, pj, ps, nullptr, 0, Nb < NB ? Nb : NB, ps, 0, 0);
  if (L.misc[1]) return false;
  init_new();
  for (int t = 0; t < C; ++t) L.isnew[t] = 0;
  for (int c0 = 0; c0 < Nb; c0 += NB) {
    const int jb = Nb - c0 < NB ? Nb - c0 : NB;
    // extract
    for (int j = 0; j < NB; ++j) for (int t = 0; t < C; ++t) L.PT[j * C + t] = j < jb ? F(t, ps[j]) : 0.0;
    // diagonal block
    double A[NB][NB];
    for (int a = 0; a < NB; ++a) for (int j = 0; j < NB; ++j) A[a][j] = (a < jb && j <= a) ? L.PT[j * C + ps[a]] : (a == j ? 1.0 : 0.0);
    for (int j = 0; j < NB; ++j) {
      double d = A[j][j];
      if (j < jb) d = pivot_fix(c, d, sign[c0 + j]);
      const double inv = 1.0 / d;
      for (int a = j + 1; a < NB; ++a) { A[a][j] *= inv; L.DL[j * NB + a] = A[a][j]; }
      L.dv[j] = d; L.dv[32 + j] = inv;
      for (int jj = j + 1; jj < NB; ++jj) for (int a = jj; a < NB; ++a) A[a][jj] -= A[a][j] * d * A[jj][j];
    }
    for (int a = 0; a < jb; ++a) {
      GD* dst = c.Kfb + (long long)(c0 + a) * W1 + (w - a);
      for (int j = 0; j < a; ++j) dst[j] = L.DL[j * NB + a];
      dst[a] = L.dv[a];
    }
    // this panel's pivots give their slots back; the next panel's entering rows take slots
    const int c0n = c0 + NB;
    rf_enter(c, L, C, pjn, psn, nullptr, c0n, c0n < Nb ? (Nb - c0n < NB ? Nb - c0n : NB) : 0, ps, c0, jb);
    if (L.misc[1]) return false;
#if CHD_INERTIA_RETRY && CHD_ABORT_BAD_FACTOR
    if (c.n_bad_pivots > 0) return true;
#endif
    // rows
    for (int t = 0; t < C; ++t) {
      const int p = L.slotrow[t];
      if (p < 0 || pj[t] >= 0 || L.isnew[t]) { for (int j = 0; j < NB; ++j) L.PT[j * C + t] = 0.0; continue; }
      double y[NB];
      for (int j = 0; j < NB; ++j) y[j] = L.PT[j * C + t];
      for (int k = 0; k < NB - 1; ++k) for (int j = k + 1; j < NB; ++j) y[j] -= y[k] * L.DL[k * NB + j];
      for (int j = 0; j < NB; ++j) {
        const double l = y[j] * L.dv[32 + j];
        L.PT[j * C + t] = l;
        if (j < jb) {
          if (p < Nb) { if (p - (c0 + j) <= w) c.Kfb[(long long)p * W1 + (c0 + j - p + w)] = l; }
          else c.Kfx[(long long)(p - Nb) * LD + c0 + j] = l;
        }
      }
    }
    // update + entering rows of the next panel
    for (int a = 0; a < C; ++a) for (int b = 0; b <= a; ++b) {
      double s = 0;
      for (int j = 0; j < NB; ++j) s += L.PT[j * C + a] * (L.dv[j] * L.PT[j * C + b]);
      F(a, b) -= s;
    }
    init_new();
    LdsI* t1 = pj; pj = pjn; pjn = t1; t1 = ps; ps = psn; psn = t1;
  }
  // border: Schur complement from the front
  if (bc > 0 && !(CHD_INERTIA_RETRY && CHD_ABORT_BAD_FACTOR && c.n_bad_pivots > 0)) {
    const bool in_lds = (long long)bc * bc <= c.lds_cap - LDS_RED;
    LdsD* SLl = c.lds + LDS_RED;
    for (int r = 0; r < bc; ++r) for (int q = 0; q <= r; ++q) {
      const int sr = L.rowslot[CHD_RF_RING + r], sq = L.rowslot[CHD_RF_RING + q];
      const double v = (sr >= 0 && sq >= 0) ? F(sr, sq) : rf_k0(c, diag, Nb + r, Nb + q);
      if (in_lds) SLl[r * bc + q] = v; else c.Kfx[(long long)r * LD + Nb + q] = v;
    }
    if (in_lds) {
      dense_ldlt(c, SLl, bc, bc, sign + Nb);
      for (int idx = 0; idx < bc * bc; ++idx) { const int r = idx / bc, k = idx % bc; if (k <= r) c.Kfx[(long long)r * LD + Nb + k] = SLl[idx]; }
    } else dense_ldlt(c, c.Kfx + Nb, LD, bc, sign + Nb);
  }
  return true;
}
#else
// ------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------
// lower-triangular tile number -> (row block, column block); -1 beyond the last tile
CHD_DEV void rf_tile_coords(const int tno, const int ntiles, int& br, int& bcl) {
  if (tno >= ntiles) { br = -1; bcl = -1; return; }
  int r = (int)((__fsqrt_rn(8.0f * (float)tno + 1.0f) - 1.0f) * 0.5f);
  while ((r + 1) * (r + 2) / 2 <= tno) ++r;
  while (r * (r + 1) / 2 > tno) --r;
  br = r; bcl = tno - r * (r + 1) / 2;
}
#define CHD_RF_FOR_TILES(k) _Pragma("unroll") for (int k = 0; k < TPW; ++k)

// diagonal block of a panel: the 16 x 16 pivot block from PT (pivot a's row at slot ps[a]) -> unit-lower DL, pivots dv (+ reciprocals at dv[32 ..]),
// and the block's rows of the factor storage.  One wavefront.
template <int C>
CHD_NOINLINE CHD_DEV void rf_diag(LCtx& c, const GI* sign, LdsD* dv, LdsD* DL, const LdsD* PT, const LdsI* ps, const int c0, const int jb) {
  constexpr int NB = CHD_RF_NB;
  const int lane = threadIdx.x & 63;
  const int W1 = c.w + 1, w = c.w;
      const int a = lane;
      const bool act = a < NB;
      const int sa = act ? ps[a] : -1;
      const int sg_a = a < jb ? sign[c0 + a] : 1;
      double ar[NB];          // row a of the block, one batch of LDS reads
#pragma unroll
      for (int j = 0; j < NB; ++j) ar[j] = (sa >= 0 && j <= a) ? PT[j * C + sa] : (a == j ? 1.0 : 0.0);
      const unsigned long long sg_pos = __ballot(sg_a > 0);
      double u[NB], row[NB];
#pragma unroll
      for (int k2 = 0; k2 < NB; ++k2) { u[k2] = 0.0; row[k2] = 0.0; }
      double lprev = 0.0;
      int bad = 0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (j > 0) row[j - 1] = readlane_f64(lprev, j);
        double s0 = ar[j], s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k2 = 0; k2 + 3 < j; k2 += 4) { s0 -= u[k2] * row[k2]; s1 -= u[k2 + 1] * row[k2 + 1]; s2 -= u[k2 + 2] * row[k2 + 2]; s3 -= u[k2 + 3] * row[k2 + 3]; }
#pragma unroll
        for (int k2 = j & ~3; k2 < j; ++k2) s0 -= u[k2] * row[k2];
        const double v = (s0 + s1) + (s2 + s3);
        if (j + 1 < NB) {
#pragma unroll
          for (int k2 = 0; k2 < j; ++k2) row[k2] = DL[k2 * NB + (j + 1)];
        }
        double d = readlane_f64(v, j);
        if (j < jb) { const double sg = ((sg_pos >> j) & 1ull) ? 1.0 : -1.0; if (!(d * sg > 1e-14)) { d = sg * 1e-10; ++bad; } }
        const double inv = rcp_f64(d);
        const double lj = v * inv;
        u[j] = v; lprev = lj;
        if (act && a > j) DL[j * NB + a] = lj;
        if (a == j) { dv[j] = d; dv[32 + j] = inv; }
        if (a < jb && j <= a) c.Kfb[(long long)(c0 + a) * W1 + (w - a + j)] = (j == a) ? d : lj;
      }
      if (lane == 0) c.n_bad_pivots += bad;
}

// rows of the front against the diagonal block, in place in PT (one slot per thread; threads t0, t0 + nt, ... < C), and the mask of the 16-slot
// blocks that hold such rows.  A pivot's slot may already belong to a row entering at the next panel: neither is a row of this panel.
template <int C>
CHD_DEV void rf_rows(const RfLds& L, const LdsI* pj, const int t0, const int nt) {
  constexpr int NB = CHD_RF_NB;
  for (int t = t0; t < C; t += nt) {
    const int p = L.slotrow[t];
    const bool live = p >= 0 && pj[t] < 0 && !L.isnew[t];
    if (live) {
      double y0[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) y0[j] = L.PT[j * C + t];
#pragma unroll
      for (int k2 = 0; k2 < NB - 1; ++k2) {
#pragma unroll
        for (int j = k2 + 1; j < NB; ++j) y0[j] -= y0[k2] * L.DL[k2 * NB + j];          // (broadcast reads of the diagonal block's column)
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) { L.PT[j * C + t] = y0[j] * L.dv[32 + j]; L.PTD[j * C + t] = -y0[j]; }          // (L = y / d, so -d L = -y)
      __hip_atomic_fetch_or(&L.misc[2], 1 << (t >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j) { L.PT[j * C + t] = 0.0; L.PTD[j * C + t] = 0.0; }
    }
  }
}
// the panel's columns of the factor: 8 consecutive columns of one row per task (tasks i0, i0 + ni, ...); nobody waits for these stores
template <int C>
CHD_DEV void rf_store(LCtx& c, const RfLds& L, const LdsI* pj, const int c0, const int jb, const int i0, const int ni) {
  constexpr int NB = CHD_RF_NB;
  const int Nb = c.Nb, w = c.w, W1 = c.w + 1, LD = c.LD;
  for (int idx = i0; idx < C * (NB / 8); idx += ni) {
    const int t = idx / (NB / 8), j0 = (idx % (NB / 8)) * 8;
    const int p = L.slotrow[t];
    if (p < 0 || pj[t] >= 0 || L.isnew[t]) continue;
    const bool band = p < Nb;
    GD* dst = band ? c.Kfb + (long long)p * W1 + (c0 + j0 - p + w) : c.Kfx + (long long)(p - Nb) * LD + c0 + j0;
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = L.PT[(j0 + q) * C + t];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (j0 + q < jb && (!band || p - (c0 + j0 + q) <= w)) dst[q] = v[q];
  }
}

// Two roles, two loops over the panels with the same sequence of workgroup barriers: the two SERVICE wavefronts (diagonal block; slot
// bookkeeping; staging of the entering rows' K0 values) and the six TILE wavefronts that hold the front.  Separate loops because register allocation is per function and per
// control-flow path: with the diagonal block's ~100 registers in the same loop as the accumulators the compiler spilled the front to
// scratch every panel.
#define CHD_RF_TILE_WAVES 6
// =================================== service wavefronts (0, 1) ===================================
template <int C>
CHD_NOINLINE CHD_DEV bool rf_service_loop(LCtx& c, const GD* diag, const GI* sign, LdsD* base) {
  constexpr int NB = CHD_RF_NB;
  const int Nb = c.Nb, LD = c.LD, bc = c.bc;
  const RfLds L = rf_layout(base);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    LdsI* pj = L.pjA; LdsI* ps = L.psA; LdsI* pjn = L.pjB; LdsI* psn = L.psB; LdsI* pcn = L.pcB; LdsI* pcc = L.pcA;
    __syncthreads();                                                                                                 // (c) first panel's entries are in
    if (wave == 1) { for (int t = lane; t < C; t += 64) L.isnew[t] = 0; if (lane == 0) { L.misc[3] = 0; L.newl[0] = 0; } }
    __syncthreads();                                                                                                 // (d)
    for (int c0 = 0; c0 < Nb; c0 += NB) {
      const int jb = Nb - c0 < NB ? Nb - c0 : NB;
      long long tp_ = CHD_CLOCK();
      if (jb < NB) { for (int i = tid; i < NB * C; i += 128) L.PT[i] = 0.0; __syncthreads(); }                       // (e: short last panel)
      __syncthreads();                                                                                               // (1) extracted
      TACC(c, 8, CHD_CLOCK() - tp_); tp_ = CHD_CLOCK();
      if (wave == 0) rf_diag<C>(c, sign, L.dv, L.DL, L.PT, ps, c0, jb);
      else {
        const int c0n = c0 + NB;
        const long long te_ = CHD_CLOCK();
        rf_enter(c, L, C, pjn, psn, pcn, c0n, c0n < Nb ? (Nb - c0n < NB ? Nb - c0n : NB) : 0, ps, c0, jb);
        if (lane == 0) { L.misc[2] = 0; c.tacc[13] += CHD_CLOCK() - te_; }          // (timer 13: the slot bookkeeping alone; only this thread writes it)
      }
      __syncthreads();                                                                                               // (2)
      TACC(c, 9, CHD_CLOCK() - tp_); tp_ = CHD_CLOCK();
      if (L.misc[1]) return false;
#if CHD_INERTIA_RETRY && CHD_ABORT_BAD_FACTOR
      if (c.n_bad_pivots > 0) return true;
#endif
      rf_rows<C>(L, pj, tid, 512);
      __syncthreads();                                                                                               // (3)
      TACC(c, 10, CHD_CLOCK() - tp_); tp_ = CHD_CLOCK();
      {   // K0 (+ shift) of the rows entering at the next panel against every slot -> LDS: the tile wavefronts pick them up after the barrier
        const int nnew = L.newl[0] < CHD_RF_EMAX ? L.newl[0] : (L.newl[0] > CHD_RF_EMAX ? 0 : CHD_RF_EMAX);          // (more than EMAX: the tile wavefronts load directly)
        const GD* safe = c.K0b + c.w;
        const int W2 = c.W2, wb = c.w;
        for (int i0 = tid; i0 < nnew * C; i0 += 128 * 8) {
          const GD* src[8]; double dg[8]; double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int idx = i0 + 128 * u;
            src[u] = safe; dg[u] = 0.0;
            if (idx < nnew * C) {
              const int e = idx / C, t = idx - e * C;
              const int p = L.slotrow[L.newl[1 + e]], pq = L.slotrow[t];
              if (pq >= 0) {
                if (p < Nb && pq < Nb) { const int dl = pq - p; if (dl <= wb && dl >= -wb) src[u] = c.K0b + (long long)p * W2 + (dl + wb); }
                else { const int hi = p > pq ? p : pq, lo = p > pq ? pq : p; src[u] = c.K0x + (long long)(hi - Nb) * LD + lo; }
                if (p == pq) dg[u] = diag[p];
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *src[u];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int idx = i0 + 128 * u; if (idx < nnew * C) L.KST[idx] = (src[u] != safe ? v[u] : 0.0) + dg[u]; }
        }
      }
      __syncthreads();                                                                                               // (4) updated
      TACC(c, 11, CHD_CLOCK() - tp_);
      LdsI* t1 = pj; pj = pjn; pjn = t1; t1 = ps; ps = psn; psn = t1; t1 = pcc; pcc = pcn; pcn = t1;
    }
    // border rows that never met a band column are not in the front: their entries of the Schur complement are K0's
    for (int idx = tid; idx < bc * bc; idx += 128) {
      const int r = idx / bc, q = idx % bc;
      if (q > r) continue;
      if (L.rowslot[CHD_RF_RING + r] < 0 || L.rowslot[CHD_RF_RING + q] < 0) c.Kfx[(long long)r * LD + Nb + q] = rf_k0(c, diag, Nb + r, Nb + q);
    }
    __syncthreads();                                                                                                 // (z)
    return true;
  }

// ===================================== tile wavefronts (2 .. 7) =====================================
template <int C>
CHD_NOINLINE CHD_DEV bool rf_tile_loop(LCtx& c, const GD* diag, LdsD* base) {
  constexpr int NB = CHD_RF_NB, NBK = C / 16, NT = NBK * (NBK + 1) / 2, TPW = (NT + CHD_RF_TILE_WAVES - 1) / CHD_RF_TILE_WAVES;
  const int Nb = c.Nb, LD = c.LD;
  const RfLds L = rf_layout(base);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  {
    const int tw = wave - 2, lr = lane & 15, lk = lane >> 4;
    int tbr[TPW], tbc[TPW];
    chd_f64x4 acc[TPW];
    CHD_RF_FOR_TILES(k) {
      int r_, c_; rf_tile_coords(tw + CHD_RF_TILE_WAVES * k, NT, r_, c_);
      tbr[k] = __builtin_amdgcn_readfirstlane(r_); tbc[k] = __builtin_amdgcn_readfirstlane(c_);      // (wave-uniform: scalar registers)
      acc[k] = chd_f64x4{0.0, 0.0, 0.0, 0.0};
    }
    LdsI* pj = L.pjA; LdsI* pjn = L.pjB; LdsI* pcc = L.pcA; LdsI* pcn = L.pcB;
    // front entries of the first panel's rows
    CHD_RF_FOR_TILES(k) {
      if (tbr[k] < 0) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rs = 16 * tbr[k] + lk + 4 * q, cs = 16 * tbc[k] + lr;
        const int p = L.slotrow[rs], pq = L.slotrow[cs];
        if ((L.isnew[rs] || L.isnew[cs]) && p >= 0 && pq >= 0) acc[k][q] = rf_k0(c, diag, p, pq);
      }
    }
    __syncthreads();                                                                                                 // (c)
    __syncthreads();                                                                                                 // (d)
    for (int c0 = 0; c0 < Nb; c0 += NB) {
      const int jb = Nb - c0 < NB ? Nb - c0 : NB;
      if (jb < NB) { for (int i = tid - 128; i < NB * C; i += 384) L.PT[i] = 0.0; __syncthreads(); }                 // (e)  [the service wavefronts zero the same array: harmless overlap]
      const long long tt0_ = CHD_CLOCK();
      // ---- the rows that enter at this panel: their entries of the front from the staged K0 values (or straight from K0 when more rows
      //      entered than the staging buffer holds)
      {
        const int nnew = __builtin_amdgcn_readfirstlane(L.newl[0]);
        const unsigned entm = (unsigned)__builtin_amdgcn_readfirstlane(L.misc[3]);
        if (nnew > 0) {
          CHD_RF_FOR_TILES(k) {
            if (tbr[k] < 0) continue;
            const int br = tbr[k], bcl = tbc[k];
            if (!(entm & ((1u << br) | (1u << bcl)))) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int rs = 16 * br + lk + 4 * q, cs = 16 * bcl + lr;
              const int er = L.isnew[rs], ec = L.isnew[cs];
              if (er == 0 && ec == 0) continue;
              if (nnew <= CHD_RF_EMAX) acc[k][q] = er ? L.KST[(er - 1) * C + cs] : L.KST[(ec - 1) * C + rs];
              else { const int p = L.slotrow[rs], pq = L.slotrow[cs]; if (p >= 0 && pq >= 0) acc[k][q] = rf_k0(c, diag, p, pq); }
            }
          }
        }
      }
      // ---- extract the pivot rows' columns of the front: PT[j][t] = F(t, slot of pivot j)
      const unsigned pivm = (unsigned)__builtin_amdgcn_readfirstlane(pcc[0]);
      CHD_RF_FOR_TILES(k) {
        if (tbr[k] < 0) continue;
        const int br = tbr[k], bcl = tbc[k];
        if (!(pivm & ((1u << br) | (1u << bcl)))) continue;          // no pivot of this panel in either block
        {   // the tile's columns that are pivots
          const int j = pj[16 * bcl + lr];
          if (j >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) L.PT[j * C + 16 * br + lk + 4 * q] = acc[k][q];
          }
        }
        if (br != bcl) {   // the tile's rows that are pivots (the transposed part of a pivot's column)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j = pj[16 * br + lk + 4 * q];
            if (j >= 0) L.PT[j * C + 16 * bcl + lr] = acc[k][q];
          }
        }
      }
      if (tid == 128) c.tacc[15] += CHD_CLOCK() - tt0_;          // (timers 14 / 15: the tile wavefronts' own update / entry + extraction time; only this thread writes them)
      __syncthreads();                                                                                               // (1)
      __syncthreads();                                                                                               // (2) diagonal block factored, next panel's rows have slots
      if (L.misc[1]) return false;
#if CHD_INERTIA_RETRY && CHD_ABORT_BAD_FACTOR
      if (c.n_bad_pivots > 0) return true;
#endif
      rf_rows<C>(L, pj, tid, 512);
      __syncthreads();                                                                                               // (3)
      rf_store<C>(c, L, pj, c0, jb, tid - 128, 384);         // the panel's columns of the factor (nobody waits for these stores)
      // ---- front -= L D L^T (matrix cores, operands from LDS); rows that enter at the next panel have no part in it (their PT rows are zero)
      const long long tt1_ = CHD_CLOCK();
      const unsigned occm = (unsigned)__builtin_amdgcn_readfirstlane(L.misc[2]);
      CHD_RF_FOR_TILES(k) {
        const int br = tbr[k] < 0 ? 0 : tbr[k], bcl = tbr[k] < 0 ? 0 : tbc[k];          // (a wavefront's unused tile slot mirrors tile (0, 0); nobody reads it)
        (void)occm;          // every tile, no branches: rows that are not in the panel's front are zero in PT / PTD, and straight-line code lets
                             // the LDS reads of one tile overlap the matrix-core chain of the previous one
#pragma unroll
        for (int s4 = 0; s4 < NB / 4; ++s4) {
          const int j = 4 * s4 + lk;
          acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(L.PT[j * C + 16 * br + lr], L.PTD[j * C + 16 * bcl + lr], acc[k], 0, 0, 0);
        }
      }
      if (tid == 128) c.tacc[14] += CHD_CLOCK() - tt1_;
      __syncthreads();                                                                                               // (4)
      LdsI* t1 = pj; pj = pjn; pjn = t1; t1 = pcc; pcc = pcn; pcn = t1;
    }
    // ---- what is left in the front is the border's Schur complement -> the border block of the factor storage (unfactored)
    CHD_RF_FOR_TILES(k) {
      if (tbr[k] < 0) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rs = 16 * tbr[k] + lk + 4 * q, cs = 16 * tbc[k] + lr;
        const int p = L.slotrow[rs], pq = L.slotrow[cs];
        if (p < Nb || pq < Nb) continue;                    // (free slots hold -1)
        if (tbr[k] == tbc[k] && pq > p) continue;           // diagonal tiles hold both triangles: one writer per entry
        const int hi = p > pq ? p : pq, lo = p > pq ? pq : p;
        c.Kfx[(long long)(hi - Nb) * LD + lo] = acc[k][q];
      }
    }
    __syncthreads();                                                                                                 // (z)
    return true;
  }
}

template <int C>
CHD_NOINLINE CHD_DEV bool kfactor_rf_band(LCtx& c, const GD* diag, const GI* sign, LdsD* base) {
  constexpr int NB = CHD_RF_NB, NBK = C / 16, NT = NBK * (NBK + 1) / 2, TPW = (NT + CHD_RF_TILE_WAVES - 1) / CHD_RF_TILE_WAVES;
  const int Nb = c.Nb, LD = c.LD, bc = c.bc;
  const RfLds L = rf_layout(base);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // ---- bookkeeping tables
  for (int i = tid; i < CHD_RF_RING + CHD_RF_BMAX; i += 512) L.rowslot[i] = -1;
  for (int t = tid; t < C; t += 512) { L.slotrow[t] = -1; L.freel[t] = C - 1 - t; L.isnew[t] = 0; L.pjA[t] = -1; L.pjB[t] = -1; }
  if (tid < 16) { L.psA[tid] = -1; L.psB[tid] = -1; }
  if (tid == 0) { L.misc[0] = C; L.misc[1] = 0; L.misc[6] = 0; c.n_bad_pivots = 0; }
  __syncthreads();                                                                                                   // (a)
  if (wave == 1) rf_enter(c, L, C, L.pjA, L.psA, L.pcA, 0, Nb < NB ? Nb : NB, L.psA, 0, 0);
  __syncthreads();                                                                                                   // (b)
  if (L.misc[1]) return false;
  return wave < 2 ? rf_service_loop<C>(c, diag, sign, base) : rf_tile_loop<C>(c, diag, base);
}

CHD_DEV bool kfactor_rf(LCtx& c, const GD* diag, const GI* sign) {
  const int Nb = c.Nb, w = c.w, LD = c.LD, bc = c.bc;
  if (blockDim.x != 512 || w + 2 * CHD_RF_NB >= CHD_RF_RING || bc > CHD_RF_BMAX || c.lds_cap - LDS_RED < CHD_RF_LDS_DOUBLES) return false;
  __syncthreads();
  const bool ok = rf_capacity(c) == 224 ? kfactor_rf_band<224>(c, diag, sign, c.lds + LDS_RED) : kfactor_rf_band<208>(c, diag, sign, c.lds + LDS_RED);
  if (!ok) { __syncthreads(); return false; }
  // ---- dense L D L^T of the border Schur complement
  const long long td_ = CHD_CLOCK();
  if (bc > 0 && !(CHD_INERTIA_RETRY && CHD_ABORT_BAD_FACTOR && c.n_bad_pivots > 0)) {
    LdsD* SL = c.lds + LDS_RED;
    if ((long long)bc * bc <= c.lds_cap - LDS_RED) {
      PAR_FOR(idx, bc * bc) { const int r = idx / bc, k = idx % bc; SL[idx] = k <= r ? c.Kfx[(long long)r * LD + Nb + k] : 0.0; }
      CHD_SYNC();
      dense_ldlt(c, SL, bc, bc, sign + Nb);
      PAR_FOR(idx, bc * bc) { const int r = idx / bc, k = idx % bc; if (k <= r) c.Kfx[(long long)r * LD + Nb + k] = SL[idx]; }
      CHD_SYNC();
    } else dense_ldlt(c, c.Kfx + Nb, LD, bc, sign + Nb);
  }
  TACC(c, 12, CHD_CLOCK() - td_);
  return true;
}
#endif
