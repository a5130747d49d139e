// Building blocks of the REGISTER-RESIDENT diagonal-block factorisation (gfx950), shared by the stand-alone kernel
// k_potf2_reg (kernels_chol.hip) and the diagonal role of the pipelined chain kernel k_potrf_pipe (kernels_pipe.hip):
// tile tables, the DPP pivot chain of the 16x16 diagonal tile, and rb_factor_block -- the body of the factorisation of one
// nbk x nbk block by one workgroup of 16 waves.  Replaces, on the device, the panel step of LAPACK dpotrf /
// linfa-linalg `cholesky()` (crates/gp/src/algorithm.rs:1004).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace egx {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// ---- in-launch hand-offs (k_potrf_pipe; MI355X guide, "agent-scope release/acquire"): payload stores are WRITE-THROUGH
// (sc1), every storing wave drains them (s_waitcnt vmcnt(0)) in front of a workgroup barrier, then ONE lane stores the flag
// (relaxed, agent scope).  Consumers poll that one word relaxed and read the payload with agent-scope (sc1) loads, or
// after one agent-scope acquire with plain loads.
__device__ __forceinline__ void store_d2_sc1(double *p, d2_t v) {
    // (inline asm is invisible to the hazard recogniser: the TWO wait states gfx940+ needs between a store of more than 64
    //  bits and a VALU write of its data registers are part of the statement -- with one, a v_and that followed the asm
    //  replaced the low word of a stored double in lanes 0..15)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <bool SC1>
__device__ __forceinline__ void store_d2(double *p, d2_t v) {
    if constexpr (SC1) store_d2_sc1(p, v);
    else *reinterpret_cast<d2_t *>(p) = v;
}
__device__ __forceinline__ double load_sc1(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ d2_t load_d2_sc1(const double *p) { return d2_t{load_sc1(p), load_sc1(p + 1)}; }
__device__ __forceinline__ int load_flag(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_flag(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// A: REGISTER-RESIDENT diagonal block factorisation (nbk x nbk, nbk a multiple of 16 up to 256) by ONE workgroup.  The lower triangle of the 256x256 block is 136
// tiles of 16x16; they live for the whole kernel as FP64-MFMA accumulators in the registers of seven "update" waves
// (<= 20 tiles = 160 VGPRs per wave), wave 0 is the "chain" wave.  Per 16-column strip k, two workgroup barriers:
//   phase A (waves 1-7)  TRSM_k: X_R = T(R,k) Linv_k^T for the tiles below the diagonal tile: ONE 4-MFMA chain per tile
//                        with the accumulator registers themselves as the B operand -> X to the LDS panel (18-double
//                        rows) and to global.  The wave that owns (k+1,k) also owns (k+1,k+1): it updates that tile with
//                        its own X_{k+1} at once and hands it to the chain wave (LDS), so that the next diagonal tile
//                        never waits for the trailing update.
//   phase B              wave 0 factors + inverts tile (k+1,k+1) (the chain) WHILE waves 1-7 apply  -= X_R X_C^T  to all
//                        other tiles {C > k} (look-ahead inside the kernel; operands from the LDS panel).
// The chain: one matrix row per lane of a 16-lane DPP row (replicated in the four rows of the wave); right-looking
// Cholesky whose column broadcasts are DPP operands (v_fmac_f64_dpp row_newbcast: no LDS, no v_readlane), fused with
// the inverse of the factor (forward elimination of the column-scaled unit triangle, four columns per DPP row).  One
// wave issues one FP64 VALU instruction per ~8.5 cycles whether dependent or not (profiles/r02_run20_dp_latency.txt),
// so the chain is written for INSTRUCTION COUNT: v_rsq_f64 (2^-24) + one Halley step instead of two Newton steps, the
// diagonal entry from the same multiply as the column (p * rsqrt(p)), no selects on the factor (entries right of the
// diagonal carry garbage that only ever meets other such entries),
// the pivots checked once per strip (a failed one leaves NaNs), 1/L_ii through an LDS side buffer: ~21 VALU
// instructions per column.
// Accumulator convention: acc holds the tile transposed and with the columns permuted, so that no operand ever needs a
// transposition (and the one subtraction is the MFMA's own neg:[1,0,0]):
//     lane (frow, fk), register q   <->   T[row = frow][col = 4 fk + q]         (MFMA D[i][j], i = fk + 4 q, j = frow)
// which is at once the D layout of v_mfma_f64_16x16x4 and its A/B operand layout (k slot fk of MFMA step q <-> k =
// 4 fk + q; the same k permutation is used for the other operand, read as two ds_read_b128 of 4 consecutive doubles).
// With pi(i) = 4 (i % 4) + i / 4 (the column of D row i):   D[i][j] += sum_k A'[i][k] B'[k][j]  gives
//     update:  A' = -X_C[pi(i)][k], B' = X_R[j][k]                     -> T(R,C) -= X_R X_C^T
//     TRSM:    A' = Linv[pi(i)][k], B' = acc (= T(R,k)[j][k])          -> D = X_R in the same lane layout.
// The 64x64 tile inverses (`dinv`, used by the panel solve and the triangular solves) are built from the stored factor
// by k_diag_tile_inverses at the end of the factorisation; what the panel solve below the block needs instead -- the 16x16
// inverses and refinement flags -- is left in those tile-inverse slots, see store_diag.
// ---------------------------------------------------------------------------------------------
#ifndef RB_REFINE_LOG2
#define RB_REFINE_LOG2 5  // refine the strip's TRSM when log2(max |Linv| max L_ii) reaches this (-100: always, 100: never)
#endif
constexpr int RB_LD = 18;                 // LDS row stride (doubles): b128 reads of 16 rows hit 16 distinct 4-bank groups
constexpr int RB_P_DOUBLES = 256 * RB_LD;
constexpr int RB_RD_DOUBLES = 16 + 64 + 16;  // 1 / L_jj of the current strip + a dump area (see rb_chain_run)
constexpr int RB_LDS_BYTES = (RB_P_DOUBLES + 3 * 16 * RB_LD + RB_RD_DOUBLES) * 8 + 16;
// Tile -> (update wave, slot) for NU update waves, and per strip k the slots each phase touches (bit masks examined
// by the scalar unit: two instructions per slot instead of ten).  The pairs {(r, r-1), (r, r)} are dealt first (same
// wave, consecutive slots), then the other tiles in column-major order, each to the least loaded wave: every trailing
// set {C > k} is balanced to +-1 tile (offsets / strides of the pair dealing found by exhaustive search; the 16-wave
// kernel has its own table, below).
constexpr int rb_slots(int nu) { return nu == 7 ? 20 : (nu == 11 ? 13 : 9); }
template <int NU>
struct RbTab {
    int rc[NU][rb_slots(NU)];             // R * 16 + C, 0x100 = none (32-bit: fetched by scalar loads, not through vmcnt)
    unsigned int trsm[NU][16];            // phase A of strip k: slots with C == k, R > k
    unsigned int pair[NU][16];            // ... of those, the slot of (k+1, k) (slot + 1 is tile (k+1, k+1))
    unsigned int upd[NU][16];             // phase B of strip k: slots with C > k except (k+1, k+1)
};
// The 16-wave kernel's dealing (tools/rb_deal.py): update waves 3, 7, 11 (= waves 4, 8, 12, the ones that share the chain
// wave's SIMD: wave -> SIMD was measured as [0 2 1 3][w % 4] or [1 3 0 2][w % 4]) own tiles of columns 0..4 only, so that
// from strip 4 on -- where the strip time is the chain's -- no FP64 MFMA holds the DP pipe the chain's VALU instructions
// need (an MFMA blocks it for 64 cycles: the chain ran 2 - 2.8x slower beside the trailing update, profiles/r02_run19_*).
constexpr unsigned char rb_rc15[15][9] = {
    {0x32, 0x33, 0xfe, 0xff, 0x53, 0x94, 0xe5, 0xd7, 0xe9}, {0x21, 0x22, 0xed, 0xee, 0x72, 0xa4, 0xf5, 0xe7, 0xf9},
    {0x10, 0x11, 0xdc, 0xdd, 0x82, 0xb4, 0xa6, 0xf7, 0xca}, {0xf0, 0xc0, 0xf1, 0xc1, 0xf2, 0xc2, 0xf3, 0xc3, 0xf4},
    {0xcb, 0xcc, 0x20, 0x41, 0x92, 0xc4, 0xb6, 0xa8, 0xda}, {0xba, 0xbb, 0x30, 0x51, 0x63, 0x85, 0xc6, 0xb8, 0xdb},
    {0xa9, 0xaa, 0x40, 0x61, 0x73, 0x95, 0xd6, 0xc8, 0xea}, {0xe0, 0xb0, 0xe1, 0xb1, 0xe2, 0xb2, 0xe3, 0xb3, 0xe4},
    {0x98, 0x99, 0x50, 0x71, 0x83, 0xa5, 0xe6, 0xb9, 0xfa}, {0x87, 0x88, 0x60, 0x81, 0x93, 0xb5, 0xf6, 0xd8, 0xeb},
    {0x76, 0x77, 0x70, 0x91, 0x64, 0xc5, 0x97, 0xe8, 0xfb}, {0xd0, 0xa0, 0xd1, 0xa1, 0xd2, 0xa2, 0xd3, 0xa3, 0xd4},
    {0x65, 0x66, 0x80, 0x42, 0x74, 0x86, 0xa7, 0xf8, 0xec}, {0x54, 0x55, 0x90, 0x52, 0x75, 0x96, 0xb7, 0xc9, 0xfc},
    {0x43, 0x44, 0x31, 0x62, 0x84, 0xd5, 0xc7, 0xd9, 0xfd}};
template <int NU>
constexpr RbTab<NU> rb_make_tab() {
    constexpr int NS = rb_slots(NU);
    constexpr int off = NU == 7 ? 2 : 1, step = NU == 7 ? 6 : 3;
    RbTab<NU> t{};
    int cnt[NU] = {};
    for (int w = 0; w < NU; w++)
        for (int s = 0; s < NS; s++) t.rc[w][s] = 0x100;
    if (NU == 15) {
        for (int w = 0; w < NU; w++)
            for (int s = 0; s < NS; s++) t.rc[w][s] = rb_rc15[w % 15][s % 9];
    } else {
        for (int r = 1; r < 16; r++) {
            const int w = (off + step * (r - 1)) % NU;
            t.rc[w][cnt[w]++] = r * 16 + r - 1;
            t.rc[w][cnt[w]++] = r * 16 + r;
        }
        for (int c = 0; c < 16; c++)
            for (int r = c + 2; r < 16; r++) {
                int w = 0;
                for (int v = 1; v < NU; v++)
                    if (cnt[v] < cnt[w]) w = v;
                t.rc[w][cnt[w]++] = r * 16 + c;
            }
    }
    // Slots in the order of their first use (column of the tile; a pair counts as its tile (r, r-1) and stays together):
    // the block is loaded slot by slot, and what strip 0 needs should not queue behind what strip 9 needs.
    for (int w = 0; w < NU; w++) {
        int first[NS] = {}, len[NS] = {}, units = 0;
        for (int s0 = 0; s0 < NS && t.rc[w][s0] < 0x100;) {
            const int R = t.rc[w][s0] >> 4, C = t.rc[w][s0] & 15;
            const bool pair = C + 1 == R && s0 + 1 < NS && t.rc[w][s0 + 1] == R * 16 + R;
            first[units] = s0;
            len[units++] = pair ? 2 : 1;
            s0 += pair ? 2 : 1;
        }
        int sorted[NS] = {};
        bool used[NS] = {};
        int out = 0;
        for (int n = 0; n < units; n++) {  // selection sort, stable
            int best = -1;
            for (int v = 0; v < units; v++)
                if (!used[v] && (best < 0 || (t.rc[w][first[v]] & 15) < (t.rc[w][first[best]] & 15))) best = v;
            used[best] = true;
            for (int e = 0; e < len[best]; e++) sorted[out++] = t.rc[w][first[best] + e];
        }
        for (int s0 = 0; s0 < out; s0++) t.rc[w][s0] = sorted[s0];
    }
    for (int w = 0; w < NU; w++)
        for (int k = 0; k < 16; k++) {
            t.trsm[w][k] = t.pair[w][k] = t.upd[w][k] = 0;
            for (int s = 0; s < NS; s++) {
                if (t.rc[w][s] >= 0x100) continue;
                const int R = t.rc[w][s] >> 4, C = t.rc[w][s] & 15;
                if (C == k && R > k) t.trsm[w][k] |= 1u << s;
                if (C == k && R == k + 1) t.pair[w][k] |= 1u << s;
                if (C > k && !(C == k + 1 && R == k + 1)) t.upd[w][k] |= 1u << s;
            }
        }
    return t;
}
// every tile of the lower triangle except (0,0) exactly once; (r, r) right behind (r, r-1); masks consistent
template <int NU>
constexpr bool rb_tab_ok() {
    constexpr int NS = rb_slots(NU);
    const RbTab<NU> t = rb_make_tab<NU>();
    int seen[16][16] = {};
    for (int w = 0; w < NU; w++)
        for (int s = 0; s < NS; s++) {
            if (t.rc[w][s] >= 0x100) continue;
            const int R = t.rc[w][s] >> 4, C = t.rc[w][s] & 15;
            if (C > R || (R == 0 && C == 0)) return false;
            seen[R][C]++;
            if (R == C && (s == 0 || t.rc[w][s - 1] != R * 16 + R - 1)) return false;
            if (s > 0 && R != C) {  // sorted by column (a pair counts as its first tile)
                const int pr = t.rc[w][s - 1] >> 4, pc = t.rc[w][s - 1] & 15;
                if ((pr == pc ? pc - 1 : pc) > C) return false;
            }
        }
    for (int r = 0; r < 16; r++)
        for (int c = 0; c <= r; c++)
            if (seen[r][c] != ((r == 0 && c == 0) ? 0 : 1)) return false;
    for (int k = 0; k < 15; k++) {
        int trsm = 0, pair = 0, upd = 0;
        for (int w = 0; w < NU; w++)
            for (int s = 0; s < NS; s++) {
                trsm += (t.trsm[w][k] >> s) & 1;
                pair += (t.pair[w][k] >> s) & 1;
                upd += (t.upd[w][k] >> s) & 1;
            }
        if (trsm != 15 - k || pair != 1 || upd != (15 - k) * (16 - k) / 2 - 1) return false;
    }
    return true;
}
static_assert(rb_tab_ok<7>() && rb_tab_ok<11>() && rb_tab_ok<15>(), "tile tables of the register-resident diagonal-block kernel");
template <int NU>
__constant__ const RbTab<NU> c_rb_tab = rb_make_tab<NU>();

// DPP row broadcasts of FP64 operands (gfx90a+: row_newbcast is the one DPP control the DP ALU accepts).  Inline asm
// is invisible to the compiler's hazard recogniser, so the two wait states a DPP read needs behind a VALU write of
// its source are part of the statement.
template <int L>
__device__ __forceinline__ double dpp_bcast(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
// acc += (-src[lane L of this row]) * mul
template <int L>
__device__ __forceinline__ void dpp_fnmac(double &acc, double src, double mul) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mul), "n"(L));
}
// the same without the wait states: for a `src` that was written at least two instructions earlier (volatile asm
// statements keep their program order, so "behind two other DPP statements" is such a place)
template <int L>
__device__ __forceinline__ void dpp_fnmac_late(double &acc, double src, double mul) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mul), "n"(L));
}

// Chain state of wave 0 (its own code path in the kernel: wave 0 owns no tiles, the update waves no chain state)
struct RbChain {
    double a[16];    // matrix row `i` of the 16x16 tile, then of its factor (entries right of the diagonal: garbage)
    double z[4];     // row i of the inverse of the column-scaled factor, columns fk + 4 q
};

template <int J, int... Cs>
__device__ __forceinline__ void rb_chain_trailing(RbChain &ch, double l, std::integer_sequence<int, Cs...>) {
    (dpp_fnmac_late<J + 2 + Cs>(ch.a[J + 2 + Cs], l, l), ...);  // behind fnmac<J+1> and bcast<J+1>
}
template <int J, int... Qs>
__device__ __forceinline__ void rb_chain_inverse(RbChain &ch, double m, std::integer_sequence<int, Qs...>) {
    (dpp_fnmac_late<J>(ch.z[Qs], ch.z[Qs], m), ...);  // z was written a whole column earlier
}
// column J of the factorisation; `p` = the pivot, already broadcast.  Returns the broadcast pivot of column J + 1,
// which is formed first.
template <int J>
__device__ __forceinline__ double rb_chain_step(RbChain &ch, double p, int i, double *rd_lane) {
    // 1 / sqrt(p): hardware estimate (relative error <= 2^-24, measured) + one Halley step (cubic: <= 2^-70)
    const double y0 = __builtin_amdgcn_rsq(p);
    const double t = p * y0;
    const double e = __builtin_fma(-t, y0, 1.0);
    const double h = __builtin_fma(e, 0.375, 0.5);
    const double ye = y0 * e;
    const double y = __builtin_fma(ye, h, y0);
    // Column J: L_iJ = a_iJ / sqrt(p); in lane J this is p / sqrt(p) = L_JJ itself.  Lanes above the diagonal (i < J)
    // carry garbage (their a[C], C > i, received the updates of the columns <= i): it only ever meets other entries
    // right of the diagonal, is masked out of the inverse below and zeroed when the factor is stored.
    const double l = ch.a[J] * y;
    ch.a[J] = l;
    // (a pivot that is not positive and finite makes L_JJ and everything after it NaN: looked for once per strip, in
    // rb_chain_run -- the NaNs stay inside this workgroup, and every later kernel returns on entry)
    double pnext = 0.0;
    if constexpr (J < 15) {
        dpp_fnmac<J + 1>(ch.a[J + 1], l, l);
        pnext = dpp_bcast<J + 1>(ch.a[J + 1]);
    }
    if constexpr (J < 14) rb_chain_trailing<J>(ch, l, std::make_integer_sequence<int, 14 - J>{});
    rd_lane[J] = y;  // lane 0 -> rd[J] = 1 / L_JJ, every other lane into its own dump slot (no exec change, no conflict)
    // forward elimination step J of the inverse:  z_i -= (L_iJ / L_JJ) z_J  for the rows below
    const double m = (i > J) ? l * y : 0.0;
    rb_chain_inverse<J>(ch, m, std::make_integer_sequence<int, J / 4 + 1>{});
    return pnext;
}
template <int... Js>
__device__ __forceinline__ void rb_chain_all(RbChain &ch, int i, double *rd_lane, std::integer_sequence<int, Js...>) {
    double p = dpp_bcast<0>(ch.a[0]);
    ((p = rb_chain_step<Js>(ch, p, i, rd_lane)), ...);
}

// wave 0: factor + invert the diagonal tile of a strip (rows in ch.a); publishes Linv and the raw factor rows in LDS
__device__ __forceinline__ void rb_chain_run(RbChain &ch, double *NL, double *LR, double *rd, int lane, int frow, int fk,
                                             int *info, int gcol0, int n_valid, int *fail_flag, int *refine_flag) {
#pragma unroll
    for (int q = 0; q < 4; q++) ch.z[q] = (fk + 4 * q == frow) ? 1.0 : 0.0;
    double *rd_lane = (lane == 0) ? rd : rd + 16 + lane;
    rb_chain_all(ch, frow, rd_lane, std::make_integer_sequence<int, 16>{});
    const double r_own = rd[frow];
    // log2 of  max |Linv_ij| * max L_ii  -- a cheap stand-in for cond(L) of this tile (the spread of the diagonal alone
    // is not one: the factor of a noise-level Schur complement has a flat diagonal and cond ~ 1e6).  The high words of
    // positive doubles order like the doubles, so the maxima / minima are integer ones; the four DPP rows hold the same
    // rows, a rotate-and-combine over the 16 lanes of a row finishes them.  The update waves refine X = A Linv^T once
    // when this is >= RB_REFINE_LOG2, see the TRSM phase of the kernel.
    int hmax = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const double nl = r_own * ch.z[q];
        NL[frow * RB_LD + fk + 4 * q] = nl;
        hmax = max(hmax, __double2hiint(nl) & 0x7fffffff);
    }
    {
        int ymin = __double2hiint(r_own);
#define RB_ROR(v, n) __builtin_amdgcn_update_dpp(v, v, 0x120 + n, 0xf, 0xf, false)
        hmax = max(hmax, RB_ROR(hmax, 1)), ymin = min(ymin, RB_ROR(ymin, 1));
        hmax = max(hmax, RB_ROR(hmax, 2)), ymin = min(ymin, RB_ROR(ymin, 2));
        hmax = max(hmax, RB_ROR(hmax, 4)), ymin = min(ymin, RB_ROR(ymin, 4));
        hmax = max(hmax, RB_ROR(hmax, 8)), ymin = min(ymin, RB_ROR(ymin, 8));
#undef RB_ROR
        // this DPP row saw the columns fk + 4 q only: combine the four rows through the scalar unit
        int hall = hmax;
        hall = max(hall, __builtin_amdgcn_readlane(hmax, 16));
        hall = max(hall, __builtin_amdgcn_readlane(hmax, 32));
        hall = max(hall, __builtin_amdgcn_readlane(hmax, 48));
        if (lane == 0) *refine_flag = ((hall >> 20) - (ymin >> 20) >= RB_REFINE_LOG2) ? 1 : 0;
    }
    if (fk == 0) {
#pragma unroll
        for (int c = 0; c < 16; c += 2) *reinterpret_cast<d2_t *>(LR + frow * RB_LD + c) = d2_t{ch.a[c], ch.a[c + 1]};
    }
    // the diagonal of the factor, lane i its own L_ii (read back from the rows just written): all positive and finite,
    // or the first one that is not is the first pivot that failed
    const unsigned int ok = (unsigned int)__builtin_amdgcn_ballot_w64(__builtin_amdgcn_class(LR[frow * RB_LD + frow], 0x180)) & 0xffffu;
    if (ok != 0xffffu && lane == 0) {
        const int first_bad = __builtin_ctz(~ok);
        if (gcol0 + first_bad < n_valid) atomicCAS(info, 0, gcol0 + first_bad + 1);
        *fail_flag = 1;
    }
}

// one 16x16x16 product on the matrix cores: c +-= A' B' with both operands as 4 consecutive doubles per lane.  NEG = 1
// negates A' in the instruction (the FP64 MFMAs of gfx940+ read their BLGP field as neg:[a,b,c]).
#define RB_MFMA4(c, NEG, a01, a23, b0, b1, b2, b3)                               \
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[0], b0, c, 0, 0, NEG);           \
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[1], b1, c, 0, 0, NEG);           \
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[0], b2, c, 0, 0, NEG);           \
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[1], b3, c, 0, 0, NEG)

// Where the diagonal role of k_potrf_pipe publishes its finished strips: `strips` is the matrix' monotonic count of
// published 16-column strips of the whole factorisation (block at column col0: base = col0 / 16), kernels_pipe.hip.
struct RbPublish {
    int *strips = nullptr;
    int base = 0;
#ifdef EGX_PIPE_TRACE
    long long *trace = nullptr;  // (profiling builds: [6] / [7] receive the times of the first / last publication)
#endif
};

// The factorisation of ONE nbk x nbk diagonal block (nbk a multiple of 16 up to 256) by the calling workgroup of NW waves
// (1 chain wave + NW - 1 update waves), `sm` = RB_LDS_BYTES of LDS.  Returns false when a pivot failed here (info is set)
// or had failed before (the waves of the workgroup may return at different points, none of them in front of a barrier the
// others still reach).
// PIPE: everything a consumer outside this workgroup reads -- the strip's column of the factor, its 16 x 16 diagonal
// tile, the tile's inverse and refinement flag -- is stored write-through, drained at the end of the strip's trailing
// phase (a phase after it was issued: the wait is free) and published as strip `pub.base + k + 1` right behind that
// phase's barrier; the same arithmetic in the same order as the stand-alone kernel, hence the same bits.
template <int NW, bool PIPE>
__device__ __forceinline__ bool rb_factor_block(double *__restrict__ D, int64_t ld, int nbk, double *__restrict__ lin,
                                                int *__restrict__ info, int col0, int n_valid, double *sm, RbPublish pub) {
    constexpr int NU = NW - 1, RB_NS = rb_slots(NU);
    // examined once the loads of the block are on their way (below).  Only diagonal-block kernels SET the flag and the
    // chain runs them one after the other, so every wave of this workgroup reads the same value.
    const int failed_before = PIPE ? load_flag(info) : *info;
    __builtin_amdgcn_s_setprio(2);    // above the trailing-update workgroups this kernel may share its CU with
    double *P = sm;                   // X panel of the current strip, row = row of the block
    double *Dg = sm + RB_P_DOUBLES;   // hand-off of the next diagonal tile (update wave -> chain wave)
    double *NL = Dg + 16 * RB_LD;     // Linv of the current strip (chain wave -> update waves)
    double *LR = NL + 16 * RB_LD;     // raw rows of the strip's 16x16 factor (chain wave -> the update wave that stores it)
    double *rd = LR + 16 * RB_LD;     // 1 / L_jj of the strip being factored
    int *flag = reinterpret_cast<int *>(rd + RB_RD_DOUBLES);
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb16 = nbk >> 4;
    const int prow = ((frow & 3) << 2) | (frow >> 2);  // pi(frow)
    if (tid == 0) *flag = 0;
    if (wave == 0) {
        // ================= chain wave: no tiles, no MFMA; two barriers per strip like the update waves
        __builtin_amdgcn_s_setprio(3);
        RbChain ch;
        const double *src = D + (int64_t)frow * ld;
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            const d2_t v = *reinterpret_cast<const d2_t *>(src + c);
            ch.a[c] = v[0];
            ch.a[c + 1] = v[1];
        }
        if (failed_before != 0) return false;  // a previous block of this factorisation already failed: early exit
        rb_chain_run(ch, NL, LR, rd, lane, frow, fk, info, col0, n_valid, flag, flag + 1);
        __syncthreads();
        if (*flag) return false;
        for (int k = 0; k + 1 < nb16; k++) {
            __syncthreads();  // phase A of strip k done: tile (k+1, k+1) is in Dg
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                const d2_t v = *reinterpret_cast<const d2_t *>(Dg + frow * RB_LD + c);
                ch.a[c] = v[0];
                ch.a[c + 1] = v[1];
            }
            rb_chain_run(ch, NL, LR, rd, lane, frow, fk, info, col0 + (k + 1) * 16, n_valid, flag, flag + 1);
            __syncthreads();
            if (*flag) return false;
        }
        if (PIPE) __syncthreads();  // the last strip is stored and drained (update waves, below)
        return true;
    }
    // ===================== update waves 1..NU
    const RbTab<NU> &tab = c_rb_tab<NU>;
    double acc[RB_NS * 4];  // slot s, register q: acc[4 s + q]
    int rc[RB_NS];          // (wave-uniform) tile of slot s: R * 16 + C
    unsigned int valid = 0;  // slots whose tile exists in an nbk x nbk block
#pragma unroll
    for (int s = 0; s < RB_NS; s++) {
        int t = __builtin_amdgcn_readfirstlane(tab.rc[wave - 1][s]);
        asm volatile("" : "+s"(t));  // keep it in its SGPR: re-loading it from the table costs a memory round trip per use
        rc[s] = t;
        // The loads are unconditional (a slot without a tile re-reads tile (0,0) and never uses it) and the loaded
        // registers are not touched before the tile's first MFMA: straight-line loads let the compiler wait for exactly
        // the ones a phase needs (s_waitcnt vmcnt(n) counts in issue order = order of first use), and nothing is waited
        // for before the first barrier -- 31k cycles, a fifth of the kernel, when the tiles were negated on arrival.
        const bool have = t < 0x100 && (t >> 4) < nb16;
        if (have) valid |= 1u << s;
        const int tt = have ? t : 0;
        const double *src = D + (int64_t)((tt >> 4) * 16 + frow) * ld + (tt & 15) * 16 + 4 * fk;
        const d2_t v0 = *reinterpret_cast<const d2_t *>(src), v1 = *reinterpret_cast<const d2_t *>(src + 2);
        acc[4 * s] = v0[0];
        acc[4 * s + 1] = v0[1];
        acc[4 * s + 2] = v1[0];
        acc[4 * s + 3] = v1[1];
    }
    if (failed_before != 0) return false;  // a previous block of this factorisation already failed: early exit
    // the strictly upper 16x16 tiles of the 64x64 diagonal tiles are part of the factor's contract: zeros
    for (int zt = wave - 1; zt < 24; zt += NU) {
        const int t = zt / 6, e = zt % 6;
        const int r = (e < 3) ? 0 : (e < 5 ? 1 : 2), c = (e < 3) ? e + 1 : (e < 5 ? e - 1 : 3);
        if (4 * t + c < nb16) {
            double *dst = D + (int64_t)((4 * t + r) * 16 + frow) * ld + (4 * t + c) * 16 + 4 * fk;
            *reinterpret_cast<d2_t *>(dst) = d2_t{0.0, 0.0};
            *reinterpret_cast<d2_t *>(dst + 2) = d2_t{0.0, 0.0};
        }
    }
    // the 16x16 factor of strip k (raw rows in LR): zero right of the diagonal, to global
    auto store_diag = [&](int k) {
        const double *lr = LR + frow * RB_LD + 4 * fk;
        const d2_t v0 = *reinterpret_cast<const d2_t *>(lr), v1 = *reinterpret_cast<const d2_t *>(lr + 2);
        double *dst = D + (int64_t)(k * 16 + frow) * ld + k * 16 + 4 * fk;
        store_d2<PIPE>(dst, d2_t{(4 * fk <= frow) ? v0[0] : 0.0, (4 * fk + 1 <= frow) ? v0[1] : 0.0});
        store_d2<PIPE>(dst + 2, d2_t{(4 * fk + 2 <= frow) ? v1[0] : 0.0, (4 * fk + 3 <= frow) ? v1[1] : 0.0});
        // ... and its inverse + refinement flag, for the solve of the rows below the block (k_panel_trsm16): they travel in
        // the slot of the 64x64 tile inverse this strip belongs to (tile j = k % 4 at doubles [256 j, 256 j + 256), flags at
        // 1024 + j), which k_diag_tile_inverses overwrites with the 64x64 inverses once the factorisation is complete
        const double *nl = NL + frow * RB_LD + 4 * fk;
        double *dl = lin + (int64_t)(k >> 2) * 4096 + (k & 3) * 256 + frow * 16 + 4 * fk;
        store_d2<PIPE>(dl, *reinterpret_cast<const d2_t *>(nl));
        store_d2<PIPE>(dl + 2, *reinterpret_cast<const d2_t *>(nl + 2));
        if (lane == 0) {
            double *fl = lin + (int64_t)(k >> 2) * 4096 + 1024 + (k & 3);
            const double fv = flag[1] ? 1.0 : 0.0;
            if constexpr (PIPE) __hip_atomic_store(fl, fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *fl = fv;
        }
    };
    unsigned int m_trsm = tab.trsm[wave - 1][0] & valid, m_pair = tab.pair[wave - 1][0] & valid, m_upd = tab.upd[wave - 1][0] & valid;
    __syncthreads();
    if (*flag) return false;
    // One strip.  It is a lambda so that strip 0 can be run OUTSIDE the loop: the compiler drains every outstanding load
    // before it enters a loop that uses them (s_waitcnt vmcnt(0) at the preheader), which made the TRSM of strip 0 --
    // it needs one or two of a wave's nine tiles -- wait for the whole block.
    auto strip = [&](int k) -> bool {
        // the slot masks of the next strip: scalar loads, a strip ahead of their use
        const unsigned int n_trsm = tab.trsm[wave - 1][k + 1] & valid, n_pair = tab.pair[wave - 1][k + 1] & valid,
                           n_upd = tab.upd[wave - 1][k + 1] & valid;
        // Per-slot addresses are recomputed per strip from these three: left to itself the compiler hoists 20 slots' worth
        // of loop-invariant row offsets out of the strip loop and spills them (measured: 52 VGPR spills).
        int fr = frow, pr = prow, f4 = 4 * fk;
        asm volatile("" : "+v"(fr), "+v"(pr), "+v"(f4));
        // ---- phase A: TRSM_k (the tiles (R, k), R > k, become X_R); tile (k+1, k+1) is updated by its owner at once
        if (wave == 1 + k % NU) store_diag(k);
        if (m_trsm) {
            const d2_t n01 = *reinterpret_cast<const d2_t *>(NL + pr * RB_LD + f4);
            const d2_t n23 = *reinterpret_cast<const d2_t *>(NL + pr * RB_LD + f4 + 2);
            // X = A Linv^T with an explicitly inverted triangle leaves a residual A - X L^T of order eps cond(L) |A| -- not
            // the backward-stable eps |A| of a substitution -- and that residual lands in the Schur complement (a
            // numerically rank-deficient correlation matrix, kept positive definite by its nugget alone, then loses a
            // pivot).  One step of iterative refinement, X += (A - X L^T) Linv^T, restores eps |A| as long as
            // eps cond(L)^2 < 1; it is skipped for well-conditioned diagonal tiles (the chain wave's estimate, flag[1]).
            const bool refine = flag[1] != 0;
#pragma unroll
            for (int s = 0; s < RB_NS; s++) {
                if (m_trsm & (1u << s)) {
                    const int R = rc[s] >> 4;
                    double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
                    RB_MFMA4(x, 0, n01, n23, acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]);
                    if (refine) {
                        // A' operand of X L^T: rows pi(frow) of the raw factor, zero right of the diagonal (read per tile:
                        // held across the slots it costs eight registers the pipelined form of this block does not have)
                        const d2_t r01 = *reinterpret_cast<const d2_t *>(LR + pr * RB_LD + f4);
                        const d2_t r23 = *reinterpret_cast<const d2_t *>(LR + pr * RB_LD + f4 + 2);
                        const d2_t l01 = d2_t{(f4 <= pr) ? r01[0] : 0.0, (f4 + 1 <= pr) ? r01[1] : 0.0};
                        const d2_t l23 = d2_t{(f4 + 2 <= pr) ? r23[0] : 0.0, (f4 + 3 <= pr) ? r23[1] : 0.0};
                        // (the residual is formed IN the tile's own registers: the tile is dead once it has become X, and a copy
                        //  would cost eight registers at the point where the block has none to spare)
                        double4_t r = double4_t{acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};  // A
                        RB_MFMA4(r, 1, l01, l23, x[0], x[1], x[2], x[3]);                                      // A - X L^T
                        acc[4 * s] = r[0], acc[4 * s + 1] = r[1], acc[4 * s + 2] = r[2], acc[4 * s + 3] = r[3];
                        RB_MFMA4(x, 0, n01, n23, acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]);  // X + (A - X L^T) Linv^T
                    }
                    double *px = P + (R * 16 + fr) * RB_LD + f4;
                    *reinterpret_cast<d2_t *>(px) = d2_t{x[0], x[1]};
                    *reinterpret_cast<d2_t *>(px + 2) = d2_t{x[2], x[3]};
                    if constexpr (!PIPE) {  // (PIPE: the strip's column goes to global memory from the LDS panel, in phase B)
                        double *dst = D + (int64_t)(R * 16 + fr) * ld + k * 16 + f4;
                        *reinterpret_cast<d2_t *>(dst) = d2_t{x[0], x[1]};
                        *reinterpret_cast<d2_t *>(dst + 2) = d2_t{x[2], x[3]};
                    }
                    if (s + 1 < RB_NS && (m_pair & (1u << s))) {  // slot s + 1 is tile (k+1, k+1): T -= X X^T, A' = own LDS rows
                        const double *pa = P + (R * 16 + pr) * RB_LD + f4;
                        const d2_t a01 = *reinterpret_cast<const d2_t *>(pa), a23 = *reinterpret_cast<const d2_t *>(pa + 2);
                        double4_t c4 = double4_t{acc[4 * s + 4], acc[4 * s + 5], acc[4 * s + 6], acc[4 * s + 7]};
                        RB_MFMA4(c4, 1, a01, a23, x[0], x[1], x[2], x[3]);
                        acc[4 * s + 4] = c4[0], acc[4 * s + 5] = c4[1], acc[4 * s + 6] = c4[2], acc[4 * s + 7] = c4[3];  // (in place: dead after the hand-off)
                        double *dg = Dg + fr * RB_LD + f4;
                        *reinterpret_cast<d2_t *>(dg) = d2_t{c4[0], c4[1]};
                        *reinterpret_cast<d2_t *>(dg + 2) = d2_t{c4[2], c4[3]};
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase B: all other tiles right of column k, while wave 0 factors tile (k+1, k+1)
        if constexpr (PIPE) {
            // PIPE: update wave w first sends row tile w of the strip's column (X_w, in the LDS panel since phase A) to global
            // memory, write-through: the stores leave the TRSM phase (they cost it address registers it does not have, and
            // issue slots on the chain's critical path) for the phase in which the update waves wait for the chain anyway
            if (wave > k && wave < nb16) {
                const double *px = P + (wave * 16 + fr) * RB_LD + f4;
                double *dst = D + (int64_t)(wave * 16 + fr) * ld + k * 16 + f4;
                store_d2_sc1(dst, *reinterpret_cast<const d2_t *>(px));
                store_d2_sc1(dst + 2, *reinterpret_cast<const d2_t *>(px + 2));
            }
        }
#pragma unroll
        for (int s = 0; s < RB_NS; s++) {
            if (m_upd & (1u << s)) {
                const int R = rc[s] >> 4, C = rc[s] & 15;
                const double *pa = P + (C * 16 + pr) * RB_LD + f4;
                const double *pb = P + (R * 16 + fr) * RB_LD + f4;
                const d2_t a01 = *reinterpret_cast<const d2_t *>(pa), a23 = *reinterpret_cast<const d2_t *>(pa + 2);
                const d2_t b01 = *reinterpret_cast<const d2_t *>(pb), b23 = *reinterpret_cast<const d2_t *>(pb + 2);
                double4_t c4 = double4_t{acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};
                RB_MFMA4(c4, 1, a01, a23, b01[0], b01[1], b23[0], b23[1]);
                acc[4 * s] = c4[0];
                acc[4 * s + 1] = c4[1];
                acc[4 * s + 2] = c4[2];
                acc[4 * s + 3] = c4[3];
            }
        }
        if (PIPE) drain_stores();  // strip k's column, diagonal tile and inverse: issued a phase ago
        __syncthreads();
        if (*flag) return false;
        if (PIPE && wave == 1 && lane == 0) {
            store_flag(pub.strips, pub.base + k + 1);
#ifdef EGX_PIPE_TRACE
            if (pub.trace && k == 0) pub.trace[6] = wall_clock64();
#endif
        }
        m_trsm = n_trsm;
        m_pair = n_pair;
        m_upd = n_upd;
        return true;
    };
    if (nb16 > 1 && !strip(0)) return false;
    for (int k = 1; k + 1 < nb16; k++)
        if (!strip(k)) return false;
    if (wave == 1 + (nb16 - 1) % NU) store_diag(nb16 - 1);
    if (PIPE) {
        drain_stores();
        __syncthreads();
        if (wave == 1 && lane == 0) {
            store_flag(pub.strips, pub.base + nb16);
#ifdef EGX_PIPE_TRACE
            if (pub.trace) pub.trace[7] = wall_clock64();
#endif
        }
    }
    return true;
}

}  // namespace egx
