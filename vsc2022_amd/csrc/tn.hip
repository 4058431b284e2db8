// Temporal-Network localisation, one candidate pair per (single-wavefront) workgroup (gfx950).
//
// Replaces, per candidate pair (paths relative to /root/reference):
//   vsc/baseline/localization.py:36,52-54   sims = q.feature @ r.feature.T + bias
//   vsc/baseline/localization.py:58         vcsl.vta `tn` (third-party alipay/VCSL; the source is
//                                           absent from the reference checkout -> PARITY UNPINNED,
//                                           algorithm per SURVEY.md Appendix B, kept bit-identical
//                                           to oracle/vsc_oracle_tn.c)
//   vsc/baseline/localization.py:88-91      MaxSim box score
//
// Structure: the frame x frame similarity tile is produced on the matrix cores
// (v_mfma_f32_32x32x2_f32, operands straight from L2/HBM in the engine's k-interleaved layout,
// ascending-k fp32 chain) and held in LDS; per-row top-k, the banded DAG (edges are recomputed
// from per-row intermediate-range tables, never stored), the longest-path DP (one wavefront step
// per predecessor set, first-max tie-breaks as networkx) and the box logic all run out of LDS.
// Tiles that do not fit the LDS budget of a launch spill to a per-workgroup slab in HBM with
// identical results.
//
// Bound: HBM/L2 (algorithmic bytes per pair = 4*dim*(Lq+Lr) read + 16 B/box written); in practice
// instruction-issue-bound (78 k vector + 62 k scalar instructions per pair at the reference's parameters) -- DESIGN.md 8.6.
#include <cfloat>

#include "kernels.h"

namespace vscmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


// IDX: type of the stored reference / node indices -- short while the working state lives in LDS (<= 32767 nodes
// and frames), int for over-long videos whose state lives in an HBM slab
template <class IDX>
struct TnStateT {
    int lq, lr, top, ms, n_nodes, sink, sink_q, sink_r;
    unsigned top_magic;  // ceil(2^32 / top): (x * top_magic) >> 32 == x / top for x < 2^28 (top <= 16) -- node ids are far below
    float min_sim;
    const IDX* tidx;
    const float* tsim;
    const IDX* ilo;
    const IDX* ihi;
};

template <class IDX>
__device__ __forceinline__ int tn_div_top(const TnStateT<IDX>& g, int x) {
    return g.top == 1 ? x : (int)__umulhi((unsigned)x, g.top_magic);  // (2^32 / 1 does not fit the magic word)
}
template <class IDX>
__device__ __forceinline__ int tn_node_q(const TnStateT<IDX>& g, int v) { return v == 0 ? -1 : tn_div_top(g, v - 1); }
template <class IDX>
__device__ __forceinline__ int tn_node_k(const TnStateT<IDX>& g, int v) { return (v - 1) - tn_div_top(g, v - 1) * g.top; }
template <class IDX>
__device__ __forceinline__ int tn_node_r(const TnStateT<IDX>& g, int v) { return v == 0 ? -1 : g.tidx[v - 1]; }

template <class IDX>
__device__ __forceinline__ bool tn_edge_ok(const TnStateT<IDX>& g, int qi, int a, int d, int b) {
    const int qj = qi + d;
    const int ra = g.tidx[qi * g.top + a], rb = g.tidx[qj * g.top + b];
    const int rd = rb - ra;
    if (!(rd > 0 && rd < g.ms)) return false;
    const int lo = g.ilo[qi * g.ms + d], hi = g.ihi[qi * g.ms + d];
    if (lo <= hi && !(hi < ra || lo > rb)) return false;
    return g.tsim[qj * g.top + b] >= g.min_sim;
}

template <class IDX>
__device__ __forceinline__ bool tn_sink_ok(const TnStateT<IDX>& g, int u) {
    if (u == g.sink) return false;
    const int qu = tn_node_q(g, u), ru = tn_node_r(g, u);
    return g.sink_q > qu && g.sink_r > ru && g.sink_q - qu <= g.ms && g.sink_r - ru <= g.ms;
}

template <class IDX>
__device__ __forceinline__ int tn_edge_bit(const TnStateT<IDX>& g, int qi, int a, int d, int b) {
    return (((qi + d) * g.top + b) * g.ms + d) * g.top + a;
}

// (value, order) first-maximum reductions: larger value wins, ties go to the smaller order index, order < 0 = no entry.
// Round 5: the pair is folded into ONE 64-bit key whose unsigned maximum is that first maximum -- the order-preserving
// image of the float above (2^31 - 1 - order) -- so a butterfly step is two ds_bpermute, one 64-bit compare and two
// selects.  The compare-and-branch form it replaces cost ~7 VALU + ~12 SALU / branches per step (the compiler turns the
// short-circuit conditions into exec-mask surgery), and the kernel is instruction-issue-bound.
// -0.0 is folded onto +0.0 (they compare equal); a NaN ranks below every number and above "no entry".
__device__ __forceinline__ unsigned long long first_max_key(float v, int o) {
    const float vz = v + 0.0f;
    const unsigned b = __float_as_uint(vz);
    unsigned hi = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    hi = (vz != vz) ? 1u : hi;
    const unsigned long long k = ((unsigned long long)hi << 32) | (unsigned)(0x7fffffff - o);
    return o < 0 ? 0ull : k;
}
__device__ __forceinline__ void first_max_unkey(unsigned long long k, float& v, int& o) {
    const unsigned hi = (unsigned)(k >> 32);
    const float kv = __uint_as_float((hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi);  // (hi == 1 decodes to a NaN again)
    v = k ? kv : 0.0f;
    o = k ? 0x7fffffff - (int)(unsigned)k : -1;
}
template <int FIRST_OFF>
__device__ __forceinline__ void first_max_reduce(float& v, int& o) {
    unsigned long long k = first_max_key(v, o);
#pragma unroll
    for (int off = FIRST_OFF; off >= 1; off >>= 1) {
        const unsigned long long ok = __shfl_xor(k, off);
        k = ok > k ? ok : k;
    }
    first_max_unkey(k, v, o);
}
// over the whole wave / inside each half wave (xor offsets <= 16 stay inside a half)
__device__ __forceinline__ void wave_first_max(float& v, int& o) { first_max_reduce<32>(v, o); }
__device__ __forceinline__ void half_first_max(float& v, int& o) { first_max_reduce<16>(v, o); }


#ifdef VSC_TN_PROFILE  // experiment builds only: shader cycles per phase, summed over all pairs (scripts/experiments)
__device__ unsigned long long tn_prof[8];
#define TN_T0() unsigned long long t_prof = __builtin_readcyclecounter()
#define TN_T(k)                                                                          \
    do {                                                                                 \
        const unsigned long long t_now = __builtin_readcyclecounter();                   \
        if (threadIdx.x == 0) atomicAdd(&tn_prof[k], t_now - t_prof);                    \
        t_prof = t_now;                                                                  \
    } while (0)
#else
#define TN_T0() do {} while (0)
#define TN_T(k) do {} while (0)
#endif

template <class IDX, bool GSTATE>
__global__ __launch_bounds__(64) void tn_pair_kernel(TnPairArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_lds[];
    // working state: LDS, or this workgroup's slice of the HBM state slab (over-long videos; a single wavefront per
    // workgroup, so the workgroup barriers below order its accesses there as well)
    char* const smem = GSTATE ? a.state + (int64_t)blockIdx.x * a.state_bytes : smem_lds;
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= a.n_work) return;
    const int pidx = a.work[blockIdx.x];
    int64_t qrow0 = 0, rrow0 = 0;
    int lq, lr;
    if (a.sims_in) {  // forward_sim mode: the caller supplies the similarity matrices
        lq = a.sims_lq[pidx];
        lr = a.sims_lr[pidx];
    } else {
        const int qv = a.pair_q[pidx], rv = a.pair_r[pidx];
        qrow0 = a.q_off[qv];
        rrow0 = a.r_off[rv];
        lq = (int)(a.q_off[qv + 1] - qrow0);
        lr = (int)(a.r_off[rv + 1] - rrow0);
    }
    const int ms = a.prm.tn_max_step > 1 ? a.prm.tn_max_step : 1;
    const int top_cap = a.prm.tn_top_k;
    const int top = top_cap < lr ? top_cap : lr;
    int32_t* o_nbox = a.out_nbox + pidx;
    int32_t* o_boxes = a.out_boxes + (int64_t)pidx * VSC_TN_MAX_BOXES * 4;
    float* o_bmax = a.out_boxmax + (int64_t)pidx * VSC_TN_MAX_BOXES;
    if (lq <= 0 || top <= 0) {
        if (lane == 0) *o_nbox = 0;
        return;
    }
    const int n_nodes = 1 + lq * top;

    // ---- LDS carve (one dynamic array; offsets mirror tn_state_bytes) ----
    size_t off = 0;
    IDX* tidx = reinterpret_cast<IDX*>(smem + off);
    off += (size_t)a.max_lq * top_cap * sizeof(IDX);
    off = (off + 15) & ~(size_t)15;
    float* tsim = reinterpret_cast<float*>(smem + off);
    off += (size_t)a.max_lq * top_cap * 4;
    IDX* ilo = reinterpret_cast<IDX*>(smem + off);
    off += (size_t)a.max_lq * ms * sizeof(IDX);
    IDX* ihi = reinterpret_cast<IDX*>(smem + off);
    off += (size_t)a.max_lq * ms * sizeof(IDX);
    off = (off + 15) & ~(size_t)15;
    unsigned int* zero = reinterpret_cast<unsigned int*>(smem + off);
    const int zero_words = (lq * top * ms * top + 31) / 32;
    off += ((size_t)a.max_lq * top_cap * ms * top_cap + 31) / 32 * 4;
    off = (off + 15) & ~(size_t)15;
    float* dist = reinterpret_cast<float*>(smem + off);
    off += (size_t)(1 + a.max_lq * top_cap) * 4;
    IDX* par = reinterpret_cast<IDX*>(smem + off);
    off += (size_t)(1 + a.max_lq * top_cap) * sizeof(IDX);
    IDX* order = reinterpret_cast<IDX*>(smem + off);
    off += (size_t)(1 + a.max_lq * top_cap) * sizeof(IDX);
    IDX* indeg = reinterpret_cast<IDX*>(smem + off);
    off += (size_t)(1 + a.max_lq * top_cap) * sizeof(IDX);
    off = (off + 15) & ~(size_t)15;
    int* boxes = reinterpret_cast<int*>(smem + off);
    off += (size_t)VSC_TN_MAX_BOXES * 16;
    // which lanes of a (query row, node pair) pass of the DP hold a valid edge: the graph does not change between the
    // longest-path extractions (only edge weights are zeroed), so the first extraction records one ballot per pass and
    // the other max_path reuse it instead of re-deriving every edge from five LDS reads (tn_edge_ok)
    unsigned long long* okmask = reinterpret_cast<unsigned long long*>(smem + off);
    off += (size_t)a.max_lq * ((top_cap + 1) / 2) * 8;
    off = (off + 63) & ~(size_t)63;
    float* sims;
    if (a.sims_in) sims = const_cast<float*>(a.sims_in) + a.sims_off[pidx];
    else if ((int64_t)lq * lr <= a.lds_tile_floats) sims = reinterpret_cast<float*>(smem + off);
    else sims = a.slab + (int64_t)blockIdx.x * a.slab_floats;

    TN_T0();
    // ---- 1. similarity tile on the matrix cores: 32x32 output blocks, K ascending ----
    if (!a.sims_in) {
        const int hi = lane >> 5, l31 = lane & 31;
        const int nkg = a.dpad / 8;
        // Two column blocks per pass share the A fragments: the phase waits on row loads (2 KB rows from HBM, one
        // pair's 75 rows are read once), so what counts is the number of loads in flight per wave, not the MFMAs.
        for (int qb = 0; qb < lq; qb += 32)
            for (int rb = 0; rb < lr; rb += 64) {
                // rows past the video end read the (padded) neighbour rows; their results are dropped
                const bool two = rb + 32 < lr;
                const float* ap = a.qfeat + (qrow0 + qb + l31) * a.dpad + hi * 4;
                const float* bp0 = a.rfeat + (rrow0 + rb + l31) * a.dpad + hi * 4;
                const float* bp1 = bp0 + (two ? 32 * (int64_t)a.dpad : 0);
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
                if (two) {
#pragma unroll 4
                    for (int g = 0; g < nkg; ++g) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + g * 8);
                        const f32x4 bv0 = *reinterpret_cast<const f32x4*>(bp0 + g * 8);
                        const f32x4 bv1 = *reinterpret_cast<const f32x4*>(bp1 + g * 8);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv0[s], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv1[s], acc1, 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll 4
                    for (int g = 0; g < nkg; ++g) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + g * 8);
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bp0 + g * 8);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc0, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qb + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int rr = rb + l31;
                    if (q < lq && rr < lr) sims[(int64_t)q * lr + rr] = acc0[r] + a.bias;
                    if (two && q < lq && rr + 32 < lr) sims[(int64_t)q * lr + rr + 32] = acc1[r] + a.bias;
                }
            }
    }
    for (int x = lane; x < zero_words; x += 64) zero[x] = 0;
    __threadfence_block();
    __syncthreads();
    TN_T(0);

    // ---- 2. per-row top-k by (sim desc, ref asc) ----
    // Two equivalent methods.  Lane per ROW (top_k <= 8, enough rows to fill lanes): every lane walks its row once,
    // ascending r, keeping a sorted top-8 in registers (strict '>' on insertion: of equal sims the lower ref index
    // stays ahead) -- lr steps for 64 rows instead of `top` wave reductions per row.  Otherwise: `top` rounds of
    // wave arg-best per row (few long rows).
    if (top_cap <= 8 && (lq >= 16 || lr <= 128)) {
        for (int q0 = 0; q0 < lq; q0 += 64) {
            const int q = q0 + lane;
            float ts[8];
            int ti[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ts[e] = -INFINITY;
                ti[e] = -1;
            }
            if (q < lq) {
                const float* row = sims + (int64_t)q * lr;
                for (int r = 0; r < lr; ++r) {
                    const float sv = row[r];
                    if (!(sv > ts[7]) && ti[7] >= 0) continue;  // (an unfilled slot accepts anything, -inf included)
#pragma unroll
                    for (int e = 7; e >= 1; --e) {
                        const bool empty_above = ti[e - 1] < 0;
                        const bool up = empty_above || sv > ts[e - 1];  // belongs above slot e-1: that one moves down
                        const bool here = !up && (ti[e] < 0 || sv > ts[e]);
                        ts[e] = up ? ts[e - 1] : (here ? sv : ts[e]);
                        ti[e] = up ? ti[e - 1] : (here ? r : ti[e]);
                    }
                    if (ti[0] < 0 || sv > ts[0]) {
                        ts[0] = sv;
                        ti[0] = r;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e < top) {
                        tidx[q * top + e] = (IDX)ti[e];
                        tsim[q * top + e] = ts[e];
                    }
            }
        }
    } else {
        for (int q = 0; q < lq; ++q) {
            const float* row = sims + (int64_t)q * lr;
            float prev_s = INFINITY;
            int prev_r = -1;
            for (int e = 0; e < top; ++e) {
                float bs = 0.0f;
                int br = -1;
                for (int r = lane; r < lr; r += 64) {
                    const float s = row[r];
                    const bool after = (s < prev_s) || (s == prev_s && r > prev_r);
                    if (after && (br < 0 || s > bs)) {  // r ascends per lane: first max kept
                        bs = s;
                        br = r;
                    }
                }
                wave_first_max(bs, br);
                if (lane == 0) {
                    tidx[q * top + e] = (IDX)br;
                    tsim[q * top + e] = bs;
                }
                prev_s = bs;
                prev_r = br;
            }
        }
    }
    __syncthreads();

    TnStateT<IDX> g;
    g.lq = lq; g.lr = lr; g.top = top; g.ms = ms; g.n_nodes = n_nodes; g.sink = n_nodes - 1;
    g.sink_q = lq - 1;
    g.sink_r = tidx[(lq - 1) * top + top - 1];
    g.min_sim = a.prm.min_sim;
    g.top_magic = (unsigned)((0x100000000ull + (unsigned)top - 1u) / (unsigned)top);
    g.tidx = tidx; g.tsim = tsim; g.ilo = ilo; g.ihi = ihi;

    // ---- 3. intermediate ranges: one lane per source row q_i, steps d are sequential ----
    for (int qi = lane; qi < lq; qi += 64) {
        int lo = 1, hi = 0;
        for (int d = 1; d < ms; ++d) {
            ilo[qi * ms + d] = (IDX)lo;
            ihi[qi * ms + d] = (IDX)hi;
            if (qi + d >= lq) continue;
            int nlo = lo, nhi = hi;
            for (int b = 0; b < top; ++b) {
                bool any = false;
                for (int aa = 0; aa < top && !any; ++aa) any = tn_edge_ok(g, qi, aa, d, b);
                if (any) {
                    const int rb = tidx[(qi + d) * top + b];
                    if (nlo > nhi) nlo = nhi = rb;
                    else {
                        nlo = rb < nlo ? rb : nlo;
                        nhi = rb > nhi ? rb : nhi;
                    }
                }
            }
            lo = nlo;
            hi = nhi;
        }
    }
    __syncthreads();

    TN_T(1);
    // ---- 4. longest-path extractions ----
    const int P = (ms - 1) * top;  // regular predecessor slots of a node, insertion order
    int nbox = 0;
    bool have_order = false;
    // lane constants of the half-wave passes (P <= 32): predecessor slot o = (row distance d, top-k entry aa) of the lane, and
    // the lane's share of the index arithmetic -- what is left per pass is wave-uniform and runs on the scalar unit (round 5:
    // the divisions by `top` and two 64-bit multiply-adds per pass were a fifth of the kernel's vector instructions)
    const int hw_half = lane >> 5, hw_o = lane & 31;
    const int hw_d = ms - 1 - hw_o / top, hw_aa = hw_o % top;
    const int hw_bit_off = hw_d * top + hw_aa + hw_half * ms * top;
    const int hw_pred_off = 1 + hw_aa - hw_d * top;
    for (int it = 0; it <= a.prm.max_path; ++it) {
        // DP by query row (a valid topological order; dist does not depend on which one)
        if (lane == 0) {
            dist[0] = 0.0f;
            par[0] = 0;
        }
        __syncthreads();
        for (int qj = 0; qj < lq; ++qj) {
            // The nodes of one query row depend on earlier rows only.  When a node's predecessor slots fit 32 lanes
            // (the reference's tn_max_step = 5, top 5: 20), two nodes are evaluated per pass, one per half wave
            // (xor shuffles with offsets <= 16 stay inside a half); the sink keeps the whole-wave path below.
            int b_first = 0;
            if (P <= 32) {
                const int half = hw_half, o = hw_o, d = hw_d, aa = hw_aa;
                const int qi = qj - d;
                const int pred = qj * top + hw_pred_off;  // node id of this lane's predecessor slot: 1 + qi * top + aa
                for (int b0 = 0; b0 < top; b0 += 2) {
                    const int b = b0 + half;
                    const int v = 1 + qj * top + b;
                    const bool mine = b < top && v != g.sink;
                    float c = 0.0f;
                    int oo = -1;
                    const int mslot = qj * ((top + 1) / 2) + (b0 >> 1);
                    bool ok;
                    if (it == 0) {
                        ok = mine && o < P && qi >= 0 && tn_edge_ok(g, qi, aa, d, b);
                        const unsigned long long m = __ballot(ok);
                        if (lane == 0) okmask[mslot] = m;
                    } else {
                        ok = (okmask[mslot] >> lane) & 1ull;
                    }
                    if (ok) {
                        const int bit = (qj * top + b0) * ms * top + hw_bit_off;  // = tn_edge_bit(g, qi, aa, d, b)
                        const bool z = (zero[bit >> 5] >> (bit & 31)) & 1u;
                        const float w = z ? 0.0f : tsim[qj * top + b];
                        c = dist[pred] + w;
                        oo = o;
                    }
                    half_first_max(c, oo);  // (value, slot) first-max inside the half wave
                    const int pw = __shfl(pred, (lane & 32) | (oo < 0 ? 0 : oo));  // the winning slot's node id
                    if (mine && o == 0) {
                        if (oo < 0 || !(c >= 0.0f)) {
                            dist[v] = 0.0f;
                            par[v] = (IDX)v;
                        } else {
                            dist[v] = c;
                            par[v] = (IDX)pw;
                        }
                    }
                }
                // only the sink (last node of the last row) is left for the whole-wave path
                b_first = (qj == lq - 1) ? top - 1 : top;
            }
            if (qj == lq - 1) TN_T(2);
            for (int b = b_first; b < top; ++b) {
                const int v = 1 + qj * top + b;
                const bool is_sink = (v == g.sink);
                float best = 0.0f;
                int arg = -1;  // order slot of the best predecessor
                for (int o0 = 0; o0 < P; o0 += 64) {
                    const int o = o0 + lane;
                    float c = 0.0f;
                    int oo = -1;
                    if (o < P) {
                        const int d = ms - 1 - o / top, aa = o % top;
                        const int qi = qj - d;
                        if (qi >= 0 && tn_edge_ok(g, qi, aa, d, b)) {
                            const int bit = tn_edge_bit(g, qi, aa, d, b);
                            const bool z = (zero[bit >> 5] >> (bit & 31)) & 1u;
                            const float w = (z || is_sink) ? 0.0f : tsim[qj * top + b];
                            c = dist[1 + qi * top + aa] + w;
                            oo = o;
                        }
                    }
                    wave_first_max(c, oo);
                    if (oo >= 0 && (arg < 0 || c > best)) {
                        best = c;
                        arg = oo;
                    }
                }
                int arg_node = -1;
                if (arg >= 0) arg_node = 1 + (qj - (ms - 1 - arg / top)) * top + arg % top;
                if (is_sink) {
                    // sink-rule predecessors that are not regular ones, node-id order, weight 0
                    for (int u0 = 0; u0 < n_nodes - 1; u0 += 64) {
                        const int u = u0 + lane;
                        float c = 0.0f;
                        int oo = -1;
                        if (u < n_nodes - 1 && tn_sink_ok(g, u)) {
                            bool regular = false;
                            if (u != 0) {
                                const int qi = tn_node_q(g, u), aa = tn_node_k(g, u);
                                const int d = qj - qi;
                                if (d >= 1 && d < ms) regular = tn_edge_ok(g, qi, aa, d, b);
                            }
                            if (!regular) {
                                c = dist[u] + 0.0f;
                                oo = u;
                            }
                        }
                        wave_first_max(c, oo);
                        if (oo >= 0 && (arg_node < 0 || c > best)) {
                            best = c;
                            arg_node = oo;
                        }
                    }
                }
                if (lane == 0) {
                    if (arg_node < 0 || !(best >= 0.0f)) {
                        dist[v] = 0.0f;
                        par[v] = (IDX)v;
                    } else {
                        dist[v] = best;
                        par[v] = (IDX)arg_node;
                    }
                }
            }
            __syncthreads();
        }

        TN_T(3);
        // end node: first node in topological order with maximal dist
        float mx = -FLT_MAX;
        int mxn = -1;
        for (int v = lane; v < n_nodes; v += 64) {
            const float dv = dist[v];
            if (mxn < 0 || dv > mx) {
                mx = dv;
                mxn = v;
            }
        }
        wave_first_max(mx, mxn);  // provisional: smallest node id among the maxima
        int ties = 0;
        for (int v = lane; v < n_nodes; v += 64) ties += (dist[v] == mx) ? 1 : 0;
#pragma unroll
        for (int off2 = 32; off2 >= 1; off2 >>= 1) ties += __shfl_xor(ties, off2);
        int vend = mxn;
        if (!(mx > 0.0f)) {
            vend = 0;  // all-zero: the source is first in every topological order
        } else if (ties > 1) {
            if (!have_order) {
                // Kahn generations exactly as networkx.topological_generations (lazy: only when a
                // positive maximum is tied).  in-degrees in parallel, queue by one lane.
                for (int v = lane; v < n_nodes; v += 64) {
                    int deg = 0;
                    if (v != 0) {
                        const int qj = tn_node_q(g, v), b = tn_node_k(g, v);
                        for (int d = 1; d < ms; ++d) {
                            const int qi = qj - d;
                            if (qi < 0) break;
                            for (int aa = 0; aa < top; ++aa) deg += tn_edge_ok(g, qi, aa, d, b) ? 1 : 0;
                        }
                        if (v == g.sink) {
                            for (int u = 0; u < n_nodes - 1; ++u) {
                                if (!tn_sink_ok(g, u)) continue;
                                bool regular = false;
                                if (u != 0) {
                                    const int qi = tn_node_q(g, u), aa = tn_node_k(g, u);
                                    const int d = qj - qi;
                                    if (d >= 1 && d < ms) regular = tn_edge_ok(g, qi, aa, d, b);
                                }
                                deg += regular ? 0 : 1;
                            }
                        }
                    }
                    indeg[v] = (IDX)deg;
                }
                __syncthreads();
                if (lane == 0) {
                    int head = 0, tail = 0;
                    for (int v = 0; v < n_nodes; ++v)
                        if (indeg[v] == 0) order[tail++] = (IDX)v;
                    while (head < tail) {
                        const int u = order[head++];
                        bool to_sink_regular = false;
                        if (u != 0) {
                            const int qi = tn_node_q(g, u), aa = tn_node_k(g, u);
                            for (int d = 1; d < ms && qi + d < lq; ++d)
                                for (int b = 0; b < top; ++b)
                                    if (tn_edge_ok(g, qi, aa, d, b)) {
                                        const int v = 1 + (qi + d) * top + b;
                                        if (v == g.sink) to_sink_regular = true;
                                        if (--indeg[v] == 0) order[tail++] = (IDX)v;
                                    }
                        }
                        if (!to_sink_regular && tn_sink_ok(g, u))
                            if (--indeg[g.sink] == 0) order[tail++] = (IDX)g.sink;
                    }
                }
                have_order = true;
                __syncthreads();
            }
            // first tied node in topological order
            int bestpos = 0x7fffffff;
            for (int t = lane; t < n_nodes; t += 64)
                if (dist[order[t]] == mx && t < bestpos) bestpos = t;
#pragma unroll
            for (int off2 = 32; off2 >= 1; off2 >>= 1) {
                const int o2 = __shfl_xor(bestpos, off2);
                bestpos = o2 < bestpos ? o2 : bestpos;
            }
            vend = order[bestpos];
        }

        TN_T(4);
        // back-track, zero the path's edge weights, score and box (lane 0; paths are short)
        int accepted = 0, stop = 0;
        int bq0 = 0, br0 = 0, bq1 = 0, br1 = 0;
        if (lane == 0) {
            // pass 1: zero regular edges (walking end -> start) and count interior nodes
            int cnt = 0;
            for (int v = vend;;) {
                const int u = par[v];
                if (v != 0 && v != g.sink) ++cnt;
                if (u == v) break;
                if (u != 0) {
                    const int qi = tn_node_q(g, u), aa = tn_node_k(g, u);
                    const int qj = tn_node_q(g, v), b = tn_node_k(g, v);
                    const int d = qj - qi;
                    if (d >= 1 && d < ms && tn_edge_ok(g, qi, aa, d, b)) {
                        const int bit = tn_edge_bit(g, qi, aa, d, b);
                        zero[bit >> 5] |= 1u << (bit & 31);
                    }
                }
                v = u;
            }
            if (cnt == 0) {
                stop = 1;
            } else {
                // pass 2: the score is summed in FORWARD path order (start -> end).  Reverse the
                // parent chain in place through `order`-independent scratch: reuse indeg[] as stack.
                int plen = 0;
                for (int v = vend;;) {
                    indeg[plen++] = (IDX)v;
                    if (par[v] == v) break;
                    v = par[v];
                }
                float score = 0.0f;
                int qmin = 0, qmax = 0, rmin = 0, rmax = 0, c2 = 0;
                for (int x = plen - 1; x >= 0; --x) {
                    const int v = indeg[x];
                    if (v == 0 || v == g.sink) continue;
                    const int q = tn_node_q(g, v), r = tn_node_r(g, v);
                    score += tsim[v - 1];
                    if (c2 == 0) {
                        qmin = qmax = q;
                        rmin = rmax = r;
                    } else {
                        qmin = q < qmin ? q : qmin;
                        qmax = q > qmax ? q : qmax;
                        rmin = r < rmin ? r : rmin;
                        rmax = r > rmax ? r : rmax;
                    }
                    ++c2;
                }
                if (!(score > 0.0f)) qmin = qmax = rmin = rmax = 0;
                const int dq = qmax - qmin, dr = rmax - rmin;
                const float ave = __fdiv_rn((float)(dr + dq), 2.0f);
                bool ok = (ave != 0.0f) && (__fdiv_rn(score, ave) > a.prm.min_sim) &&
                          ((dr < dq ? dr : dq) > a.prm.min_length);
                if (ok && nbox > 0) {
                    float mxi = -INFINITY;
                    for (int kx = 0; kx < nbox; ++kx) {
                        const int* o = boxes + 4 * kx;
                        const float lt0 = (float)(qmin > o[0] ? qmin : o[0]);
                        const float lt1 = (float)(rmin > o[1] ? rmin : o[1]);
                        const float rb0 = (float)(qmax < o[2] ? qmax : o[2]);
                        const float rb1 = (float)(rmax < o[3] ? rmax : o[3]);
                        const float w = rb0 - lt0 > 0.0f ? rb0 - lt0 : 0.0f;
                        const float h = rb1 - lt1 > 0.0f ? rb1 - lt1 : 0.0f;
                        const float inter = __fmul_rn(w, h);
                        const float aa2 = __fmul_rn((float)dq, (float)dr);
                        const float ab = __fmul_rn((float)(o[2] - o[0]), (float)(o[3] - o[1]));
                        const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(aa2, ab), inter));
                        mxi = iou > mxi ? iou : mxi;
                    }
                    ok = mxi < a.prm.max_iou;
                }
                if (ok && nbox < VSC_TN_MAX_BOXES) {
                    boxes[4 * nbox + 0] = qmin;
                    boxes[4 * nbox + 1] = rmin;
                    boxes[4 * nbox + 2] = qmax;
                    boxes[4 * nbox + 3] = rmax;
                    accepted = 1;
                    bq0 = qmin; br0 = rmin; bq1 = qmax; br1 = rmax;
                }
            }
            // indeg[] was used as scratch: a later lazy Kahn run recomputes it (have_order guards
            // only `order`, which is untouched here)
        }
        stop = __shfl(stop, 0);
        accepted = __shfl(accepted, 0);
        __syncthreads();
        TN_T(5);
        if (stop) break;
        if (accepted) {
            bq0 = __shfl(bq0, 0); br0 = __shfl(br0, 0); bq1 = __shfl(bq1, 0); br1 = __shfl(br1, 0);
            // MaxSim box score: max of sims[q_lo:q_hi, r_lo:r_hi] (half-open, localization.py:91) - bias
            float m = -INFINITY;
            const int w = br1 - br0, h = bq1 - bq0;
            for (int y = 0; y < h; ++y) {  // (row by row: no 64-bit divisions per element)
                const float* srow = sims + (int64_t)(bq0 + y) * lr + br0;
                for (int x = lane; x < w; x += 64) {
                    const float s = srow[x];
                    m = s > m ? s : m;
                }
            }
#pragma unroll
            for (int off2 = 32; off2 >= 1; off2 >>= 1) {
                const float o2 = __shfl_xor(m, off2);
                m = o2 > m ? o2 : m;
            }
            if (lane == 0) {
                o_boxes[4 * nbox + 0] = bq0;
                o_boxes[4 * nbox + 1] = br0;
                o_boxes[4 * nbox + 2] = bq1;
                o_boxes[4 * nbox + 3] = br1;
                o_bmax[nbox] = __fsub_rn(m, a.bias);
            }
            ++nbox;
        }
        TN_T(6);
    }
    if (lane == 0) *o_nbox = nbox;
}

#ifdef VSC_TN_PROFILE
extern "C" int vsc_tn_prof_read(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(tn_prof), sizeof(tn_prof)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(tn_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

size_t tn_state_bytes_host(int max_lq, int top_cap, int ms, int idx_bytes) {
    const size_t n_nodes = 1 + (size_t)max_lq * top_cap;
    const size_t ib = (size_t)idx_bytes;
    size_t b = 0;
    b += (size_t)max_lq * top_cap * ib;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)max_lq * top_cap * 4;
    b += (size_t)max_lq * ms * ib * 2;
    b = (b + 15) & ~(size_t)15;
    b += ((size_t)max_lq * top_cap * ms * top_cap + 31) / 32 * 4;
    b = (b + 15) & ~(size_t)15;
    b += n_nodes * 4;
    b += n_nodes * ib * 3;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)VSC_TN_MAX_BOXES * 16;
    b += (size_t)max_lq * ((top_cap + 1) / 2) * 8;  // okmask
    b = (b + 63) & ~(size_t)63;
    return b;
}

int launch_tn_pairs(const TnPairArgs& a, size_t lds_bytes, hipStream_t stream) {
    if (a.n_work <= 0) return VSC_OK;
    static PerDeviceOnce once;
    if (a.state) {  // over-long videos: state in the HBM slab, 32-bit indices
        hipLaunchKernelGGL((tn_pair_kernel<int, true>), dim3((unsigned)a.n_work), dim3(64), 0, stream, a);
        VSC_HIP(hipGetLastError());
        return VSC_OK;
    }
    if (once.first()) {
        VSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tn_pair_kernel<short, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        once.commit();
    }
    hipLaunchKernelGGL((tn_pair_kernel<short, false>), dim3((unsigned)a.n_work), dim3(64), lds_bytes, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

// LocalizationWithMetadata.similarity (vsc/baseline/localization.py:33-36,48-54) for one pair:
// out[lq][lr] = q.feature @ r.feature.T + bias, one 32x32 block per wavefront.

__global__ __launch_bounds__(64) void tn_sims_kernel(TnSimsArgs a) {
    const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31;
    const int qb = blockIdx.y * 32, rb = blockIdx.x * 32;
    const float* ap = a.qfeat + (a.qrow0 + qb + l31) * a.dpad + hi * 4;
    const float* bp = a.rfeat + (a.rrow0 + rb + l31) * a.dpad + hi * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int nkg = a.dpad / 8;
#pragma unroll 4
    for (int g = 0; g < nkg; ++g) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + g * 8);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + g * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = qb + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int rr = rb + l31;
        if (q < a.lq && rr < a.lr) a.out[(int64_t)q * a.lr + rr] = acc[r] + a.bias;
    }
}

int launch_tn_sims(const TnSimsArgs& a, hipStream_t stream) {
    if (a.lq <= 0 || a.lr <= 0) return VSC_OK;
    hipLaunchKernelGGL(tn_sims_kernel, dim3((a.lr + 31) / 32, (a.lq + 31) / 32), dim3(64), 0, stream, a);
    VSC_HIP(hipGetLastError());
    return VSC_OK;
}

}  // namespace vscmi
