// Fused (flash-style) attention forward for the SD1.5 head sizes 40 / 80 / 160 (+ 128 FLUX, + 64 causal CLIP) on gfx950.
//
//   out[b, q, h, :] = softmax(scale * Q K^T) V        (no mask; self and cross attention)
//
// Design (wave64 / MFMA 16x16x32 f16, not a warp-shaped port):
//   * workgroup = 4 waves; each wave owns QT*16 query rows of one (batch, head) and walks the keys in
//     tiles of 64; K/V tiles are shared through LDS (register-staged, double buffered, one barrier
//     per tile: loads for tile t+1 are issued before tile t's MFMAs and written after them);
//   * scores are computed TRANSPOSED, S^T = K Q^T (K = MFMA A operand, Q = B operand), so every lane
//     owns one query column: the online-softmax max/sum are register reductions + 2 shuffles, and the
//     exponentiated tile is already laid out as the B operand of the second product
//     O^T = V^T P^T  -- P never touches LDS;
//   * V^T fragments come from the row-major V tile with the gfx950 transposing LDS read
//     (ds_read_b64_tr_b16); row strides (odd multiples of 32 B, or XOR-swizzled 128-B rows) keep both the ds_read_b128 K reads
//     and the transposed V reads bank-conflict free under gfx950's lane grouping;
//   * the MFMA k-index <-> key permutation induced by reusing the accumulator layout as an operand
//     (keys 4g..4g+3 and 16+4g..16+4g+3 per lane group g) is applied identically to the V reads;
//   * head dims are zero-padded in registers/LDS only (40 -> 64 for Q K^T, 40 -> 48 for P V);
//   * O^T leaves each lane with 4 consecutive channels of one query row -> 8-byte stores.
#include "ops.h"
#include "el.h"
#include <type_traits>


namespace {

struct AttnParams {
    const f16* q; const f16* k; const f16* v; f16* out;
    int q_stride, k_stride, v_stride, out_stride;
    int H, Nq, Nk;
    float c;   // scale * log2(e)
    const float* bias;   // [H][Nq][Nk] fp32, times log2(e), or null
    // workgroup -> (query block, head, batch): lin = lin0 + blockIdx.x, query block fastest (the order a 3-D grid is dispatched in)
    int nqb, lin0;
    // split-KV tail (SPLIT kernels): blockIdx.y = key range, results go to the fp32 partial buffers instead of `out`
    int splits, tiles_per_split, part_rows;
    float* part_o;       // [splits][part_rows][DH] unnormalised O, relative to the partial's own reference maximum
    float* part_ml;      // [splits][part_rows][2]  (reference maximum in log2 units, denominator)
    int prio;            // 1: the wave in the odd hardware slot of its SIMD runs at raised priority (see the kernel)
};

typedef __fp16 hf4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) hf4* lds_hf4_ptr;

__device__ __forceinline__ u32x2 tr_read(const char* lds_addr) {      // 4 x 16-bit, type agnostic
    hf4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hf4_ptr)lds_addr);
    union { hf4 a; u32x2 b; } u; u.a = r;
    return u.b;
}

// max over the lanes {l, l^16} / {l, l^32} with the gfx950 permlane swaps (pure VALU, no LDS round trip)
__device__ __forceinline__ float xor16_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <typename T, int DH, int QT, bool CAUSAL, bool BIAS, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnParams p) {
    typedef typename El<T>::frag frag;
    constexpr int DK = (DH + 31) / 32 * 32;
    constexpr int KSTEPS = DK / 32;
    constexpr int DVT = (DH + 15) / 16;
    constexpr int DVP = DVT * 16;
    constexpr bool KSWZ = (DK == 64);                                 // 128-B rows: XOR-swizzled chunks, conflict-free b128 reads
    // bytes per K row in LDS.  ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), so
    // with the lane -> (row i16, 16-byte chunk g) map of the K fragments a row stride of 16 B mod 32 B puts two lanes of every group on one bank
    // (measured: SQ_LDS_BANK_CONFLICT = 25 % of the LDS cycles at dh 128); an odd multiple of 32 B is conflict free (simulated per group and
    // confirmed with the counter), the same rule as for the transposed V reads.
    constexpr int KS = KSWZ ? 128 : ((DK * 2) % 64 == 32 ? DK * 2 : DK * 2 + 32);
    constexpr int VS = ((DVP * 2) % 64 == 32) ? DVP * 2 : DVP * 2 + 32;  // odd multiple of 32 B
    // spare padded V column (dh = 40 -> 48): a column of ones makes the P V MFMA also produce the softmax
    // denominator (row DH of O^T), replacing 16*QT v_add_f32 per tile per lane and the separate l rescale
    constexpr bool ONES = (DVP > DH);
    constexpr int CPR = DH / 8;                                       // 16-byte chunks per K/V row
    constexpr int NCH = (64 * CPR + 255) / 256;
    constexpr int KBUF = 64 * KS, VBUF = 64 * VS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lK = smem;               // [2][KBUF]
    char* const lV = smem + 2 * KBUF;    // [2][VBUF]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    // Two workgroups share a CU, so every SIMD hosts two waves that run this same loop (score MFMAs -> exp / convert VALU -> P V MFMAs ->
    // barrier).  With equal priority their MFMA phases share the matrix pipe 1:1 and their VALU phases the issue port, so both finish
    // every phase late; a static priority for ONE of the two (the wave in the odd hardware wave slot, HW_REG_HW_ID[3:0]) lets its MFMA
    // phase run uncontested while the partner is in its VALU phase and vice versa -- the effect measured on the conv kernel's wave groups.
    if (p.prio) {
        const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID (id 4), bits [3:0] = wave slot in the SIMD
        if (slot & 1) __builtin_amdgcn_s_setprio(2);
    }
    // Workgroup id -> XCD is id % 8 (round-robin dispatch); give every XCD a CONTIGUOUS range of (query block, head) items so that the ~64
    // workgroups resident on one XCD walk the same head's K/V at the same time and share it through that XCD's 4 MB L2 (in plain order an XCD
    // sees every 8th query block of every head: 8x the K/V bytes through its L2; FLUX: 4.45 MB of K/V per head, 1632 workgroups).
    int lin;
    {
        const int n = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = n >> 3, r = n & 7;
        lin = p.lin0 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int qblk = lin % p.nqb, hb = lin / p.nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int q0 = qblk * (4 * QT * 16) + w * (QT * 16);
    // key range of this workgroup: everything, or (SPLIT) tiles [blockIdx.y * tiles_per_split, ...)
    const int key0 = SPLIT ? blockIdx.y * p.tiles_per_split * 64 : 0;
    const int Nk = SPLIT ? min(p.Nk - key0, p.tiles_per_split * 64) : p.Nk;

    // zero both buffers once: pad columns stay zero, tails are rewritten with zeros explicitly
    // (with the swizzled K layout the pad chunks 5..7 of a row land in permuted slots: still never written)
    for (int o = tid * 16; o < 2 * KBUF + 2 * VBUF; o += 256 * 16) *reinterpret_cast<u32x4*>(smem + o) = u32x4{0, 0, 0, 0};

    // ---- Q fragments (B operand: lane = query column i16, k = 8g + j) ---------------------------
    // (all QT * KSTEPS loads issued together, branch-free: rows / head dims past the end read a clamped address and are zeroed afterwards.  Under
    //  `if (qrow < p.Nq && d < DH)` every load came out as load, s_waitcnt vmcnt(0), convert: 5-8 memory round trips one behind the other per workgroup)
    frag qf[QT][KSTEPS];
    {
        u32x4 raw[QT][KSTEPS];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int qrow = q0 + t * 16 + i16, qr = qrow < p.Nq ? qrow : p.Nq - 1;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int d = ks * 32 + 8 * g, dc = d < DH ? d : 0;
                raw[t][ks] = *reinterpret_cast<const u32x4*>(p.q + ((size_t)(b * p.Nq + qr) * p.q_stride + h * DH + dc));
            }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int qrow = q0 + t * 16 + i16;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int d = ks * 32 + 8 * g;
                const bool ok = qrow < p.Nq && d < DH;
                u32x4 v = raw[t][ks];
#pragma unroll
                for (int e = 0; e < 4; ++e)      // fold scale*log2(e) into Q
                    v[e] = ok ? pack2<T>(El<T>::tof((u16)(v[e] & 0xffff)) * p.c, El<T>::tof((u16)(v[e] >> 16)) * p.c) : 0u;
                qf[t][ks] = as_frag<T>(v);
            }
        }
    }

    f32x4 o_acc[DVT][QT];
#pragma unroll
    for (int a = 0; a < DVT; ++a)
#pragma unroll
        for (int t = 0; t < QT; ++t) o_acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { m_run[t] = 0.f; l_run[t] = 0.f; }

    const int ntiles = (Nk + 63) / 64;
    const f16* kbase = p.k + ((size_t)b * p.Nk + key0) * p.k_stride + h * DH;
    const f16* vbase = p.v + ((size_t)b * p.Nk + key0) * p.v_stride + h * DH;

    u32x4 kreg[NCH], vreg[NCH];
    // per-thread staging slots (loop invariant): chunk id -> (row, 16-byte chunk) of the 64-key tile.  When the 256 threads cover whole rows
    // (dh 64 / 128) slot c is slot 0 moved down by c * RPC rows, so only slot 0's offsets are kept in registers.
    constexpr bool REG = (256 % CPR == 0) && (NCH * 256 == 64 * CPR);
    constexpr int RPC = 256 / CPR;
    int st_row_[NCH], st_koff_[NCH], st_voff_[NCH], st_lk_[NCH], st_lv_[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int id = tid + c * 256;
        const int row = id / CPR, ch = id - row * CPR;
        st_row_[c] = (id < 64 * CPR) ? row : (1 << 20);            // slots past the tile never pass a bounds test
        st_koff_[c] = (id < 64 * CPR) ? row * p.k_stride + ch * 8 : 0;   // idle slots re-read key 0 (never stored)
        st_voff_[c] = (id < 64 * CPR) ? row * p.v_stride + ch * 8 : 0;
        st_lk_[c] = row * KS + (KSWZ ? (ch ^ (row & 7)) : ch) * 16;
        st_lv_[c] = row * VS + ch * 16;
    }
    auto st_row = [&](int c) { return REG ? st_row_[0] + c * RPC : st_row_[c]; };
    auto st_koff = [&](int c) { return REG ? st_koff_[0] + c * RPC * p.k_stride : st_koff_[c]; };
    auto st_voff = [&](int c) { return REG ? st_voff_[0] + c * RPC * p.v_stride : st_voff_[c]; };
    auto st_lk = [&](int c) { return REG ? st_lk_[0] + c * RPC * KS : st_lk_[c]; };
    auto st_lv = [&](int c) { return REG ? st_lv_[0] + c * RPC * VS : st_lv_[c]; };
    auto load_tile = [&](int tile) {
        const f16* kt_base = kbase + (size_t)tile * 64 * p.k_stride;
        const f16* vt_base = vbase + (size_t)tile * 64 * p.v_stride;
        const int rows_left = Nk - tile * 64;                   // >= 64 for every tile but a ragged last one
        if (rows_left >= 64) {                                    // wave-uniform fast path: no per-lane guards
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                kreg[c] = *reinterpret_cast<const u32x4*>(kt_base + st_koff(c));
                vreg[c] = *reinterpret_cast<const u32x4*>(vt_base + st_voff(c));
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                u32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
                if (st_row(c) < rows_left) {
                    kv = *reinterpret_cast<const u32x4*>(kt_base + st_koff(c));
                    vv = *reinterpret_cast<const u32x4*>(vt_base + st_voff(c));
                }
                kreg[c] = kv; vreg[c] = vv;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (st_row(c) < 64) {
                *reinterpret_cast<u32x4*>(lK + buf * KBUF + st_lk(c)) = kreg[c];
                *reinterpret_cast<u32x4*>(lV + buf * VBUF + st_lv(c)) = vreg[c];
            }
        }
    };

    load_tile(0);
    __syncthreads();            // zero fill complete
    if (ONES) {
        for (int r = tid; r < 128; r += 256) *reinterpret_cast<u16*>(lV + (r >> 6) * VBUF + (r & 63) * VS + DH * 2) = El<T>::fromf(1.0f);
    }
    store_tile(0);
    __syncthreads();

    // Online softmax with the running reference max folded into the MFMA accumulator: Q is pre-scaled by
    // c = scale*log2(e), and the score MFMA chain starts from C = -m_ref[q] (one register quad per query
    // tile, constant along the key rows), so the accumulator already holds the exponent S' = c q.k - m_ref.
    // m_ref is only moved when some column's tile maximum exceeds it by more than THR (= 2^8 headroom in the
    // fp16 P values; exactness is unaffected because P and the denominator share the same reference), which
    // is a rare, wave-uniform slow path: the steady state is max3 / v_exp / cvt only.
    constexpr float THR = 8.0f;
    auto do_tile = [&](int tile, auto ragged_tag, auto first_tag, auto fast_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;       // no per-tile maxima after the first tile (see the driver below)
        const int buf = tile & 1;
        if (tile + 1 < ntiles) load_tile(tile + 1);
        const char* tk = lK + buf * KBUF;
        const char* tv = lV + buf * VBUF;

        // ---- S' = c K Q^T - m_ref : s[kt][t] holds keys kt*16 + 4g + r, query column i16 -------------
        f32x4 s[4][QT];
        f32x4 negm[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) { const float v = FIRST ? 0.f : -m_run[t]; negm[t] = f32x4{v, v, v, v}; }
        // K fragments run RING reads ahead of the MFMAs that consume them (the compiler's own order is read -> wait -> MFMA: every group of
        // QT MFMAs then waits out a full LDS latency)
        {
            constexpr int NF = 4 * KSTEPS, RING = NF < 4 ? NF : 4;
            auto kaddr = [&](int f) {
                const int kt = f / KSTEPS, ks = f - kt * KSTEPS;
                return tk + (kt * 16 + i16) * KS + (KSWZ ? ((ks * 4 + g) ^ (i16 & 7)) * 16 : (ks * 32 + 8 * g) * 2);
            };
            frag kf[RING];
#pragma unroll
            for (int f = 0; f < RING; ++f) kf[f] = *reinterpret_cast<const frag*>(kaddr(f));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int kt = f / KSTEPS, ks = f - kt * KSTEPS;
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    s[kt][t] = El<T>::mfma(kf[f % RING], qf[t][ks], ks == 0 ? negm[t] : s[kt][t]);
                if (f + RING < NF) kf[f % RING] = *reinterpret_cast<const frag*>(kaddr(f + RING));
                __builtin_amdgcn_sched_barrier(0);          // pin the program order: the scheduler would sink the read next to its use
            }
        }
        if (BIAS) {     // T5 relative-position bias: four consecutive keys of one query per load
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int qi = q0 + t * 16 + i16;
                const float* brow = p.bias + ((size_t)h * p.Nq + (qi < p.Nq ? qi : 0)) * p.Nk + tile * 64 + 4 * g;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    if (tile * 64 + kt * 16 + 4 * g + 3 < p.Nk) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(brow + kt * 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[kt][t][r] += bv[r];
                    }
                }
            }
        }
        if (CAUSAL) {   // text-encoder attention: key j is visible to query i iff j <= i (Nq == Nk)
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int qi = q0 + t * 16 + i16;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (tile * 64 + kt * 16 + 4 * g + r > qi) s[kt][t][r] = -INFINITY;
            }
        }
        if (RAGGED) {   // ragged last tile: mask keys >= Nk (compiled only into the peeled last iteration)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (tile * 64 + kt * 16 + 4 * g + r >= Nk) {
#pragma unroll
                        for (int t = 0; t < QT; ++t) s[kt][t][r] = -INFINITY;
                    }
        }

        // ---- per-lane tile maxima; move the reference only when needed (wave-uniform decision) ---------
        if (FIRST || !FAST) {
        float mxl[QT];
        bool over = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = s[0][t][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][t][r]);
            mxl[t] = mx;
            over |= mx > THR;
        }
        if (FIRST || __builtin_amdgcn_ballot_w64(over) != 0) {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const float mx = xor32_max(xor16_max(mxl[t]));         // column maximum over all 64 keys
                const float delta = FIRST ? mx : fmaxf(mx, 0.f);
                m_run[t] = FIRST ? delta : m_run[t] + delta;
                if (!FIRST) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    if (!ONES) l_run[t] *= alpha;
#pragma unroll
                    for (int a = 0; a < DVT; ++a) o_acc[a][t] *= alpha;
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][t][r] -= delta;
            }
        }
        }

        // ---- P = exp2(S') -> fp16 fragments of the second product ------------------------------------------
        frag pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[kt][t][r]);     // raw v_exp_f32
                    s[kt][t][r] = e;
                    if (!ONES) ps += e;
                }
            if (!ONES) l_run[t] += ps;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                u32x4 f;
                f[0] = pack2<T>(s[2 * t2][t][0], s[2 * t2][t][1]); f[1] = pack2<T>(s[2 * t2][t][2], s[2 * t2][t][3]);
                f[2] = pack2<T>(s[2 * t2 + 1][t][0], s[2 * t2 + 1][t][1]); f[3] = pack2<T>(s[2 * t2 + 1][t][2], s[2 * t2 + 1][t][3]);
                pf[t][t2] = as_frag<T>(f);
            }
        }

        // ---- O^T += V^T P^T ------------------------------------------------------------------------
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int a = 0; a < DVT; ++a) {
                const char* addr = tv + (32 * t2 + 4 * g + (i16 >> 2)) * VS + (a * 16 + 4 * (i16 & 3)) * 2;
                const u32x2 lo = tr_read(addr);
                const u32x2 hi = tr_read(addr + 16 * VS);
                const frag vf = as_frag<T>(u32x4{lo[0], lo[1], hi[0], hi[1]});
#pragma unroll
                for (int t = 0; t < QT; ++t) o_acc[a][t] = El<T>::mfma(vf, pf[t][t2], o_acc[a][t]);
            }
        }

        if (tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    };
    const int full_tiles = Nk / 64;
    auto run_tiles = [&](auto fast_tag) {
        if (full_tiles > 0) do_tile(0, std::false_type{}, std::true_type{}, fast_tag);
        else do_tile(0, std::true_type{}, std::true_type{}, fast_tag);
        for (int tile = 1; tile < full_tiles; ++tile) do_tile(tile, std::false_type{}, std::false_type{}, fast_tag);
        if (full_tiles > 0 && full_tiles < ntiles) do_tile(full_tiles, std::true_type{}, std::false_type{}, fast_tag);
    };
    // Head dim 40 is issue-bound on the softmax VALU work (64 v_exp + 58 v_max per 56 MFMAs per tile), so its steady state drops
    // the maxima: every tile is exponentiated against the FIRST tile's column maximum.  That is exact as long as no later score
    // exceeds it by 2^16 (the fp16 range of P); if one does, P holds an inf, the ones-column denominator comes out non-finite,
    // and the whole workgroup redoes its rows with the maxima-tracking loop (block-uniform decision, never seen on real inputs).
    // Head dim 128 (FLUX) does the same (VALU work per tile exceeds the MFMA time there too); without a spare V column the overflow shows up
    // as a non-finite output accumulator (inf P times V) or denominator, so all of them are checked once at the end.
    constexpr bool FASTABLE = ONES || (DH == 128 && !CAUSAL && !BIAS);
    if constexpr (FASTABLE) {
        run_tiles(std::true_type{});
        bool bad = false;
        if constexpr (ONES) {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const float l = __shfl(o_acc[DH / 16][t][(DH % 16) % 4], ((DH % 16) / 4) * 16 + i16, 64);
                bad |= !(l < INFINITY);                      // inf or NaN
            }
        } else {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float chk = l_run[t];
#pragma unroll
                for (int a = 0; a < DVT; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) chk += fabsf(o_acc[a][t][r]);
                bad |= !(chk < INFINITY);                    // any inf or NaN among the accumulators or the denominator
            }
        }
        if (__syncthreads_or(bad ? 1 : 0)) {
#pragma unroll
            for (int a = 0; a < DVT; ++a)
#pragma unroll
                for (int t = 0; t < QT; ++t) o_acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < QT; ++t) { m_run[t] = 0.f; l_run[t] = 0.f; }
            load_tile(0);
            store_tile(0);
            __syncthreads();
            run_tiles(std::false_type{});
        }
    } else {
        run_tiles(std::false_type{});
    }

    // ---- epilogue ---------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l;
        if (ONES) {
            // row DH of O^T = sum_k P[k][q]: tile a = DH/16, lane group g = (DH%16)/4, register 0
            l = __shfl(o_acc[DH / 16][t][(DH % 16) % 4], ((DH % 16) / 4) * 16 + i16, 64);
        } else {
            l = l_run[t];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        if constexpr (SPLIT) {
            // partial result of this key range: O unnormalised (fp32), its reference maximum and denominator
            const int prow = (lin - p.lin0) * (4 * QT * 16) + w * (QT * 16) + t * 16 + i16;
            float* po = p.part_o + ((size_t)blockIdx.y * p.part_rows + prow) * DH;
#pragma unroll
            for (int a = 0; a < DVT; ++a) *reinterpret_cast<f32x4*>(po + a * 16 + 4 * g) = o_acc[a][t];
            if (g == 0) { float* pm = p.part_ml + ((size_t)blockIdx.y * p.part_rows + prow) * 2; pm[0] = m_run[t]; pm[1] = l; }
            continue;
        }
        const float inv = 1.0f / l;
        const int qrow = q0 + t * 16 + i16;
        if (qrow >= p.Nq) continue;
        f16* orow = p.out + (size_t)(b * p.Nq + qrow) * p.out_stride + h * DH;
#pragma unroll
        for (int a = 0; a < DVT; ++a) {
            const int d = a * 16 + 4 * g;
            if (d + 4 <= DH) {
                const u32x2 o = {pack2<T>(o_acc[a][t][0] * inv, o_acc[a][t][1] * inv), pack2<T>(o_acc[a][t][2] * inv, o_acc[a][t][3] * inv)};
                *reinterpret_cast<u32x2*>(orow + d) = o;
            }
        }
    }
}

// out[row] = sum_i 2^(m_i - m) O_i / sum_i 2^(m_i - m) l_i over the key ranges of the split-KV tail; one thread per 8 channels
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnParams p, int rows_per_wg) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int prow = idx / (DH / 8), d = (idx - prow * (DH / 8)) * 8;
    if (prow >= p.part_rows) return;
    const int lin = p.lin0 + prow / rows_per_wg;
    const int qblk = lin % p.nqb, hb = lin / p.nqb, h = hb % p.H, b = hb / p.H;
    const int qrow = qblk * rows_per_wg + prow % rows_per_wg;
    if (qrow >= p.Nq) return;
    float m = -INFINITY;
    for (int i = 0; i < p.splits; ++i) m = fmaxf(m, p.part_ml[((size_t)i * p.part_rows + prow) * 2]);
    float l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < p.splits; ++i) {
        const float* pm = p.part_ml + ((size_t)i * p.part_rows + prow) * 2;
        const float sc = __builtin_amdgcn_exp2f(pm[0] - m);
        l += sc * pm[1];
        const float* po = p.part_o + ((size_t)i * p.part_rows + prow) * DH + d;
        const f32x4 a = *reinterpret_cast<const f32x4*>(po), c = *reinterpret_cast<const f32x4*>(po + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] += sc * a[k]; o[4 + k] += sc * c[k]; }
    }
    const float inv = 1.0f / l;
    u32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = pack2<T>(o[2 * k] * inv, o[2 * k + 1] * inv);
    *reinterpret_cast<u32x4*>(p.out + (size_t)(b * p.Nq + qrow) * p.out_stride + h * DH + d) = r;
}

// ------------------------------------------------------------------------------------------------
// Head dim 40 self-attention with DEDICATED LOADER WAVES and a hand-placed instruction stream (round 3).
// attn_kernel<f16, 40, 4> spends ~2000 SIMD cycles per 64-key tile and wave (45 % matrix-pipe busy): two waves per SIMD each run scores -> softmax -> P V
// one after the other, the compiler waits out LDS latencies inside both MFMA phases, and all four waves meet at a barrier per tile.  The tile's work
// is 56 MFMAs (896 matrix-pipe cycles) and ~1300 cycles of vector issue (MFMA 8, v_exp_f32 8, everything else 4: MI355X_MICROARCH.md), so what a SIMD can
// do is bounded by the issue port, and only a stream that never stalls gets near it.  This kernel is conv3_lw_kernel's answer applied to attention:
//   * 8 waves, one workgroup per CU: waves 0-3 (one per SIMD, 64 queries each) compute and never touch global memory inside the loop, waves 4-7 stage K / V
//     tiles by LDS-DMA into a ring of NS stages (row stride 96 B for both: conflict-free ds_read_b128 and transposing reads; K's pad chunk is a copy of
//     real data and meets zero Q columns, V's pad chunk is (1, 0, ..) -- the ones column that makes the P V product produce the softmax denominator);
//   * the compute stream is software-pipelined over UNITS of 32 keys: slot n issues the 16 score MFMAs of unit n, the 32 exponentials + 16 conversions of
//     unit n - 1 and the 12 P V MFMAs of unit n - 2 interleaved (three mutually independent strands, so nothing in a slot waits for anything in it), every
//     operand fragment is read one slot ahead with immediate offsets and a hand-counted lgkmcnt; one barrier per 64-key tile, none of the waves waits at it
//     for LDS;
//   * the reference maximum of the softmax is the first tile's column maximum (as in attn_kernel's FAST path); a wave whose denominators come out non-finite
//     redoes its 64 rows alone with a plain maxima-tracking loop straight from global memory (never seen on real inputs; tests force it).
// With K16 = false bit-identical to attn_kernel<f16, 40, 4> (same MFMA chains, same exponent arguments, same rounding, same accumulation order); the default K16 form sums
// head dims 32..39 in a 16x16x16 MFMA (another association of the same products).
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned g_attn_ones_chunk[8] = {0x00003c00u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};     // fp16 (1, 0, 0, 0, 0, 0, 0, 0) | eight zeros

typedef const __attribute__((address_space(1))) void* a40_gptr;
typedef __attribute__((address_space(3))) void* a40_lptr;
__device__ __forceinline__ void glds16a(const void* src, void* lds_wave_base) { __builtin_amdgcn_global_load_lds((a40_gptr)src, (a40_lptr)lds_wave_base, 16, 0, 0); }
template <int OFF>
__device__ __forceinline__ void a40_read_k(f16x8& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void a40_read_vt(u32x2& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void a40_read_k64(u32x2& d, unsigned addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
__device__ __forceinline__ void a40_mfma16(f32x4& c, const u32x2& a, const u32x2& b) { asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
template <int N>
__device__ __forceinline__ void a40_wait64(u32x2& d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
template <int N>
__device__ __forceinline__ void a40_wait(f16x8& d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
template <int N>
__device__ __forceinline__ void a40_wait2(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
__device__ __forceinline__ void a40_mfma(f32x4& c, const f16x8& a, const f16x8& b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void a40_mfma_c(f32x4& d, const f16x8& a, const f16x8& b, const f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void a40_mfma_z(f32x4& d, const f16x8& a, const f16x8& b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b)); }
// (macros: a vector element cannot bind to a reference)
#define a40_exp(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define a40_cvt(d, a, b) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
template <int N, class F, int... I>
__device__ __forceinline__ void a40_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void a40_for(F&& f) { a40_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// timing experiments only (cs_set_tuning("debug", 16384) selects the TRACE instantiation; cs_debug_attn_trace_read): per workgroup
// [0] shader cycles of compute wave 0 over the steady loop, [1] 100 MHz ticks over the same, [2] tiles in it, [3] loader wave 4's cycles in vmcnt waits, [4] at barriers
#define A40_TRACE_SLOTS 4096
#define A40_TRACE_W 12
__device__ unsigned long long g_a40_trace[A40_TRACE_SLOTS * A40_TRACE_W];

// DBG (TRACE builds only, results wrong): 1 steady loop without the exponentials / conversions, 2 without the MFMAs
// K16: the second k step of the scores (head dims 32..39 of 40) as v_mfma_f32_16x16x16_f16 on 64-bit fragments (head dims 32..47; K's pad chunk is zeros then)
// instead of a 16x16x32 whose upper 24 k are padding: the same instruction count and matrix-pipe time, half the multiplies of a quarter of the MFMAs -- under the
// board's power limit that is clock (profiles/r03_probe_exp.txt, mode 6).  The sums associate differently: not bit-identical to attn_kernel any more.
template <int NS, bool TRACE = false, int DBG = 0, int PAD = 0, bool K16 = true>
__global__ __launch_bounds__(512) void attn40_lw_kernel(AttnParams p) {
    unsigned long long te0 = 0;
    if (TRACE) te0 = __builtin_amdgcn_s_memrealtime();
    constexpr int DH = 40, RS = 96, KB = 64 * RS, ST = 2 * KB;                  // stage = K tile (6 KB) | V tile (6 KB)
    static_assert(NS >= 5 && NS <= 12, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    int lin;
    {
        const int n = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = n >> 3, r = n & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int qblk = lin % p.nqb, hb = lin / p.nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int ntiles = p.Nk >> 6;
    const f16* kbase = p.k + (size_t)b * p.Nk * p.k_stride + h * DH;
    const f16* vbase = p.v + (size_t)b * p.Nk * p.v_stride + h * DH;

    // (no LDS initialisation: every byte of a stage is written by its twelve DMA pieces -- the pad chunk of a K row is a copy of real data, the pad chunk of a V row
    //  is the ones chunk -- and fragments are only multiplied out of stages that have landed)
    if (w >= 4) {
        // =============================== loader waves: pieces l, l + 4, l + 8 of the 12 KiB stage ===============================
        const int l = w - 4;
        const char* src[3]; size_t step[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = l + 4 * j, kv = q >= 6 ? 1 : 0, pq = q - 6 * kv;
            const int off = pq * 1024 + lane * 16, row = off / RS, ch = (off - row * RS) >> 4;
            if (!kv && ch == 5 && K16) { src[j] = reinterpret_cast<const char*>(g_attn_ones_chunk + 4); step[j] = 0; }        // (zeros: the 16x16x16 step multiplies head dims 40..47)
            else if (!kv) { src[j] = reinterpret_cast<const char*>(kbase + (size_t)row * p.k_stride + (ch == 5 ? 0 : ch) * 8); step[j] = (size_t)64 * p.k_stride * 2; }
            else if (ch < 5) { src[j] = reinterpret_cast<const char*>(vbase + (size_t)row * p.v_stride + ch * 8); step[j] = (size_t)64 * p.v_stride * 2; }
            else { src[j] = reinterpret_cast<const char*>(g_attn_ones_chunk); step[j] = 0; }
        }
        auto issue = [&](int tile, int stage) {
            const int t = tile < ntiles ? tile : ntiles - 1;                   // past the end: the last tile again, into a free stage (keeps the counts uniform)
#pragma unroll
            for (int j = 0; j < 3; ++j) glds16a(src[j] + (size_t)t * step[j], smem + stage * ST + (l + 4 * j) * 1024);
        };
        for (int t = 0; t <= NS - 3; ++t) issue(t, t);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (NS - 4)) : "memory");     // tiles 0 and 1 have landed
        __builtin_amdgcn_s_barrier();                                            // B(-1)
        int st = NS - 2;
        unsigned long long tl_wait = 0, tl_bar = 0;
        for (int j = 0; j < ntiles; ++j) {
            issue(j - 2 + NS, st);                                               // into the stage of tile j - 2, released at B(j - 1)
            st = st + 1 == NS ? 0 : st + 1;
            unsigned long long s0 = 0, s1 = 0;
            if (TRACE) s0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (NS - 4)) : "memory"); // tile j + 2 has landed (first read behind B(j))
            if (TRACE) s1 = __builtin_readcyclecounter();
            __builtin_amdgcn_s_barrier();                                        // B(j)
            if (TRACE) { const unsigned long long s2 = __builtin_readcyclecounter(); tl_wait += s1 - s0; tl_bar += s2 - s1; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TRACE && l == 0 && lane == 0 && blockIdx.x < A40_TRACE_SLOTS) { g_a40_trace[blockIdx.x * A40_TRACE_W + 3] = tl_wait; g_a40_trace[blockIdx.x * A40_TRACE_W + 4] = tl_bar; }
        return;
    }

    // =============================== compute waves ===============================
    if (p.prio) __builtin_amdgcn_s_setprio(1);                 // (cs_set_tuning("attn_prio", 1): static priority over the loader partner; measured)
    const int q0 = qblk * 256 + w * 64;
    f16x8 qf[4][2];                     // B operands of the score MFMAs: head dims 0..31 (and, without K16, 32..63 with 40.. zero)
    u32x2 qh[4];                        // K16: head dims 32 + 4 g .. + 3 as the 16x16x16 step's B operand (g >= 2: zero)
    auto load_qf = [&](int t, int ks) {                                          // Q fragment pre-scaled by scale * log2(e)
        const int qrow = q0 + t * 16 + i16, d = ks * 32 + 8 * g;
        u32x4 v = {0, 0, 0, 0};
        if (d < DH) {
            v = *reinterpret_cast<const u32x4*>(p.q + ((size_t)(b * p.Nq + qrow) * p.q_stride + h * DH + d));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = pack2<f16>(El<f16>::tof((u16)(v[e] & 0xffff)) * p.c, El<f16>::tof((u16)(v[e] >> 16)) * p.c);
        }
        return as_frag<f16>(v);
    };
    if constexpr (!K16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { qf[t][0] = load_qf(t, 0); qf[t][1] = load_qf(t, 1); qh[t] = u32x2{0, 0}; }
    } else {
        // all eight loads of the wave's Q rows in one go, branch-free (a load under `if (d < DH)` came out as load, s_waitcnt vmcnt(0), convert -- four memory round
        // trips one behind the other in front of the first MFMA): lanes past the head dim read a clamped address and drop the value
        u32x4 ra[4]; u32x2 rb[4];
        const int dh = 32 + 4 * g, dhc = dh < DH ? dh : DH - 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f16* qp = p.q + ((size_t)(b * p.Nq + q0 + t * 16 + i16) * p.q_stride + h * DH);
            ra[t] = *reinterpret_cast<const u32x4*>(qp + 8 * g);
            rb[t] = *reinterpret_cast<const u32x2*>(qp + dhc);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 v = ra[t];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = pack2<f16>(El<f16>::tof((u16)(v[e] & 0xffff)) * p.c, El<f16>::tof((u16)(v[e] >> 16)) * p.c);
            qf[t][0] = as_frag<f16>(v);
            qf[t][1] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            u32x2 u = rb[t];
#pragma unroll
            for (int e = 0; e < 2; ++e)
                u[e] = dh < DH ? pack2<f16>(El<f16>::tof((u16)(u[e] & 0xffff)) * p.c, El<f16>::tof((u16)(u[e] >> 16)) * p.c) : 0u;
            qh[t] = u;
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(qf[t][0]), "+v"(qf[t][1]), "+v"(qh[t]));     // (the conversions above complete in front of the asm MFMAs that read them)
    asm volatile("s_nop 7" ::: "memory");
    f32x4 o_acc[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) o_acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sc[2][2][4];                  // [unit parity][key tile of the unit][query tile]
    u32x4 pfw[2][4];                    // [unit parity][query tile]: P fragment (B operand) as four packed words
    f32x4 negm[4];
    float m_run[4];
    f16x8 kf[4];                        // K fragments of the NEXT unit: (kt', ks) = (0,0) (0,1) (1,0) (1,1)
    u32x2 kh[2];                        // K16: the ks = 1 fragments as 64-bit operands (head dims 32 + 4 g .. + 3 of key tile kt')
    const unsigned g8 = (unsigned)g * 8;
    u32x2 vlo[3], vhi[3];               // V^T fragments (a = 0..2) of the unit whose P V product runs in the next slot
    const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned kl = sbase + i16 * RS + g * 16;
    const unsigned vl = sbase + KB + (4 * g + (i16 >> 2)) * RS + (i16 & 3) * 8;

    // One slot.  PAR = parity of unit n; reads: K fragments of unit n + 1 (half 1 - PAR of its tile, address ka), V^T fragments of unit n - 1 (half 1 - PAR, va).
    // MFMA groups in order: S(kt'0, ks0) | P V a0 | S(kt'0, ks1) | P V a1 | S(kt'1, ks0) | P V a2 | S(kt'1, ks1); behind group j the fragment it used is re-read
    // for the next slot, so in front of a group (10 - its own events) reads are younger than its operand: lgkmcnt 9 (K, one event) / 8 (V^T, two events).
    // E strand: items k = (t, kt') of 4 exponentials + 2 conversions, the conversions one item behind their exponentials, spread evenly behind the 28 MFMAs.
    auto slot = [&](auto par_tag, auto s_tag, auto e_tag, auto p_tag, unsigned ka, unsigned va) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr int DO_S = decltype(s_tag)::value;          // 0 none, 1 accumulate from -m_ref, 2 from zero (first tile: raw scores)
        constexpr bool DO_E = decltype(e_tag)::value, DO_P = decltype(p_tag)::value;
        constexpr int H2 = 1 - PAR;                          // half (of its tile) of units n + 1 and n - 1
        auto e_instr = [&](auto q_tag) {
            constexpr int Q = decltype(q_tag)::value;
            if constexpr (DO_E) {
                constexpr int SP = 1 - PAR;                  // scores of unit n - 1, P fragment of unit n - 1
                if constexpr (Q < 4) a40_exp(sc[SP][0][0][Q]);
                else {
                    constexpr int QQ = Q - 4, K = 1 + QQ / 6, R = QQ % 6;
                    if constexpr (K <= 7) {
                        if constexpr (R < 4) a40_exp(sc[SP][K % 2][K / 2][R]);
                        else { constexpr int KP = K - 1, C = R - 4; a40_cvt(pfw[SP][KP / 2][2 * (KP % 2) + C], sc[SP][KP % 2][KP / 2][2 * C], sc[SP][KP % 2][KP / 2][2 * C + 1]); }
                    } else { constexpr int C = QQ - 42; a40_cvt(pfw[SP][3][2 + C], sc[SP][1][3][2 * C], sc[SP][1][3][2 * C + 1]); }
                }
            }
        };
        auto filler = [&](auto m_tag) {                       // E instructions behind MFMA m of the slot
            constexpr int M = decltype(m_tag)::value;
            constexpr int LO = 48 * M / 28, HI = 48 * (M + 1) / 28;
            a40_for<HI - LO>([&](auto i_tag) { e_instr(std::integral_constant<int, LO + decltype(i_tag)::value>{}); });
        };
        auto s_group = [&](auto j_tag) {                      // j = kt' * 2 + ks
            constexpr int J = decltype(j_tag)::value, KT = J / 2, KS = J % 2, G = 2 * J;
            constexpr bool H16 = K16 && KS == 1;              // this group's fragment is the 64-bit one
            if constexpr (H16) a40_wait64<9>(kh[KT]); else a40_wait<9>(kf[J]);
            a40_for<4>([&](auto t_tag) {
                constexpr int T = decltype(t_tag)::value;
                if constexpr (DO_S != 0) {
                    if constexpr (KS == 0) { if constexpr (DO_S == 1) a40_mfma_c(sc[PAR][KT][T], kf[J], qf[T][0], negm[T]); else a40_mfma_z(sc[PAR][KT][T], kf[J], qf[T][0]); }
                    else if constexpr (H16) a40_mfma16(sc[PAR][KT][T], kh[KT], qh[T]);
                    else a40_mfma(sc[PAR][KT][T], kf[J], qf[T][1]);
                }
                filler(std::integral_constant<int, G * 4 + T>{});
            });
            if constexpr (H16) a40_read_k64<(2 * H2 + KT) * 16 * RS + 64>(kh[KT], ka - g8);
            else a40_read_k<(2 * H2 + KT) * 16 * RS + KS * 64>(kf[J], ka);
        };
        auto p_group = [&](auto a_tag) {
            constexpr int A = decltype(a_tag)::value, G = 2 * A + 1;
            a40_wait2<8>(vlo[A], vhi[A]);
            if constexpr (DO_P) {
                const f16x8 vf = as_frag<f16>(u32x4{vlo[A][0], vlo[A][1], vhi[A][0], vhi[A][1]});
                a40_for<4>([&](auto t_tag) {
                    constexpr int T = decltype(t_tag)::value;
                    a40_mfma(o_acc[A][T], vf, as_frag<f16>(pfw[PAR][T]));
                    filler(std::integral_constant<int, G * 4 + T>{});
                });
            } else {
                a40_for<4>([&](auto t_tag) { filler(std::integral_constant<int, G * 4 + decltype(t_tag)::value>{}); });
            }
            a40_read_vt<32 * H2 * RS + A * 32>(vlo[A], va);
            a40_read_vt<32 * H2 * RS + A * 32 + 16 * RS>(vhi[A], va);
        };
        s_group(std::integral_constant<int, 0>{}); p_group(std::integral_constant<int, 0>{});
        s_group(std::integral_constant<int, 1>{}); p_group(std::integral_constant<int, 1>{});
        s_group(std::integral_constant<int, 2>{}); p_group(std::integral_constant<int, 2>{});
        s_group(std::integral_constant<int, 3>{});
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using BT = std::true_type; using BF = std::false_type;

    // stage byte offsets of tiles i - 1, i, i + 1
    int so_prev = 0, so_cur = 0, so_next = ST;
    __builtin_amdgcn_s_barrier();                                                // B(-1): tiles 0 and 1 are in LDS
    unsigned long long te1 = 0;
    if (TRACE) te1 = __builtin_amdgcn_s_memrealtime();
    // ---- tile 0: raw scores of both halves, the reference maximum, the exponentials of unit 0 --------------------------------------------
    // (the priming slot only issues the reads of a slot, so that every counted wait below sees the steady-state history; its waits are trivially true)
#pragma unroll
    for (int j = 0; j < 4; ++j) kf[j] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    kh[0] = u32x2{0, 0}; kh[1] = u32x2{0, 0};
#pragma unroll
    for (int a = 0; a < 3; ++a) { vlo[a] = u32x2{0, 0}; vhi[a] = u32x2{0, 0}; }
    slot(I1{}, I0{}, BF{}, BF{}, kl, vl);                                         // reads K(unit 0)
    slot(I0{}, I2{}, BF{}, BF{}, kl, vl);                                         // S(0) raw; reads K(unit 1)
    slot(I1{}, I2{}, BF{}, BF{}, kl + so_next, vl);                               // S(1) raw; reads K(unit 2) of tile 1, V^T(unit 0)
    // The MFMA results are read by compiler-scheduled code next.  hipcc does not know that the producers are MFMAs (no hazard padding) and is free to move the
    // consumers up between the volatile asm statements (seen in one build: the column maxima were taken one instruction behind the MFMAs, the reference maximum
    // came out different and the outputs differed in the last bit): the padding is followed by empty asm statements that re-define every score register, so
    // that every consumer depends on something behind the padding in the (ordered) sequence of volatile statements.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(sc[u][kt][t]));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float mx = sc[0][0][t][0];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[u][kt][t][r]);
        mx = xor32_max(xor16_max(mx));
        m_run[t] = mx;
        negm[t] = f32x4{-mx, -mx, -mx, -mx};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[u][kt][t][r] -= mx;
    }
    asm volatile("s_nop 7" ::: "memory");
    {   // E(0) alone (in the steady state it runs inside slot 1)
        a40_for<48>([&](auto q_tag) {
            constexpr int Q = decltype(q_tag)::value;
            if constexpr (Q < 4) a40_exp(sc[0][0][0][Q]);
            else {
                constexpr int QQ = Q - 4, K = 1 + QQ / 6, R = QQ % 6;
                if constexpr (K <= 7) {
                    if constexpr (R < 4) a40_exp(sc[0][K % 2][K / 2][R]);
                    else { constexpr int KP = K - 1, C = R - 4; a40_cvt(pfw[0][KP / 2][2 * (KP % 2) + C], sc[0][KP % 2][KP / 2][2 * C], sc[0][KP % 2][KP / 2][2 * C + 1]); }
                } else { constexpr int C = QQ - 42; a40_cvt(pfw[0][3][2 + C], sc[0][1][3][2 * C], sc[0][1][3][2 * C + 1]); }
            }
        });
    }
    __builtin_amdgcn_s_barrier();                                                // B(0)
    unsigned long long tc0 = 0, tr0 = 0;
    if (TRACE) { tc0 = __builtin_readcyclecounter(); tr0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned long long te3 = 0;
    // ---- tiles 1 .. ntiles - 1 ----------------------------------------------------------------------------------------------------------
    // Placement of the hand-written loop in the instruction stream: without this anchor the product instantiation ran 1.06-1.09 ms where the TRACE instantiation of the
    // same source ran 0.92 (loop head at 60 vs 12 mod 64 bytes); behind a 64-byte alignment every padding 0..15 words measured 0.84-0.87 ms
    // (profiles/r03_attn_pad_sweep.txt; MI355X_MICROARCH.md "code-placement sensitivity of hand-written streams").
    asm volatile(".p2align 6\n\t.rept %0\n\ts_nop 0\n\t.endr" :: "n"(PAD));
    for (int i = 1; i < ntiles; ++i) {
        so_prev = so_cur; so_cur = so_next; so_next = so_next + ST == NS * ST ? 0 : so_next + ST;
        using SE = std::integral_constant<bool, DBG != 1>; using SP = std::integral_constant<bool, DBG != 2>; using SS = std::integral_constant<int, DBG != 2 ? 1 : 0>;
        slot(I0{}, SS{}, SE{}, SP{}, kl + so_cur, vl + so_prev);                  // unit 2i:     reads K(2i + 1) of tile i,     V^T(2i - 1) of tile i - 1
        slot(I1{}, SS{}, SE{}, SP{}, kl + so_next, vl + so_cur);                  // unit 2i + 1: reads K(2i + 2) of tile i + 1, V^T(2i)     of tile i
        __builtin_amdgcn_s_barrier();                                            // B(i): every read of tile i - 1 has been waited for
    }
    if (TRACE && w == 0 && lane == 0 && blockIdx.x < A40_TRACE_SLOTS) {
        g_a40_trace[blockIdx.x * A40_TRACE_W + 0] = __builtin_readcyclecounter() - tc0;
        g_a40_trace[blockIdx.x * A40_TRACE_W + 1] = __builtin_amdgcn_s_memrealtime() - tr0;
        g_a40_trace[blockIdx.x * A40_TRACE_W + 2] = (unsigned long long)(ntiles - 1);
    }
    if (TRACE) te3 = __builtin_amdgcn_s_memrealtime();
    // ---- drain: E(N - 1) | P V(N - 2), then P V(N - 1) -----------------------------------------------------------------------------------
    slot(I0{}, I0{}, BT{}, BT{}, kl + so_cur, vl + so_cur);                       // reads V^T(N - 1) of the last tile
    slot(I1{}, I0{}, BF{}, BT{}, kl + so_cur, vl + so_cur);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the slot's own trailing reads land before their registers are reused
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(kf[j]));
    asm volatile("" : "+v"(kh[0]), "+v"(kh[1]));
#pragma unroll
    for (int a = 0; a < 3; ++a) asm volatile("" : "+v"(vlo[a]), "+v"(vhi[a]));
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(o_acc[a][t]));            // (as above: the accumulators' consumers stay behind the padding)

    // ---- denominators; a wave with a non-finite one redoes its rows with running maxima, straight from global memory --------------------
    float lsum[4];
    bool bad = false;
#pragma unroll
    for (int t = 0; t < 4; ++t) { lsum[t] = __shfl(o_acc[2][t][0], 32 + i16, 64); bad |= !(lsum[t] < INFINITY); }
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) o_acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) m_run[t] = 0.f;
        for (int tile = 0; tile < ntiles; ++tile) {
            f32x4 s2[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const f16* krow = kbase + (size_t)(tile * 64 + kt * 16 + i16) * p.k_stride;
                const u32x4 k0 = *reinterpret_cast<const u32x4*>(krow + 8 * g);
                u32x4 k1 = {0, 0, 0, 0};
                if (g == 0) k1 = *reinterpret_cast<const u32x4*>(krow + 32);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float nm = tile == 0 ? 0.f : -m_run[t];
                    s2[kt][t] = El<f16>::mfma(as_frag<f16>(k0), qf[t][0], f32x4{nm, nm, nm, nm});
                    s2[kt][t] = El<f16>::mfma(as_frag<f16>(k1), K16 ? load_qf(t, 1) : qf[t][1], s2[kt][t]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float mx = s2[0][t][0];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s2[kt][t][r]);
                mx = xor32_max(xor16_max(mx));
                const float delta = tile == 0 ? mx : fmaxf(mx, 0.f);
                m_run[t] = tile == 0 ? delta : m_run[t] + delta;
                if (tile != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int a = 0; a < 3; ++a) o_acc[a][t] *= alpha;
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s2[kt][t][r] = __builtin_amdgcn_exp2f(s2[kt][t][r] - delta);
            }
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int d = a * 16 + i16;
                    union { u16 e[8]; u32x4 v; } vv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int key = tile * 64 + 32 * t2 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
                        vv.e[j] = d < DH ? *reinterpret_cast<const u16*>(vbase + (size_t)key * p.v_stride + d) : (d == DH ? (u16)0x3c00 : (u16)0);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const u32x4 f = {pack2<f16>(s2[2 * t2][t][0], s2[2 * t2][t][1]), pack2<f16>(s2[2 * t2][t][2], s2[2 * t2][t][3]),
                                         pack2<f16>(s2[2 * t2 + 1][t][0], s2[2 * t2 + 1][t][1]), pack2<f16>(s2[2 * t2 + 1][t][2], s2[2 * t2 + 1][t][3])};
                        o_acc[a][t] = El<f16>::mfma(as_frag<f16>(vv.v), as_frag<f16>(f), o_acc[a][t]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) lsum[t] = __shfl(o_acc[2][t][0], 32 + i16, 64);
    }
    // ---- epilogue ------------------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float inv = 1.0f / lsum[t];
        const int qrow = q0 + t * 16 + i16;
        f16* orow = p.out + (size_t)(b * p.Nq + qrow) * p.out_stride + h * DH;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int d = a * 16 + 4 * g;
            if (d + 4 <= DH) {
                const u32x2 o = {pack2<f16>(o_acc[a][t][0] * inv, o_acc[a][t][1] * inv), pack2<f16>(o_acc[a][t][2] * inv, o_acc[a][t][3] * inv)};
                *reinterpret_cast<u32x2*>(orow + d) = o;
            }
        }
    }
    if (TRACE && w == 0 && lane == 0 && blockIdx.x < A40_TRACE_SLOTS) {
        // 100 MHz ticks: entry -> B(-1) | B(-1) -> B(0) | loop end -> exit; entry and exit stamps; hardware id (CU, SE, XCC)
        unsigned long long* tr = g_a40_trace + blockIdx.x * A40_TRACE_W;
        const unsigned long long te4 = __builtin_amdgcn_s_memrealtime();
        tr[5] = te1 - te0; tr[6] = tr0 - te1; tr[7] = te4 - te3; tr[8] = te0; tr[9] = te4;
        tr[10] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) | __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    }
}

constexpr int ATTN_SLOTS = 512;      // 256 CUs x 2 resident workgroups (launch_bounds(256, 2), <= 72 KB of LDS each)

template <typename T, int DH, int QT, bool CAUSAL = false, bool BIAS = false>
int launch_attn(AttnParams p, int B, hipStream_t s, void* split_ws = nullptr, size_t split_ws_bytes = 0) {
    constexpr int DK = (DH + 31) / 32 * 32, DVP = (DH + 15) / 16 * 16;
    constexpr int KS = (DK == 64) ? 128 : ((DK * 2) % 64 == 32 ? DK * 2 : DK * 2 + 32);
    constexpr int VS = ((DVP * 2) % 64 == 32) ? DVP * 2 : DVP * 2 + 32;
    constexpr size_t lds = 2 * 64 * (size_t)(KS + VS);
    auto kfn = attn_kernel<T, DH, QT, CAUSAL, BIAS>;
    static bool configured = false;
    if (!configured) {
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    constexpr int ROWS = 4 * QT * 16;
    p.nqb = (p.Nq + ROWS - 1) / ROWS; p.lin0 = 0; p.splits = 1; p.tiles_per_split = 0; p.part_rows = 0; p.part_o = nullptr; p.part_ml = nullptr;
    const long total = (long)p.nqb * p.H * B;
    if (total > 0x7fffffffL) CS_FAIL(CS_E_SHAPE, "attention: too many workgroups");
    long main_wgs = total;
    if constexpr (DH == 128 && !CAUSAL && !BIAS) {
        // Split-KV tail.  The grid runs in rounds of ATTN_SLOTS workgroups; a last round that fills under half of the chip (FLUX-Kontext: 1632
        // workgroups = 3.19 rounds) costs a whole round.  Its workgroups are instead launched once per key range (the ranges run side by side)
        // and a small kernel merges the partial softmaxes.
        const long tail = total % ATTN_SLOTS;
        const int tiles = (p.Nk + 63) / 64;
        if (split_ws && total > ATTN_SLOTS && tail > 0 && tail * 2 <= ATTN_SLOTS && tiles >= 8) {
            int splits = (int)std::min<long>(std::min<long>(ATTN_SLOTS / tail, 8), tiles / 4);
            const int per = (tiles + splits - 1) / splits;
            splits = (tiles + per - 1) / per;
            const size_t rows = (size_t)tail * ROWS;
            const size_t need = (size_t)splits * rows * (DH + 2) * sizeof(float);
            if (splits >= 2 && need <= split_ws_bytes) {
                main_wgs = total - tail;
                hipLaunchKernelGGL(kfn, dim3((unsigned)main_wgs), dim3(256), lds, s, p);
                AttnParams t = p;
                t.lin0 = (int)main_wgs; t.splits = splits; t.tiles_per_split = per; t.part_rows = (int)rows;
                t.part_o = (float*)split_ws; t.part_ml = t.part_o + (size_t)splits * rows * DH;
                auto sfn = attn_kernel<T, DH, QT, CAUSAL, BIAS, true>;
                static bool sconf = false;
                if (!sconf) {
                    CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    sconf = true;
                }
                hipLaunchKernelGGL(sfn, dim3((unsigned)tail, splits), dim3(256), lds, s, t);
                hipLaunchKernelGGL((attn_combine_kernel<T, DH>), dim3((unsigned)((rows * (DH / 8) + 255) / 256)), dim3(256), 0, s, t, ROWS);
                CS_CHECK_LAUNCH();
                return CS_OK;
            }
        }
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)main_wgs), dim3(256), lds, s, p);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

constexpr int A40_NS = 6;
int launch_attn40_lw(AttnParams p, int B, hipStream_t s) {
    constexpr size_t lds = (size_t)A40_NS * 12288;
    const bool trace = (tune().debug & 16384) != 0;
    auto kfn = trace ? ((tune().debug & 1) ? attn40_lw_kernel<A40_NS, true, 1> : (tune().debug & 2) ? attn40_lw_kernel<A40_NS, true, 2> : attn40_lw_kernel<A40_NS, true>)
                     : tune().attn_lw == 2 ? attn40_lw_kernel<A40_NS, false, 0, 0, false> : attn40_lw_kernel<A40_NS, false>;
    static bool configured = false;
    if (!configured) {
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn40_lw_kernel<A40_NS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn40_lw_kernel<A40_NS, false, 0, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn40_lw_kernel<A40_NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn40_lw_kernel<A40_NS, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn40_lw_kernel<A40_NS, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    p.nqb = p.Nq / 256; p.lin0 = 0; p.splits = 1; p.tiles_per_split = 0; p.part_rows = 0; p.part_o = nullptr; p.part_ml = nullptr;
    const long total = (long)p.nqb * p.H * B;
    if (total > 0x7fffffffL) CS_FAIL(CS_E_SHAPE, "attention: too many workgroups");
    hipLaunchKernelGGL(kfn, dim3((unsigned)total), dim3(512), lds, s, p);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

}  // namespace

int debug_attn_trace_read(void* dst, size_t bytes) {
    if (bytes > sizeof(unsigned long long) * A40_TRACE_SLOTS * A40_TRACE_W) bytes = sizeof(unsigned long long) * A40_TRACE_SLOTS * A40_TRACE_W;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_a40_trace), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? CS_OK : CS_E_HIP;
}

size_t attention_split_workspace_bytes(int B, int H, int Nq, int Nk, int dh) {
    if (dh != 128) return 0;
    const long total = (long)((Nq + 127) / 128) * H * B, tail = total % ATTN_SLOTS;
    if (total <= ATTN_SLOTS || tail == 0 || tail * 2 > ATTN_SLOTS || (Nk + 63) / 64 < 8) return 0;
    return (size_t)8 * tail * 128 * (dh + 2) * sizeof(float);
}

int launch_attention(const AttnArgs& a, hipStream_t s) {
    if (!a.q || !a.k || !a.v || !a.out) CS_FAIL(CS_E_ARG, "attention: null pointer");
    if (a.B <= 0 || a.Nq <= 0) return a.B < 0 ? CS_E_SHAPE : CS_OK;
    if (a.Nk <= 0) CS_FAIL(CS_E_SHAPE, "attention: Nk must be positive");
    if ((a.q_stride | a.k_stride | a.v_stride | a.out_stride) & 7) CS_FAIL(CS_E_SHAPE, "attention: strides must be multiples of 8 halfs");
    AttnParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.out = a.out;
    p.q_stride = a.q_stride; p.k_stride = a.k_stride; p.v_stride = a.v_stride; p.out_stride = a.out_stride;
    p.H = a.H; p.Nq = a.Nq; p.Nk = a.Nk;
    p.c = a.scale * 1.4426950408889634f;
    p.bias = a.bias;
    p.prio = tune().attn_prio < 0 ? (a.dh == 128 ? 1 : 0) : tune().attn_prio;
    if (a.bias) {
        if (a.dh != 64 || a.causal || a.Nk % 4) CS_FAIL(CS_E_UNSUPPORTED, "attention: the biased form is built for head dim 64, no mask, Nk %% 4 == 0");
        if (a.dtype == CS_BF16) return launch_attn<bf16_el, 64, 2, false, true>(p, a.B, s);
        return launch_attn<f16, 64, 2, false, true>(p, a.B, s);
    }
    if (a.causal) {
        if (a.dh != 64 || a.dtype == CS_BF16 || a.Nq != a.Nk) CS_FAIL(CS_E_UNSUPPORTED, "attention: the causal form is built for f16, head dim 64, Nq == Nk");
        return launch_attn<f16, 64, 2, true>(p, a.B, s);
    }
    switch (a.dh) {
        case 40: {
            const int qt = tune().attn_qt40;      // cs_set_tuning("attn_qt40", 2 | 4): query tiles per wave at head dim 40
            if (a.dtype == CS_BF16) CS_FAIL(CS_E_UNSUPPORTED, "attention: bf16 is built for head dim 128 only");
            if (tune().attn_lw && qt == 4 && a.Nq % 256 == 0 && a.Nk % 64 == 0) return launch_attn40_lw(p, a.B, s);
            if (qt == 4) return launch_attn<f16, 40, 4>(p, a.B, s);
            return launch_attn<f16, 40, 2>(p, a.B, s);
        }
        case 80: if (a.dtype == CS_BF16) break; return launch_attn<f16, 80, 2>(p, a.B, s);
        case 160: if (a.dtype == CS_BF16) break; return launch_attn<f16, 160, 1>(p, a.B, s);
        case 128:
            if (a.dtype == CS_BF16) return launch_attn<bf16_el, 128, 2>(p, a.B, s, a.split_ws, a.split_ws_bytes);
            return launch_attn<f16, 128, 2>(p, a.B, s, a.split_ws, a.split_ws_bytes);
        default: break;
    }
    CS_FAIL(CS_E_UNSUPPORTED, "attention: head dim %d / dtype %d not built (f16: 40/80/160/128, bf16: 128)", a.dh, a.dtype);
}
