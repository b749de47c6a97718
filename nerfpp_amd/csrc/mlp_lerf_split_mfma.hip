// mlp_lerf_split_mfma.hip -- the LeRF language head fused with its render pass (see mlp_lerf_mfma.hip) at fp32-grade precision: NRF_PREC_F16_SPLIT.
//
// The reference runs LeRFImpl::forward (LeRF.cpp:28-111) in fp32.  Here, as in mlp_nerf_split_mfma.hip, every fp32 quantity travels as an unevaluated sum
// of two fp16 numbers (hi = f16(v), lo = f16(v - hi): 22 significant bits); weights -- the Gram matrix of the embedding layer included -- are split at pack
// time, activations when a D tile becomes the next operand, and a product is three matrix instructions (Wh.xh into the main accumulator, Wl.xh + Wh.xl into
// a second one).  CuHashEmbedder features are exact fp16 numbers (CuHashEmbedder.cu:95), so the layers fed by them need two.
//
//   kernel A  sigma net (128 -> 256 -> 33): sigma_le per point
//   kernel B  sigma net -> LE0 (cat[geo, in] -> 256, ReLU) = a;  ||W a||^2 = a^T (W^T W) a from the Gram product (8 tiles);  the ray's
//             sum_s (w_s / ||W a_s||) a_s by reduce-scatter + one 128-byte float atomic per 32 neurons
//   kernel C  W applied once per ray to that sum, operand and weights as (hi, lo) pairs
//
// One wave per SIMD (4 waves, 128 points per workgroup pass): the (hi, lo) activations of a 256-wide layer are 128 registers, input and output of a layer 256.
// Weight stream: chunks of ONE neuron tile (<= 32 fragments: 16 k-steps x (hi, lo)), LDS-DMA through three buffers two chunks ahead, the DMA pieces issued
// between the matrix instructions of the running chunk.
#include "mlp_lerf_net.h"

#include <utility>

namespace nrf {
namespace lerf {

#ifndef NRF_LERF_GRAM_ALO
#define NRF_LERF_GRAM_ALO 0              // 1: the lo part of a enters the Gram product
#endif
#ifndef NRF_LERF_GRAM_GLO
#define NRF_LERF_GRAM_GLO 0              // 1: the lo part of the Gram matrix enters it
#endif
#ifndef NRF_LERF_GRAM_HI_STREAM
#define NRF_LERF_GRAM_HI_STREAM 1        // Gram layer without its lo part: only the hi fragments are streamed into LDS and read (0: both, as the image holds them)
#endif
#ifndef NRF_LERF_GRAM_TRI
#define NRF_LERF_GRAM_TRI 1              // Gram layer as the upper block triangle of the symmetric matrix (the image holds it that way, see mlp_lerf_mfma.hip): neuron tile T
#endif                                   // starts at k-step 2 T -- 72 instead of 128 k-steps per point tile; 0: all sixteen (the skipped blocks of the image are zero)

#ifndef NRF_LERF_GRAM_ORDER
#define NRF_LERF_GRAM_ORDER 1            // block-triangular Gram layer: neuron tiles in the order 0 7 1 6 2 5 3 4 (16, 2, 14, 4, ... k-steps), so any two consecutive chunks
#endif                                   // hold 18 k-steps -- the weight stream runs two CHUNKS ahead, and behind tiles 6, 7 (4 + 2 k-steps) that was ~200 cycles
#ifndef NRF_LERF_ABLATE
#define NRF_LERF_ABLATE 0                // timing-only ablations of kernel B (results are garbage): 1 no weight stream, 2 no chunk barrier, 4 no conversions, 8 no Gram dot
#endif                                   // products, 16 no ray sum, 32 no operand prefetch (tools/scratch/lerf_ablate.sh)

constexpr int SNW = 4;                 // waves per workgroup: one per SIMD
constexpr int SNBLK = 32 * SNW;
constexpr int SMAXF = 32;              // fragments in the largest chunk: 16 k-steps x (hi, lo)

// one neuron tile per chunk; per (layer, tile, k-step) the hi fragment then the lo fragment
// L0: first layer the kernel runs (2: kernel B fed the sigma net's output by kernel A, see Args::geo); chunks are numbered from that layer's first tile
template <int NL, int L0 = 0>
struct NetS {
    using F = Net<NL>;
    static constexpr int first_chunk(int l) { int n = 0; for (int i = L0; i < l; i++) n += F::tiles(i); return n; }
    static constexpr int total_chunks() { return first_chunk(NL); }
    static constexpr int layer_of(int ci) { int l = L0; while (first_chunk(l + 1) <= ci) l++; return l; }
    static constexpr int chunk_frags(int ci) { return 2 * F::ks(layer_of(ci)); }
    // The Gram layer (layer 3 of the 4-layer net) applies the hi fragments only (NRF_LERF_GRAM_GLO = 0): its chunks then stream and read HALF of what the image holds --
    // fragment 2 i (hi of k-step i) into its usual slot, the lo slots left alone.  The weight stream's LDS-side write is what bounds kernel B (38 GB/s per CU taken in,
    // matrix pipe busy 0.37-0.40), and the Gram chunks were a quarter of it.
    static constexpr bool hi_only(int ci) { return NL == 4 && layer_of(ci) == 3 && !(NRF_LERF_GRAM_GLO) && (NRF_LERF_GRAM_HI_STREAM); }
    // Gram layer, block-triangular (NRF_LERF_GRAM_TRI): tile T of the layer multiplies k-steps 2 T .. 15 only; its stream starts at the multiple of four below (the
    // pieces are dealt out four fragments -- one per wave -- at a time)
    // neuron tile a chunk computes: its position in the layer, except in the block-triangular Gram layer (NRF_LERF_GRAM_ORDER)
    static constexpr int tile_of(int ci)
    {
        const int c = ci - first_chunk(layer_of(ci));
        if (NL == 4 && layer_of(ci) == 3 && (NRF_LERF_GRAM_TRI) && (NRF_LERF_GRAM_ORDER)) return (c & 1) ? 7 - (c >> 1) : (c >> 1);
        return c;
    }
    static constexpr int k0(int ci) { return (NL == 4 && layer_of(ci) == 3 && (NRF_LERF_GRAM_TRI)) ? 2 * tile_of(ci) : 0; }
    static constexpr int k0_dma(int ci) { return k0(ci) & ~3; }
    static constexpr int dma_frags(int ci) { return (hi_only(ci) ? 1 : 2) * (F::ks(layer_of(ci)) - k0_dma(ci)); }
    static constexpr int image_off() { int n = 0; for (int i = 0; i < L0; i++) n += F::tiles(i) * 2 * F::ks(i); return n; }      // fragments of the skipped layers
    static constexpr int chunk_off(int ci) { int n = image_off(); for (int i = 0; i < first_chunk(layer_of(ci)); i++) n += chunk_frags(i); return n + tile_of(ci) * chunk_frags(ci); }
};
static_assert(NetS<4, 2>::total_chunks() == 16 && NetS<4, 2>::chunk_off(0) == 2 * (8 * 8 + 2 * 16), "kernel B from LE0 on");
static_assert(NetS<2>::total_chunks() == 10 && NetS<4>::total_chunks() == 26, "chunk counts");
static_assert(NetS<4>::chunk_off(26) == 2 * LE1_FRAG0, "the split image doubles the fp16 image");

template <class N, int CI, int Q>
__device__ __forceinline__ void stage_piece(half8 *__restrict__ dst, const half8 *__restrict__ packed, int wave, int lane)
{
    constexpr int ci = CI % N::total_chunks();
    constexpr int nf = N::dma_frags(ci);
    constexpr int STEP = N::hi_only(ci) ? 2 : 1;          // hi-only chunks: every second fragment of the image, into its usual slot
    static_assert(nf % SNW == 0, "fragments per chunk must divide by the wave count");
    if constexpr (Q * SNW < nf && !((NRF_LERF_ABLATE) & 1)) {
        constexpr int base = N::chunk_off(ci);
        // the fragment's address = SGPR base (its constant offset added on the scalar side, then made opaque) + lane * 16: the saddr form of the DMA.  With the offset
        // added after the opaque point the compiler forms a 64-bit per-lane address instead -- two v_lshl_add_u64 per DMA, ~1 070 per iteration of the classic kernel
        const half8 *pk = packed + (size_t)wave * (64 * STEP);
        asm volatile("" : "+s"(pk));                          // opaque: the addresses derived from it cannot be hoisted out of the persistent loop (533 SGPR pairs would spill)
        constexpr int F0 = 2 * N::k0_dma(ci);                 // first fragment of the chunk that travels (block-triangular Gram tiles skip their leading k-steps)
        pk += (size_t)(base + F0 + STEP * Q * SNW) * 64;
        asm volatile("" : "+s"(pk));                          // the offset is added HERE, on the scalar side (s_add_u32 / s_addc_u32)
        __builtin_amdgcn_global_load_lds(pk + lane, (__attribute__((address_space(3))) void *)(dst + (F0 + STEP * (Q * SNW + wave)) * 64), 16, 0, 0);
    }
}

template <class N, int CI, int... Qs>
__device__ __forceinline__ void stage_all(half8 *__restrict__ dst, const half8 *__restrict__ packed, int wave, int lane, std::integer_sequence<int, Qs...>)
{
    (stage_piece<N, CI, Qs>(dst, packed, wave, lane), ...);
}

__device__ __forceinline__ void split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
}

// registers 8s..8s+7 of a finished tile -> the (hi, lo) operand fragments of k-step s.  The asm reads VALU results only (the max / the add).
template <bool RELU>
__device__ __forceinline__ void tile_to_frag2(const f32x16 &t, int s, half8 &hi, half8 &lo)
{
    union { half8 v; uint32_t u[4]; } h, l;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float lim = RELU ? 0.0f : -3.402823466e38f;          // max with -FLT_MAX: the identity on every finite value, and a VALU result for the asm to read
        split_pair(fmaxf(t[8 * s + 2 * j], lim), fmaxf(t[8 * s + 2 * j + 1], lim), h.u[j], l.u[j]);
    }
    hi = h.v; lo = l.v;
}

struct CtxS {
    half8 *wbuf;
    const half8 *packed;
    int lane, h, wave;
    int *cur;
    f32x16 *accs;           // two accumulator tiles: tile T of a layer accumulates into accs[T & 1] while tile T - 1 (in the other) is being converted
};

// Prefetch policy of a kernel's tile loop: loads issued at the TAIL of chunk ci (after its DMA pieces, before its end-of-chunk wait) into registers the
// running layers do not read.  The end-of-chunk wait counts vector memory operations in issue order ("all but the youngest n have landed" must cover chunk
// ci + 1's weights, issued during chunk ci - 1), so the count(ci) loads of a tail stay in flight across TWO chunks: the wait of chunk ci and that of chunk
// ci + 1 allow count(ci) more, the wait of chunk ci + 2 retires them.  count() is a compile-time constant per chunk: every load of a tail is unconditional.
struct NoPF {
    static constexpr int count(int) { return 0; }
    template <int CI> __device__ __forceinline__ void issue() {}
    template <int CI> __device__ __forceinline__ void kstep(int) {}          // vector work a kernel hides behind chunk CI's k-step k (see GeoPF)
};

// One chunk = neuron tile T of layer L; the finished tile (main + correction accumulator) goes to hook(T, tile).  BNLO / BCLO: the natural / chained operand has a lo part.
template <class N, int L, int T, bool BNLO, bool BCLO, bool WLO, int NN, int NC, class Hook, class PF>
__device__ __forceinline__ void chunk_body_s(const CtxS &cx, const half8 *__restrict__ w, half8 *__restrict__ dma_dst, const half8 (&bn)[NN][2], const half8 (&bc)[NC][2], Hook &hook, PF &pf)
{
    using F = typename N::F;
    constexpr int KSN = F::ks_nat(L), KSC = F::ks_ch(L), KS = KSN + KSC;
    constexpr int CI = N::first_chunk(L) + T;
    constexpr bool NATF = F::nat_first(L);
    static_assert(KSN <= NN && KSC <= NC, "operand fragment arrays too small");
    constexpr int K0 = N::k0(CI);                 // first k-step of this chunk (block-triangular Gram tiles: 2 T)
    constexpr int TILE = N::tile_of(CI), TILE_PREV = T > 0 ? N::tile_of(CI - 1) : 0;          // constants: as call arguments the compiler evaluated them at run time (and the hooks' fragment arrays went to scratch)
    constexpr int KRUN = KS - K0;
    constexpr int NQ = N::dma_frags((CI + 2) % N::total_chunks()) / SNW;
    constexpr int EVERY = (KRUN / NQ) > 0 ? (KRUN / NQ) : 1;
    constexpr int LEAD = NQ > KRUN / EVERY ? NQ - KRUN / EVERY : 0;
    static_assert(NQ <= 8, "piece switch covers 8 pieces per wave");
    stage_all<N, CI + 2>(dma_dst, cx.packed, cx.wave, cx.lane, std::make_integer_sequence<int, LEAD>{});
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // One accumulator per tile, small terms first within a k-step (a dependent chain of this instruction issues back to back).  Hooks that convert a tile into
    // the next layer's operands (UNITS > 0) are DEFERRED within a layer: tile T - 1, finished in the other accumulator, is converted one unit per k-step behind
    // tile T's matrix instructions instead of after its own (at one wave per SIMD nothing else overlaps those ~180 vector instructions); a layer's last tile is
    // converted at once.
    constexpr int NT = F::tiles(L);
    constexpr bool DEFER = Hook::UNITS > 0;
    constexpr bool PEND = DEFER && T > 0;
    static_assert(!DEFER || Hook::UNITS <= KRUN, "a deferred tile's units must fit the next tile's k-steps");
    f32x16 &acc = cx.accs[T & 1];
    const f32x16 &prev = cx.accs[(T & 1) ^ 1];
    acc = zero;
    int q = LEAD;
    // weight fragments two k-steps ahead of their matrix instructions through three register slots, with counted waits (see mlp_nerf_split_mfma.hip: left to
    // the compiler the reads sit 32 pipe cycles before their use, or it waits for ALL outstanding reads)
    half8 fa[3][2];
    const uint32_t waddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(w + cx.lane);
    constexpr bool HI = N::hi_only(CI);          // the lo fragments of this chunk were not streamed: they are not read either
    static_assert(!HI || !WLO, "a hi-only chunk cannot apply the weights' lo part");
    auto read_pair = [&](int kk_, int slot) {
        if constexpr (HI) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(fa[slot][0]) : "v"(waddr), "i"(kk_ * 2048));
        else asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(fa[slot][0]), "=&v"(fa[slot][1]) : "v"(waddr), "i"(kk_ * 2048), "i"(kk_ * 2048 + 1024));     // offsets in the instruction, not a v_add_u32 per pair
    };
    read_pair(K0, K0 % 3);
    if (KRUN > 1) read_pair(K0 + 1, (K0 + 1) % 3);
#pragma unroll
    for (int k = K0; k < KS; k++) {
        if (k + 2 < KS) {
            read_pair(k + 2, (k + 2) % 3);
            if constexpr (HI) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[k % 3][0]));
            else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[k % 3][0]), "+v"(fa[k % 3][1]));
        } else if (k + 1 < KS) {
            if constexpr (HI) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa[k % 3][0]));
            else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[k % 3][0]), "+v"(fa[k % 3][1]));
        } else {
            if constexpr (HI) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[k % 3][0]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[k % 3][0]), "+v"(fa[k % 3][1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        const half8 ah = fa[k % 3][0], al = HI ? fa[k % 3][0] : fa[k % 3][1];
        const bool nat = NATF ? (k < KSN) : (k >= KSC);
        const int kk = NATF ? (nat ? k : k - KSN) : (nat ? k - KSC : k);
        const half8 bh = nat ? bn[nat ? kk : 0][0] : bc[nat ? 0 : kk][0];
        if constexpr (WLO) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);      // WLO = false: the weights' lo fragments are not applied (see the Gram layer)
        if (nat ? BNLO : BCLO) {
            const half8 bl = nat ? bn[nat ? kk : 0][1] : bc[nat ? 0 : kk][1];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        if constexpr (PEND) { if (k - K0 < Hook::UNITS) hook.unit(TILE_PREV, prev, k - K0); }
        pf.template kstep<CI>(k);
        if (((k - K0) % EVERY) == EVERY - 1 && q < NQ) {
            const int qq = q;
            switch (qq) {
#define NRF_PIECE(Q) case Q: stage_piece<N, CI + 2, Q>(dma_dst, cx.packed, cx.wave, cx.lane); break;
                NRF_PIECE(0) NRF_PIECE(1) NRF_PIECE(2) NRF_PIECE(3) NRF_PIECE(4) NRF_PIECE(5) NRF_PIECE(6) NRF_PIECE(7)
#undef NRF_PIECE
            }
            q++;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int TOT = N::total_chunks();
    constexpr int EXTRA = PF::count(CI) + PF::count((CI + TOT - 1) % TOT);
    static_assert(NQ + EXTRA < 64, "vmcnt is a 6-bit counter");
    if constexpr (PF::count(CI) > 0) {
        pf.template issue<CI>();
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!DEFER || T == NT - 1) hook(TILE, acc);
    __builtin_amdgcn_sched_barrier(0);          // the tile is consumed here (see mlp_lerf_mfma.hip)
    if constexpr ((NRF_LERF_ABLATE) & 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(((NRF_LERF_ABLATE) & 1) ? EXTRA : NQ + EXTRA) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(((NRF_LERF_ABLATE) & 1) ? EXTRA : NQ + EXTRA) : "memory");
}

template <class N, int L, int T, bool BNLO, bool BCLO, bool WLO, int NN, int NC, class Hook, class PF>
__device__ __forceinline__ void chunk_s(const CtxS &cx, const half8 (&bn)[NN][2], const half8 (&bc)[NC][2], Hook &hook, PF &pf)
{
    const int cur = *cx.cur;
    chunk_body_s<N, L, T, BNLO, BCLO, WLO>(cx, cx.wbuf + cur * (SMAXF * 64), cx.wbuf + (cur == 0 ? 2 : cur - 1) * (SMAXF * 64), bn, bc, hook, pf);
    *cx.cur = cur == 2 ? 0 : cur + 1;
}

template <class N, int L, bool BNLO, bool BCLO, bool WLO, int NN, int NC, class Hook, class PF, int... Ts>
__device__ __forceinline__ void layer_seq_s(const CtxS &cx, const half8 (&bn)[NN][2], const half8 (&bc)[NC][2], Hook &hook, PF &pf, std::integer_sequence<int, Ts...>)
{
    (chunk_s<N, L, Ts, BNLO, BCLO, WLO>(cx, bn, bc, hook, pf), ...);
}

template <class N, int L, bool BNLO, bool BCLO, bool WLO = true, int NN, int NC, class Hook, class PF>
__device__ __forceinline__ void layer_s(const CtxS &cx, const half8 (&bn)[NN][2], const half8 (&bc)[NC][2], Hook &hook, PF &pf)
{
    layer_seq_s<N, L, BNLO, BCLO, WLO>(cx, bn, bc, hook, pf, std::make_integer_sequence<int, N::F::tiles(L)>{});
}

template <class N, int L, bool BNLO, bool BCLO, bool WLO = true, int NN, int NC, class Hook>
__device__ __forceinline__ void layer_s(const CtxS &cx, const half8 (&bn)[NN][2], const half8 (&bc)[NC][2], Hook &hook)
{
    NoPF pf;
    layer_s<N, L, BNLO, BCLO, WLO>(cx, bn, bc, hook, pf);
}

template <bool RELU, int NOUT, bool KEEP0 = false>
struct ConvHookS {
    half8 (&bout)[NOUT][2];
    float row0;
    static constexpr int UNITS = 8;          // deferrable: unit u = values 8s + 2j, 8s + 2j + 1 (s = u >> 2, j = u & 3) -> words j of fragments 2 tile + s
    __device__ __forceinline__ void operator()(int tile, const f32x16 &t)
    {
        if constexpr ((NRF_LERF_ABLATE) & 4) { asm volatile("" :: "v"(t[0])); return; }
        if (2 * tile + 1 < NOUT) {
            tile_to_frag2<RELU>(t, 0, bout[2 * tile][0], bout[2 * tile][1]);
            tile_to_frag2<RELU>(t, 1, bout[2 * tile + 1][0], bout[2 * tile + 1][1]);
        }
        if (KEEP0 && tile == 0) row0 = t[0];
    }
    __device__ __forceinline__ void unit(int tile, const f32x16 &t, int u)
    {
        if constexpr ((NRF_LERF_ABLATE) & 4) { asm volatile("" :: "v"(t[0])); return; }
        if (2 * tile + 1 < NOUT) {
            const int s = u >> 2, j = u & 3;
            const float lim = RELU ? 0.0f : -3.402823466e38f;
            uint32_t hi, lo;
            split_pair(fmaxf(t[8 * s + 2 * j], lim), fmaxf(t[8 * s + 2 * j + 1], lim), hi, lo);
            u32x4 hv = __builtin_bit_cast(u32x4, bout[2 * tile + s][0]), lv = __builtin_bit_cast(u32x4, bout[2 * tile + s][1]);
            hv[j] = hi; lv[j] = lo;
            bout[2 * tile + s][0] = __builtin_bit_cast(half8, hv); bout[2 * tile + s][1] = __builtin_bit_cast(half8, lv);
        }
        if (KEEP0 && tile == 0 && u == 0) row0 = t[0];
    }
};

// Two ways of hiding kernel B's open vector work behind matrix instructions, both measured on the LeRF frame in one call against the build without them
// (docs/history/profiles/round3/r3u_lerf_defer_ab.log; 74.1-74.5 ms of LeRF passes per frame): DOT, the Gram tiles' dot products one per k-step behind the next tile --
// 74.1-75.3 ms, nothing; SUM, a tile's share of the ray's sum behind the next tile's LE0 -- 77.5-79.5 ms, slower (the conversions it interleaves with already
// fill those k-steps).  The matrix pipe is this kernel's clock-limited resource; cycles freed beside it buy nothing.  Both off.
#ifndef NRF_LERF_DEFER_DOT
#define NRF_LERF_DEFER_DOT 0
#endif
#ifndef NRF_LERF_DEFER_SUM
#define NRF_LERF_DEFER_SUM 0
#endif
// a . (G a) with a = hi + lo.  NRF_LERF_DEFER_DOT: deferred like the conversions, tile T - 1's sixteen products one per k-step behind tile T's matrix instructions
// (same order of additions as consuming each tile at once), the layer's last tile in the open.
// acc + x * (hi + lo) with the (hi, lo) pair's halves entering as fp16 operands of two mixed-precision FMAs (v_fma_mix_f32 converts in the instruction): two vector
// instructions per value where convert, convert, add, fma took four -- kernel B reconstructs its 128 activations per lane twice per tile (Gram dot product, ray sum)
#ifndef NRF_LERF_MIX_FMA
#define NRF_LERF_MIX_FMA 1
#endif
__device__ __forceinline__ float mix_dot(float x, const half8 (&pair)[2], int k, float acc)
{
#if NRF_LERF_MIX_FMA
    acc = __builtin_fmaf(x, (float)pair[1][k], acc);          // the small term first
    return __builtin_fmaf(x, (float)pair[0][k], acc);
#else
    return __builtin_fmaf(x, (float)pair[0][k] + (float)pair[1][k], acc);
#endif
}
// the same where the compiler does not fold the conversions by itself (the ray sums: it converts both halves and issues two v_fmac).  Only for operands that vector
// instructions produced: x, the pair and acc here are -- the asm hides its reads from the MFMA -> VALU hazard handling, so it must never read a matrix result.
__device__ __forceinline__ float mix_dot_asm(float x, const half8 (&pair)[2], int k, float acc)
{
#if NRF_LERF_MIX_FMA
    const uint32_t wh = __builtin_bit_cast(u32x4, pair[0])[k >> 1], wl = __builtin_bit_cast(u32x4, pair[1])[k >> 1];
    if (k & 1) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(x), "v"(wl));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(x), "v"(wh));
    } else {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(x), "v"(wl));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(x), "v"(wh));
    }
    return acc;
#else
    return mix_dot(x, pair, k, acc);
#endif
}

struct DotHookS {
    const half8 (&a)[16][2];
    float ss = 0.0f;
    static constexpr int UNITS = NRF_LERF_DEFER_DOT ? 16 : 0;
    __device__ __forceinline__ void unit(int tile, const f32x16 &t, int i)
    {
        ss = mix_dot(t[i], a[2 * tile + (i >> 3)], i & 7, ss);
        asm volatile("" : "+v"(ss));          // pin the partial sum (see mlp_lerf_mfma.hip)
    }
    __device__ __forceinline__ void operator()(int tile, const f32x16 &t)
    {
        if constexpr ((NRF_LERF_ABLATE) & 8) { ss += t[0]; return; }
#pragma unroll
        for (int i = 0; i < 16; i++) ss = mix_dot(t[i], a[2 * tile + (i >> 3)], i & 7, ss);
        asm volatile("" : "+v"(ss));
    }
};

// the reduce-scatter over 32 sample slots of mlp_lerf_mfma.hip's ReduceHook -- here once per RAY (the wave walks all of the ray's tiles and keeps the per-slot
// sums in registers), with plain stores: no atomics, no zero-initialised scratch, a deterministic sum
struct ReduceS {
    float *out_row;
    int r, h;
    template <int NKEEP>
    __device__ __forceinline__ void step(float (&v)[16]) const
    {
        const bool up = (r & (2 * NKEEP)) != 0;
#pragma unroll
        for (int i = 0; i < NKEEP; i++) {
            const float keep = up ? v[i + NKEEP] : v[i], send = up ? v[i] : v[i + NKEEP];
            v[i] = keep + __shfl_xor(send, 2 * NKEEP);
        }
    }
    // v: this lane's sum over the ray's tiles of (w_s / ||h_s||) a_s for its own sample slot; the wave owns the ray, so the result is stored, not added
    __device__ __forceinline__ void operator()(int tile, float (&v)[16]) const
    {
        step<8>(v); step<4>(v); step<2>(v); step<1>(v);
        v[0] += __shfl_xor(v[0], 1);
        const int i = r >> 1;
        if (out_row && (r & 1) == 0) out_row[tile * 32 + 16 * (i >> 3) + 8 * ((i & 7) >> 2) + 4 * h + (i & 3)] = v[0];
    }
};

// XLO: the input features carry a lo part (fp32 rows); level-major CuHashEmbedder features are exact fp16
template <int NL, bool XLO>
__global__ void __launch_bounds__(64 * SNW, 1)
k_lerf_split(int64_t npts, Args in, const half8 *__restrict__ packed)
{
    using N = NetS<NL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *wbuf = reinterpret_cast<half8 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    stage_all<N, 0>(wbuf, packed, wave, lane, std::make_integer_sequence<int, N::dma_frags(0) / SNW>{});
    stage_all<N, 1>(wbuf + SMAXF * 64, packed, wave, lane, std::make_integer_sequence<int, N::dma_frags(1) / SNW>{});
    __syncthreads();
    int cur = 0;
    f32x16 accs[2];
    // Kernel A walks blocks of 4 x 32 points.  Kernel B is RAY-owned: a wave takes one ray and walks its s / 32 tiles (the four waves of a workgroup walk four
    // rays in step, so they still share every weight chunk), summing (w_s / ||h_s||) a_s per sample slot in 128 registers; the reduce-scatter over the slots and
    // the store happen once per ray instead of once per tile with atomics (cycle stamps: the per-tile tail was 11.7 k of an iteration's 48.5 k cycles).
    constexpr bool RAYS = NL != 2;
    const int tpr = RAYS ? in.s / 32 : 1;                          // tiles per ray
    const int64_t nrays = RAYS ? npts / in.s : 0;
    const int64_t nblocks = RAYS ? (nrays + SNW - 1) / SNW : (npts + SNBLK - 1) / SNBLK;
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
      float vsum[RAYS ? 8 : 1][16];
      if constexpr (RAYS) {
#pragma unroll
          for (int t = 0; t < 8; t++)
#pragma unroll
              for (int i = 0; i < 16; i++) vsum[t][i] = 0.0f;
      }
      const int64_t ray = blk * SNW + wave;                          // kernel B
      const bool rlive = ray < nrays;
      for (int jt = 0; jt < tpr; jt++) {
        CtxS cx{wbuf, packed, lane, h, wave, &cur, accs};
        const int64_t p0 = RAYS ? (rlive ? ray : nrays - 1) * in.s + jt * 32 : blk * SNBLK + wave * 32;
        const int64_t q = p0 + r;
        const bool live = RAYS ? rlive : q < npts;
        const int64_t qc = RAYS ? q : (live ? q : npts - 1);
        half8 none[1][2];
        // input operand: element j of k-step s is x[q][16 s + 8 h + j]; needed by layer 0 and again by layer 2 (cat[geo, in]): read twice
        auto load_x = [&](half8 (&xin)[8][2]) {
            if (in.x_lm) {
                const int64_t col = in.src ? (int64_t)in.src[qc] : qc;
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    xin[s][0] = *reinterpret_cast<const half8 *>(in.x_lm + ((int64_t)(2 * s + h) * in.pstride + col) * 8);
                    xin[s][1] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                }
                return;
            }
            const float *row = in.x + qc * in.x_stride;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const float4 lo4 = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h);
                const float4 hi4 = *reinterpret_cast<const float4 *>(row + 16 * s + 8 * h + 4);
                const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
                for (int j = 0; j < 8; j++) { const _Float16 t = (_Float16)v[j]; xin[s][0][j] = t; xin[s][1][j] = (_Float16)(v[j] - (float)t); }
            }
        };
        half8 ba[16][2], bb[4][2];
        half8 *geo = reinterpret_cast<half8 *>(in.geo);
        {
            ConvHookS<true, 16> c0{ba, 0.0f};
            {
                half8 xin[8][2];
                load_x(xin);
                layer_s<N, 0, XLO, false>(cx, xin, none, c0);                 // sigma0: 128 -> 256, ReLU
            }
            ConvHookS<false, 4, NL == 2> c1{bb, 0.0f};
            layer_s<N, 1, false, true>(cx, none, ba, c1);                     // sigma1: 256 -> (sigma, geo32)
            if constexpr (NL == 2) {
                if (h == 0 && live) {
                    float sg = c1.row0;
                    if (in.keep && !in.keep[q]) sg = 0.0f;                    // raw_le[~keep, -1] = 0 (LeRFRenderer.cpp:22-23)
                    in.sigma[q] = sg;
                }
                if (geo && live) {
#pragma unroll
                    for (int f = 0; f < GEO_FRAGS; f++)
#pragma unroll
                        for (int part = 0; part < 2; part++) geo[(((int64_t)(f * 2 + part) * in.geo_stride + q) << 1) + h] = bb[f][part];
                }
            }
        }
        if constexpr (NL != 2) {
            ConvHookS<true, 16> c2{ba, 0.0f};
            {
                half8 xin[8][2];
                load_x(xin);
                layer_s<N, 2, XLO, true>(cx, xin, bb, c2);                // LE0: cat[geo, in] -> 256, ReLU
            }
            DotHookS ssq{ba};
            layer_s<N, 3, false, NRF_LERF_GRAM_ALO != 0, NRF_LERF_GRAM_GLO != 0>(cx, none, ba, ssq);   // ||LE1(a)||^2 = a . (W^T W) a  (see k_lerf_split_geo)
            const float tot = fmaxf(ssq.ss + __shfl_xor(ssq.ss, 32), 0.0f) * in.gram_scale;
            const float wgt = live ? in.weights[q] : 0.0f;
            const float f = wgt / fmaxf(sqrtf(tot), 1e-8f);
            // fragments 2t, 2t+1 of a hold, on each lane, the neurons of D-tile t's 16 registers
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int i = 0; i < 16; i++) vsum[t][i] = mix_dot_asm(f, ba[2 * t + (i >> 3)], i & 7, vsum[t][i]);
        }
      }
      if constexpr (RAYS) {
          ReduceS red{rlive ? in.out + ray * (int64_t)HID : nullptr, r, h};
#pragma unroll
          for (int t = 0; t < 8; t++) red(t, vsum[t]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Kernel B started from the sigma net's output (Args::geo, level-major features): LE0 and the Gram product only, 16 chunks per tile of 32 samples.  At one wave
// per SIMD nothing hides a tile's operand loads, and they hang on a dependent chain (merge map src[q] -> column -> six geo fragments and eight feature
// fragments, then the matrix instructions): the tile loop is therefore a software pipeline over the wave's whole tile sequence (ray after ray), and everything
// it prefetches travels by LDS-DMA into the wave's private 15 KB behind the weight buffers -- the kernel has no registers to spare (ONE more value live across
// the tile costs ~200 spills), and a DMA leaves the compiler nothing to wait for.  During tile t:
//   chunk 0's tail   src[q] of tile t + 1's sample -> column slot
//   chunk 4's tail   the column back (one LDS read), then tile t + 1's 14 operand fragments, each lane from its own column, and the sample's render weight
// and tile t + 1 starts with 14 LDS reads.  Waits: the counted scheme described at NoPF.
constexpr int GEO_OPER_FRAGS = 2 * GEO_FRAGS + 8;          // per tile and lane: geo (hi, lo) x 3, features x 8
struct GeoPF {
    const Args &in;
    half8 *oper;               // the wave's slots: [GEO_OPER_FRAGS][64 lanes] fragments, then 64 floats x 2 (weights, by tile parity), then 64 columns
    int h;
    int64_t q1;                // tile t + 1: sample index of this lane (clamped to the launch)
    int par;                   // weight slot of the tile whose operands were issued last
    static constexpr int count(int ci) { return ((NRF_LERF_ABLATE) & 32) ? 0 : ci == 0 ? 1 : ci == 4 ? GEO_OPER_FRAGS + 1 : 0; }
    __device__ __forceinline__ float *wslot(int p) const { return reinterpret_cast<float *>(oper + GEO_OPER_FRAGS * 64) + p * 64; }
    __device__ __forceinline__ int32_t *cslot() const { return reinterpret_cast<int32_t *>(oper + GEO_OPER_FRAGS * 64) + 128; }
    // one DMA either way (the count is a constant): without a merge map the slot receives a dummy word and the column is the sample index
    __device__ __forceinline__ void load_col()
    {
        __builtin_amdgcn_global_load_lds(in.src ? in.src + q1 : reinterpret_cast<const int32_t *>(in.weights) + q1, (__attribute__((address_space(3))) void *)cslot(), 4, 0, 0);
    }
    __device__ __forceinline__ void load_operands(int lane)
    {
        int c;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(cslot() + lane)) : "memory");
        // per-lane part of the addresses (column, lane half) formed HERE, plane offsets uniform: nothing the compiler could precompute per lane and carry (or
        // spill) across the tile
        int64_t col = in.src ? (int64_t)c : q1;
        if constexpr ((NRF_LERF_ABLATE) & 64) col = col & 1023;          // timing only: every tile gathers the same cache-resident columns
        const half8 *gb = reinterpret_cast<const half8 *>(in.geo) + ((col << 1) + h);
        const __half *xb = in.x_lm + ((int64_t)h * in.pstride + col) * 8;
#pragma unroll
        for (int f = 0; f < 2 * GEO_FRAGS; f++)
            __builtin_amdgcn_global_load_lds(gb + (int64_t)f * in.geo_stride * 2, (__attribute__((address_space(3))) void *)(oper + f * 64), 16, 0, 0);
#pragma unroll
        for (int s = 0; s < 8; s++)
            __builtin_amdgcn_global_load_lds(xb + (int64_t)(2 * s) * in.pstride * 8, (__attribute__((address_space(3))) void *)(oper + (2 * GEO_FRAGS + s) * 64), 16, 0, 0);
        par ^= 1;                   // the weight is read at the END of its tile, after this tail has run for the next one: two slots
        __builtin_amdgcn_global_load_lds(in.weights + q1, (__attribute__((address_space(3))) void *)wslot(par), 4, 0, 0);
    }
    int lane_;
    template <int CI>
    __device__ __forceinline__ void issue()
    {
        if constexpr ((NRF_LERF_ABLATE) & 32) return;
        if constexpr (CI == 0) load_col();
        else load_operands(lane_);
    }
    // NRF_LERF_DEFER_SUM: the PREVIOUS tile's share of the ray's sum, vsum[t][.] += f_prev a_prev[.], taken behind LE0's matrix instructions: chunk t (neuron tile t of LE0) adds the
    // sixteen values of fragments 2t, 2t + 1 of a_prev, two per k-step, BEFORE the running tile's conversion overwrites them (tile t - 1's deferred units write
    // fragments 2t - 2, 2t - 1 during chunk t; tile 7 is converted after its k-loop).  f_prev = 0 on a ray's first tile; a ray's last tile is added in the open.
    float (&vsum)[8][16];
    const half8 (&ba)[16][2];
    float f_prev;
    template <int CI>
    __device__ __forceinline__ void kstep(int k)
    {
        if constexpr (NRF_LERF_DEFER_SUM && CI < 8) {
            if (k < 8) {
#pragma unroll
                for (int i = 2 * k; i < 2 * k + 2; i++) vsum[CI][i] = mix_dot(f_prev, ba[2 * CI + (i >> 3)], i & 7, vsum[CI][i]);
            }
        }
    }
    // weight of the RUNNING tile's sample (call after the tile's chunk 4: par already points at the next tile's slot)
    __device__ __forceinline__ float weight(int lane) const
    {
        float w;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(wslot(par ^ 1) + lane)) : "memory");
        return w;
    }
    // the tile's operands out of the slots (they landed chunks ago: the counted waits retire the DMA by the end of chunk 6)
    __device__ __forceinline__ void take(half8 (&bb)[4][2], half8 (&xin)[8][2], int lane) const
    {
        const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(oper + lane);
#pragma unroll
        for (int f = 0; f < GEO_FRAGS; f++)
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(bb[f][0]), "=&v"(bb[f][1]) : "v"(a), "i"(f * 2048), "i"(f * 2048 + 1024));
#pragma unroll
        for (int s = 0; s < 8; s += 2)
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(xin[s][0]), "=&v"(xin[s + 1][0]) : "v"(a), "i"((2 * GEO_FRAGS + s) * 1024), "i"((2 * GEO_FRAGS + s) * 1024 + 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the asm reads are opaque to the compiler: tie the registers to the wait
#pragma unroll
        for (int f = 0; f < GEO_FRAGS; f++) asm volatile("" : "+v"(bb[f][0]), "+v"(bb[f][1]));
#pragma unroll
        for (int s = 0; s < 8; s++) asm volatile("" : "+v"(xin[s][0]));
    }
};

__global__ void __launch_bounds__(64 * SNW, 1)
k_lerf_split_geo(int64_t npts, Args in, const half8 *__restrict__ packed)
{
    using N = NetS<4, 2>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8 *wbuf = reinterpret_cast<half8 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    half8 *oper = wbuf + 3 * SMAXF * 64 + wave * ((GEO_OPER_FRAGS + 1) * 64);
    stage_all<N, 0>(wbuf, packed, wave, lane, std::make_integer_sequence<int, N::dma_frags(0) / SNW>{});
    stage_all<N, 1>(wbuf + SMAXF * 64, packed, wave, lane, std::make_integer_sequence<int, N::dma_frags(1) / SNW>{});
    __syncthreads();
    int cur = 0;
    f32x16 accs[2];
    const int tpr = in.s / 32;                                     // tiles per ray
    const int64_t nrays = npts / in.s;
    const int64_t nblocks = (nrays + SNW - 1) / SNW;               // the four waves of a workgroup walk four rays in step (see k_lerf_split)
    // this lane's sample of tile (blk, jt) of the wave's sequence; rays past the end repeat the last one (loaded, computed, never stored)
    auto sample_of = [&](int64_t blk, int jt) -> int64_t {
        const int64_t ray = blk * SNW + wave;
        return (ray < nrays ? ray : nrays - 1) * in.s + jt * 32 + r;
    };
    // cursor over the tile sequence, one tile ahead of the running one
    int64_t nb = blockIdx.x; int nj = 0;
    auto advance = [&]() { if (++nj == tpr) { nj = 0; nb += gridDim.x; } };
    float vsum[8][16];
    half8 ba[16][2];
#pragma unroll
    for (int i = 0; i < 16; i++) { ba[i][0] = half8{0, 0, 0, 0, 0, 0, 0, 0}; ba[i][1] = ba[i][0]; }          // read (times f_prev = 0) by the first tile
    GeoPF pf{in, oper, h, 0, 0, lane, vsum, ba, 0.0f};
    // prologue: tile 0's column, then its operands
    pf.q1 = sample_of(nb, nj);
    pf.load_col();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pf.load_operands(lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
#pragma unroll
      for (int t = 0; t < 8; t++)
#pragma unroll
          for (int i = 0; i < 16; i++) vsum[t][i] = 0.0f;
      const int64_t ray = blk * SNW + wave;
      const bool rlive = ray < nrays;
#pragma unroll 1
      for (int jt = 0; jt < tpr; jt++) {
        CtxS cx{wbuf, packed, lane, h, wave, &cur, accs};
        half8 bb[4][2], xin[8][2], none[1][2];
        // tile t: its operands and its weight wait in the wave's slots; the cursor moves on to tile t + 1
        pf.take(bb, xin, lane);
        bb[3][0] = half8{0, 0, 0, 0, 0, 0, 0, 0}; bb[3][1] = bb[3][0];
#pragma unroll
        for (int s = 0; s < 8; s++) xin[s][1] = bb[3][0];
        advance();
        pf.q1 = sample_of(nb, nj);
        ConvHookS<true, 16> c2{ba, 0.0f};
        layer_s<N, 2, false, true>(cx, xin, bb, c2, pf);                  // LE0: cat[geo, in] -> 256, ReLU      (chunk 0's tail: column of tile t + 1; chunk 4's: its operands)
        DotHookS ssq{ba};
        // The Gram product supplies ONE scalar per sample, the norm ||W a|| that scales the sample's share of the ray's sum; the embedding's DIRECTION comes from
        // the split-precision sum of a and kernel C.  An fp16-grade norm (relative error ~1e-4: rounding of a and of G, no lo parts) perturbs the shares by as much
        // and the rendered unit embedding by < 1e-6 per component -- measured against the CPU oracle on 256 rays of the bench frame (max abs error, rms):
        //   three products (Gh.ah + Gl.ah + Gh.al) 3.6e-7 / 3.4e-8, 97.1 ms of LeRF passes per frame;  two (a's hi part) 7.6e-7 / 9.1e-8, 87.4 ms;
        //   ONE (Gh.ah) 8.8e-7 / 1.2e-7, 77.5 ms -- the default.  cos >= 1 - 1.2e-7 in all three (docs/history/profiles/round3/r3i_lerf_gram_products.log).
        layer_s<N, 3, false, NRF_LERF_GRAM_ALO != 0, NRF_LERF_GRAM_GLO != 0>(cx, none, ba, ssq, pf);   // ||LE1(a)||^2 = a . (W^T W) a
        const float tot = fmaxf(ssq.ss + __shfl_xor(ssq.ss, 32), 0.0f) * in.gram_scale;
        const float wgt = rlive ? pf.weight(lane) : 0.0f;
        const float f = wgt / fmaxf(sqrtf(tot), 1e-8f);
        // fragments 2t, 2t+1 of a hold, on each lane, the neurons of D-tile t's 16 registers.  The tile's share is added behind the next tile's LE0 (GeoPF::kstep),
        // except a ray's last one
        if (NRF_LERF_DEFER_SUM && jt + 1 < tpr) pf.f_prev = f;
        else if constexpr ((NRF_LERF_ABLATE) & 16) vsum[0][0] += f;
        else {
            pf.f_prev = 0.0f;
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int i = 0; i < 16; i++) vsum[t][i] = mix_dot_asm(f, ba[2 * t + (i >> 3)], i & 7, vsum[t][i]);
        }
      }
      ReduceS red{rlive ? in.out + ray * (int64_t)HID : nullptr, r, h};
#pragma unroll
      for (int t = 0; t < 8; t++) red(t, vsum[t]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// kernel C: out[ray][768] = W . asum[ray][256], operand and weights as (hi, lo) pairs: three matrix instructions per fragment pair.  One wave per 32 rays;
// the 768 split fragments (768 KB, L2-resident) are read straight from global memory.
__global__ void __launch_bounds__(256)
k_lerf_embed_split(int64_t nrays, const float *__restrict__ asum, const half8 *__restrict__ packed, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t ray = tile * 32 + r;
    if (tile * 32 >= nrays) return;
    const int64_t rc = ray < nrays ? ray : nrays - 1;
    half8 bh[16], bl[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const float4 lo4 = *reinterpret_cast<const float4 *>(asum + rc * HID + 16 * s + 8 * h), hi4 = *reinterpret_cast<const float4 *>(asum + rc * HID + 16 * s + 8 * h + 4);
        const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
        for (int j = 0; j < 8; j++) { const _Float16 t = (_Float16)v[j]; bh[s][j] = t; bl[s][j] = (_Float16)(v[j] - (float)t); }
    }
    const half8 *w = packed + (size_t)(2 * LE1_FRAG0) * 64 + lane;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 24; t++) {
        f32x16 acc = zero, cor = zero;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const half8 ah = w[(size_t)(2 * (t * 16 + k)) * 64], al = w[(size_t)(2 * (t * 16 + k) + 1) * 64];
            cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[k], cor, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[k], acc, 0, 0, 0);
            cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[k], cor, 0, 0, 0);
        }
        if (ray < nrays) {
            float *o = out + ray * EMB + 32 * t + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; g++) *reinterpret_cast<float4 *>(o + 8 * g) = float4{acc[4 * g] + cor[4 * g], acc[4 * g + 1] + cor[4 * g + 1], acc[4 * g + 2] + cor[4 * g + 2], acc[4 * g + 3] + cor[4 * g + 3]};
        }
    }
}

template <int NL>
static int launch_lerf_split(const nrf_mlp *m, const Args &a_in, int64_t p, hipStream_t st)
{
    Args a = a_in;
    a.gram_scale = m->lerf_gram_scale;
    const size_t lds = (size_t)3 * SMAXF * 1024;
    const size_t lds_geo = lds + (size_t)SNW * (GEO_OPER_FRAGS + 1) * 1024;          // + the waves' operand slots (k_lerf_split_geo)
    const int64_t nblocks = NL == 2 ? ceil_div(p, SNBLK) : ceil_div(p / a.s, (int64_t)SNW);       // kernel B: one ray per wave
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);       // persistent: one 4-wave workgroup per CU
    const half8 *img = reinterpret_cast<const half8 *>(m->d_packed_split);
    static PerDeviceOnce attr_set;          // idempotent one-time setup per device (common.h)
    if (attr_set.needed()) {
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lerf_split<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lerf_split<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lerf_split<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lerf_split<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lerf_split_geo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_geo));
        attr_set.done();
    }
    if (NL == 4 && a.geo) {
        if (!a.x_lm) { set_error("LeRF passes: the sigma net's output is handed over on the level-major input path only"); return NRF_ERR_INVALID_ARG; }
        hipLaunchKernelGGL(k_lerf_split_geo, dim3(grid), dim3(64 * SNW), lds_geo, st, p, a, img);
    } else if (a.x_lm) hipLaunchKernelGGL((k_lerf_split<NL, false>), dim3(grid), dim3(64 * SNW), lds, st, p, a, img);
    else hipLaunchKernelGGL((k_lerf_split<NL, true>), dim3(grid), dim3(64 * SNW), lds, st, p, a, img);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

}  // namespace lerf

int lerf_split_available(const nrf_mlp *m) { return m && m->family == MLP_LERF && m->d_packed_split != nullptr && m->packed_split_bytes == (size_t)2 * lerf::IMAGE_FRAGS * 1024; }

int lerf_split_sigma(const nrf_mlp *m, const lerf::Args &a, int64_t p, hipStream_t st)
{
    ProfScope prof(NRF_PROF_MLP, st);
    return lerf::launch_lerf_split<2>(m, a, p, st);
}

// kernel B into a stream-ordered scratch asum [n][256] (zeroed), then kernel C
int lerf_split_embedding_passes(const nrf_mlp *m, lerf::Args a, int64_t n, int s, float *d_out, hipStream_t st)
{
    float *asum = nullptr;
    const size_t bytes = (size_t)n * lerf::HID * sizeof(float);
    NRF_HIP(scratch_take(reinterpret_cast<void **>(&asum), bytes, st));
    int rc = NRF_OK;              // every row of asum is stored by the wave that owns its ray: no zero fill
    if (rc == NRF_OK) {
        ProfScope prof(NRF_PROF_MLP, st);
        a.out = asum;
        rc = lerf::launch_lerf_split<4>(m, a, n * (int64_t)s, st);
        if (rc == NRF_OK) {
            hipLaunchKernelGGL(lerf::k_lerf_embed_split, dim3((unsigned)ceil_div(ceil_div(n, (int64_t)32), (int64_t)4)), dim3(256), 0, st, n, (const float *)asum,
                               reinterpret_cast<const lerf::half8 *>(m->d_packed_split), d_out);
            if (hipGetLastError() != hipSuccess) { set_error("k_lerf_embed_split launch failed"); rc = NRF_ERR_HIP; }
        }
    }
    (void)scratch_give(asum, st);          // stream-ordered: also on the error paths
    return rc;
}

}  // namespace nrf
