// mlp_nerf_split_mfma.hip -- NeRFImpl::forward (NeRF.cpp:41-126), the classic 8 x 256 MLP with view directions, on the gfx950 matrix
// cores at fp32-grade precision (NRF_PREC_F16_SPLIT).
//
// The reference evaluates this network in fp32 (torch::nn::Linear on fp32 tensors).  The plain fp16 matrix-core kernel (mlp_nerf_mfma.hip)
// rounds weights and activations to 11 significant bits: 58 dB against the oracle, 44 % of pixel values within 1e-4.  Here every fp32 quantity
// is carried as an UNEVALUATED SUM of two fp16 numbers, v = hi + lo with hi = f16(v), lo = f16(v - hi) (22 significant bits): the weights
// are split once at pack time, the activations when a D tile becomes the next layer's B fragments (one v_max, half a v_cvt_pk and one
// v_fma_mix per value), and a product is three matrix instructions, Wh.xh into the main accumulator and Wl.xh + Wh.xl into a second one
// that is added at the end of the tile (the dropped Wl.xl term is 2^-22 relative).  Same transposed formulation, same fragment permutation
// and the same merged views_linears_0 o feature_linear layer as the fp16 kernel (mlp_nerf_net.h).
//
// Resources.  A 256-wide activation vector of 32 points is 64 VGPRs as fp16, 128 as (hi, lo); a layer needs its input and its output:
// 256 registers for activations alone.  So ONE wave per SIMD (4 waves, 128 points per workgroup pass, 512 registers per wave: the unified
// VGPR + AGPR file), where the fp16 kernel runs two.  What hides latency inside the single wave is the two independent accumulator chains
// and the weight stream's look-ahead: the image (2.1 MB: hi and lo fragment of every k-step adjacent) is cut into 70 chunks of ONE neuron
// tile (<= 40 fragments = 40 KB), streamed L2 -> LDS by LDS-DMA through three buffers two chunks ahead (counted vmcnt + raw s_barrier, as in
// the fp16 kernel).  The DMA instructions of a chunk are issued one every other k-step, between the matrix instructions, instead of in a
// burst at the top: a piece costs its issuing wave ~60 cycles (MI355X_MICROARCH.md), which then overlap the 96 matrix-pipe cycles of the
// k-step in flight.
#include "mlp_nerf_net.h"

#include <utility>

namespace nrf {

constexpr int SNW = 4;                 // waves per workgroup: one per SIMD
constexpr int SNBLK = 32 * SNW;        // points per workgroup iteration
constexpr int SMAXF = 40;              // fragments (1 KB each) in the largest chunk: 20 k-steps x (hi, lo)

// one neuron tile per chunk; the image holds, per (layer, tile, k-step), the hi fragment then the lo fragment
struct NerfNetS {
    static constexpr int first_chunk(int l) { int n = 0; for (int i = 0; i < l; i++) n += NerfNet::tiles(i); return n; }
    static constexpr int total_chunks() { return first_chunk(NerfNet::NLAYER); }
    static constexpr int layer_of(int ci) { int l = 0; while (first_chunk(l + 1) <= ci) l++; return l; }
    static constexpr int chunk_frags(int ci) { return 2 * NerfNet::ks(layer_of(ci)); }
    static constexpr int chunk_off(int ci) { int n = 0; for (int i = 0; i < ci; i++) n += chunk_frags(i); return n; }
    static constexpr int total_frags() { return chunk_off(total_chunks()); }
};
static_assert(NerfNetS::total_chunks() == 70, "chunk count");
static_assert(NerfNetS::total_frags() == 2 * NerfNet::total_frags(), "the split image is twice the fp16 image");

// piece Q (0 .. pieces per wave) of chunk CI -> LDS buffer `dst`: wave w moves fragments w, w + SNW, ... (every chunk's fragment count is a multiple of SNW)
template <int CI, int Q>
__device__ __forceinline__ void stage_piece(half8 *__restrict__ dst, const half8 *__restrict__ packed, int wave, int lane)
{
    constexpr int ci = CI % NerfNetS::total_chunks();
    constexpr int nf = NerfNetS::chunk_frags(ci);
    static_assert(nf % SNW == 0, "fragments per chunk must divide by the wave count");
    if constexpr (Q * SNW < nf) {
        constexpr int base = NerfNetS::chunk_off(ci);
        // the fragment's address = SGPR base (its constant offset added on the scalar side, then made opaque) + lane * 16: the saddr form of the DMA.  With the offset
        // added after the opaque point the compiler forms a 64-bit per-lane address instead -- two v_lshl_add_u64 per DMA, ~1 070 per iteration of the classic kernel
        const half8 *pk = packed + (size_t)wave * 64;
        asm volatile("" : "+s"(pk));                          // opaque: the addresses derived from it cannot be hoisted out of the persistent loop (533 SGPR pairs would spill)
        pk += (size_t)(base + Q * SNW) * 64;
        asm volatile("" : "+s"(pk));                          // the offset is added HERE, on the scalar side (s_add_u32 / s_addc_u32)
        __builtin_amdgcn_global_load_lds(pk + lane, (__attribute__((address_space(3))) void *)(dst + (Q * SNW + wave) * 64), 16, 0, 0);
    }
}

template <int CI, int... Qs>
__device__ __forceinline__ void stage_all(half8 *__restrict__ dst, const half8 *__restrict__ packed, int wave, int lane, std::integer_sequence<int, Qs...>)
{
    (stage_piece<CI, Qs>(dst, packed, wave, lane), ...);
}

// (hi, lo) of two fp32 values, packed: hi = RNE(v), lo = RNE(v - hi) through one mixed-precision FMA each (see mlp_small_mfma.hip, split_pair)
__device__ __forceinline__ void nerf_split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
}

#ifdef NRF_NERF_TRACE
// diagnostic build only (tools/scratch/classic_trace.py): cycle stamps of wave 0 of every workgroup, summed per section: [0] tile loops, [2] end-of-chunk wait +
// barrier, [3] input encoding, [4] whole iterations, [5] iterations, [6] / [7] the 4-step / 16-step tile loops alone
__device__ unsigned long long g_nerf_trace[256 * 8];
#define NRF_STAMP() __builtin_readcyclecounter()
#endif

struct CtxS {
    half8 *wbuf;            // [3][SMAXF*64]
    const float *bias_s;    // LDS
    const half8 *packed;    // the weight image (wave-uniform)
    int lane, h, wave;
    int *cur;               // LDS buffer (0..2) holding the chunk being consumed; wave-uniform, advanced by every chunk
    f32x16 *accs;           // two accumulator tiles: chunk CI accumulates into accs[CI & 1] while accs[(CI & 1) ^ 1], the previous tile, is being converted
#ifdef NRF_NERF_TRACE
    unsigned long long *tr;
#endif
};

typedef uint32_t nerf_u32x4 __attribute__((ext_vector_type(4)));

// One conversion unit of a finished tile: values 8s + 2j, 8s + 2j + 1 (s = u >> 2, j = u & 3) of its accumulator -> ReLU -> the (hi, lo) words j of the next layer's
// B fragments 2 PT + s.  Seven vector instructions (two AGPR reads, two v_max, one v_cvt_pk, two v_fma_mix).
template <int PT, int NB>
__device__ __forceinline__ void nerf_conv_unit(const f32x16 &tile, int u, half8 (&tgt)[NB][2])
{
    const int s = u >> 2, j = u & 3;
    uint32_t hi, lo;
    nerf_split_pair(fmaxf(tile[8 * s + 2 * j], 0.0f), fmaxf(tile[8 * s + 2 * j + 1], 0.0f), hi, lo);
    nerf_u32x4 hv = __builtin_bit_cast(nerf_u32x4, tgt[2 * PT + s][0]), lv = __builtin_bit_cast(nerf_u32x4, tgt[2 * PT + s][1]);
    hv[j] = hi; lv[j] = lo;
    tgt[2 * PT + s][0] = __builtin_bit_cast(half8, hv); tgt[2 * PT + s][1] = __builtin_bit_cast(half8, lv);
}

// One chunk = neuron tile T of layer L.  w / dma_dst / bias_s are __restrict__ PARAMETERS on purpose (alias-scope metadata after inlining: this chunk's
// LDS reads do not touch the look-ahead's destination, so no vmcnt(0) is inserted before them -- see mlp_nerf_mfma.hip).
//
// What a single wave per SIMD has to hide by itself (cycle stamps of a diagnostic build, per 16-step tile = 1536 matrix-pipe cycles; numbers in DESIGN section 9):
//   * the conversion of a finished tile into the next layer's operands -- ~180 vector instructions = ~700 cycles during which the pipe sat idle when the
//     conversion followed its own tile.  Now the tile of chunk CI - 1 is converted DURING chunk CI: two accumulator tiles alternate, and the eight conversion
//     units of the previous tile are dealt out one per k-step behind this tile's matrix instructions (two per step on layer 0's four-step tiles).  The operands
//     it writes (fragments 2 PT, 2 PT + 1 of the next layer's input; for a layer's last tile, fragments 14 and 15 of THIS layer's input) are first read at
//     chained k-step >= 14, after the last unit;
//   * the LDS latency of the weight fragments: read two k-steps ahead through three register slots, with counted waits (the reads and waits are written out:
//     left alone the compiler puts a step's reads 32 pipe cycles before their use, or -- given the order -- waits for ALL outstanding reads, lgkmcnt(0));
//   * one accumulator per tile: the three products of a k-step go into the same tile, small terms first (a dependent chain of this instruction issues
//     back to back, MI355X_MICROARCH.md), which halves the AGPR reads and drops the additions of the former main + correction pair.
template <int L, int T, int NN, int NC, int NOUT>
__device__ __forceinline__ void nerf_chunk_body_s(const CtxS &cx, const half8 *__restrict__ w, half8 *__restrict__ dma_dst, const float *__restrict__ bias_s,
                                                  half8 (&bn)[NN][2], half8 (&bc)[NC][2], half8 (&bout)[NOUT][2], f32x16 &last)
{
    constexpr int KSN = NerfNet::ks_nat(L), KSC = NerfNet::ks_ch(L), KS = KSN + KSC;
    constexpr int CI = NerfNetS::first_chunk(L) + T;
    constexpr int BOFF = NerfNet::bias_off(L);
    constexpr int NTILES = NerfNet::tiles(L);
    constexpr bool NATF = NerfNet::nat_first(L);
    static_assert(KSN <= NN && KSC <= NC, "operand fragment arrays too small");
    // the previous chunk's tile, still unconverted in the other accumulator
    constexpr int PL = T > 0 ? L : L - 1;                                          // its layer (L = 0, T = 0: none -- the last layer's tile is never converted)
    constexpr int PT = T > 0 ? T - 1 : (L > 0 ? NerfNet::tiles(L > 0 ? L - 1 : 0) - 1 : 0);
    constexpr bool PEND = (T > 0 || L > 0) && PL < NerfNet::NLAYER - 1 && 2 * PT + 1 < 16;
    constexpr int UPS = KS >= 8 ? 1 : 2;                                            // conversion units per k-step
    static_assert(!PEND || 8 <= UPS * KS, "the previous tile's eight conversion units must fit this tile's k-steps");
    static_assert(!PEND || T > 0 || NC == 16, "a layer's last tile becomes fragments 14, 15 of the next layer's chained input");
    // chained fragments 2 PT, 2 PT + 1 must not be read before the last unit has written them: first chained k-step that reads them vs last unit's step
    static_assert(!PEND || T > 0 || (NATF ? KSN : 0) + 14 >= (8 + UPS - 1) / UPS, "conversion finishes too late for this layer's chained operands");
    f32x16 &acc = cx.accs[CI & 1];
    const f32x16 &prev = cx.accs[(CI & 1) ^ 1];
    // pieces of chunk CI + 2 this wave issues, spread over this chunk's k-steps
    constexpr int NQ = NerfNetS::chunk_frags((CI + 2) % NerfNetS::total_chunks()) / SNW;
    constexpr int EVERY = (KS / NQ) > 0 ? (KS / NQ) : 1;          // one piece every EVERY k-steps ...
    constexpr int LEAD = NQ > KS / EVERY ? NQ - KS / EVERY : 0;   // ... and what does not fit that way, at the top
#ifdef NRF_NERF_TRACE
    const unsigned long long t0_ = NRF_STAMP();
#endif
    stage_all<CI + 2>(dma_dst, cx.packed, cx.wave, cx.lane, std::make_integer_sequence<int, LEAD>{});
    half8 fa[3][2];
    const uint32_t waddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(w + cx.lane);
    auto read_pair = [&](int kk, int slot) {
        // the k-step's byte offset rides in the instruction (16-bit field; a chunk is < 64 KB) instead of a v_add_u32 per pair of reads (~990 per iteration)
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(fa[slot][0]), "=&v"(fa[slot][1]) : "v"(waddr), "i"(kk * 2048), "i"(kk * 2048 + 1024));
    };
    read_pair(0, 0);
    if (KS > 1) read_pair(1, 1);
    const float *bp = bias_s + BOFF + T * 32 + 4 * cx.h;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 bv = *reinterpret_cast<const float4 *>(bp + 8 * g);
        acc[4 * g + 0] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
    }
    int q = LEAD;
#pragma unroll
    for (int k = 0; k < KS; k++) {
        // counted waits: conservative against LDS / scalar loads the compiler issues itself (they only add to the number outstanding)
        if (k + 2 < KS) {
            read_pair(k + 2, (k + 2) % 3);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[k % 3][0]), "+v"(fa[k % 3][1]));
        } else if (k + 1 < KS) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[k % 3][0]), "+v"(fa[k % 3][1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[k % 3][0]), "+v"(fa[k % 3][1]));
        __builtin_amdgcn_sched_barrier(0);
        const half8 ah = fa[k % 3][0], al = fa[k % 3][1];
        half8 bh, bl;
        if (NATF) { bh = (k < KSN) ? bn[k < KSN ? k : 0][0] : bc[k >= KSN ? k - KSN : 0][0]; bl = (k < KSN) ? bn[k < KSN ? k : 0][1] : bc[k >= KSN ? k - KSN : 0][1]; }
        else { bh = (k < KSC) ? bc[k < KSC ? k : 0][0] : bn[k >= KSC ? k - KSC : 0][0]; bl = (k < KSC) ? bc[k < KSC ? k : 0][1] : bn[k >= KSC ? k - KSC : 0][1]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        if constexpr (PEND) {
#pragma unroll
            for (int uu = 0; uu < UPS; uu++) {
                const int u = k * UPS + uu;
                if (u < 8) {
                    if constexpr (T > 0) nerf_conv_unit<PT>(prev, u, bout);
                    else if constexpr (NC == 16) nerf_conv_unit<PT>(prev, u, bc);
                }
            }
        }
        if ((k % EVERY) == EVERY - 1 && q < NQ) {
            // the next piece of the look-ahead chunk, in the shadow of this k-step's matrix instructions (q is a compile-time value after unrolling)
            const int qq = q;
            switch (qq) {
#define NRF_PIECE(Q) case Q: stage_piece<CI + 2, Q>(dma_dst, cx.packed, cx.wave, cx.lane); break;
                NRF_PIECE(0) NRF_PIECE(1) NRF_PIECE(2) NRF_PIECE(3) NRF_PIECE(4) NRF_PIECE(5) NRF_PIECE(6) NRF_PIECE(7) NRF_PIECE(8) NRF_PIECE(9)
#undef NRF_PIECE
            }
            q++;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(NQ <= 10, "piece switch covers 10 pieces per wave");
    if (T == NTILES - 1 && L >= 8) last = acc;          // alpha (layer 8) and rgb (layer 9) are read from the accumulators
#ifdef NRF_NERF_TRACE
    const unsigned long long t1_ = NRF_STAMP();
#endif
    // End of the chunk: chunk CI + 1 (requested during chunk CI - 1) must have landed, chunk CI + 2 (requested during this one) may stay in flight.
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NQ) : "memory");
#ifdef NRF_NERF_TRACE
    cx.tr[0] += t1_ - t0_; cx.tr[2] += NRF_STAMP() - t1_;
    if (KS == 4) cx.tr[6] += t1_ - t0_;
    if (KS == 16) cx.tr[7] += t1_ - t0_;
#endif
}

template <int L, int T, int NN, int NC, int NOUT>
__device__ __forceinline__ void nerf_chunk_s(const CtxS &cx, half8 (&bn)[NN][2], half8 (&bc)[NC][2], half8 (&bout)[NOUT][2], f32x16 &last)
{
    // chunk CI + 2 -> the buffer chunk CI - 1 was consumed from (every wave is past the barrier that ended it)
    const int cur = *cx.cur;
    nerf_chunk_body_s<L, T>(cx, cx.wbuf + cur * (SMAXF * 64), cx.wbuf + (cur == 0 ? 2 : cur - 1) * (SMAXF * 64), cx.bias_s, bn, bc, bout, last);
    *cx.cur = cur == 2 ? 0 : cur + 1;
}

template <int L, int NN, int NC, int NOUT, int... Ts>
__device__ __forceinline__ void nerf_layer_seq_s(const CtxS &cx, half8 (&bn)[NN][2], half8 (&bc)[NC][2], half8 (&bout)[NOUT][2], f32x16 &last,
                                                 std::integer_sequence<int, Ts...>)
{
    (nerf_chunk_s<L, Ts>(cx, bn, bc, bout, last), ...);
}

template <int L, int NN, int NC, int NOUT>
__device__ __forceinline__ void nerf_layer_s(const CtxS &cx, half8 (&bn)[NN][2], half8 (&bc)[NC][2], half8 (&bout)[NOUT][2], f32x16 &last)
{
    nerf_layer_seq_s<L>(cx, bn, bc, bout, last, std::make_integer_sequence<int, NerfNet::tiles(L)>{});
}

__device__ __forceinline__ void split_f32(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

template <bool FUSED>
__global__ void __launch_bounds__(64 * SNW, 1)
k_mlp_nerf_split(int64_t npts, NerfInput in, const half8 *__restrict__ packed, const float *__restrict__ biases, float *__restrict__ out, int out_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ __attribute__((aligned(16))) float bias_s[NBIAS];          // its own LDS object: reads provably clear of the DMA destinations in `smem`
    half8 *wbuf = reinterpret_cast<half8 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < NBIAS; i += 64 * SNW) bias_s[i] = biases[i];
    stage_all<0>(wbuf, packed, wave, lane, std::make_integer_sequence<int, NerfNetS::chunk_frags(0) / SNW>{});
    stage_all<1>(wbuf + SMAXF * 64, packed, wave, lane, std::make_integer_sequence<int, NerfNetS::chunk_frags(1) / SNW>{});
    __syncthreads();                               // vmcnt(0): chunks 0 and 1 are in place
    int cur = 0;
    const int64_t nblocks = (npts + SNBLK - 1) / SNBLK;
    f32x16 accs[2];
#ifdef NRF_NERF_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
#ifdef NRF_NERF_TRACE
        CtxS cx{wbuf, bias_s, packed, lane, h, wave, &cur, accs, tr};
        const unsigned long long ti0_ = NRF_STAMP();
#else
        CtxS cx{wbuf, bias_s, packed, lane, h, wave, &cur, accs};
#endif
        const int64_t q_raw = blk * SNBLK + wave * 32 + r;
        const int64_t q = q_raw < npts ? q_raw : npts - 1;          // clamp loads; the store is guarded
        half8 pe[4][2];                                             // positions: layer 0 and the skip layer 5
        {
            float px[3] = {0.0f, 0.0f, 0.0f};
            const float *row = nullptr;
            if constexpr (FUSED) {
                if (in.x) { px[0] = in.x[q * 3]; px[1] = in.x[q * 3 + 1]; px[2] = in.x[q * 3 + 2]; }      // explicit sample points (stochastic branches)
                else {
                    const float *rp = in.rays + (int64_t)((uint32_t)q / (uint32_t)in.s) * in.ray_stride;
                    const float zz = in.z[q];
                    px[0] = rp[0] + rp[3] * zz; px[1] = rp[1] + rp[4] * zz; px[2] = rp[2] + rp[5] * zz;
                }
            } else row = in.x + q * in.x_stride;
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float v;
                    if constexpr (FUSED) {
                        // feature k = 16s + 8h + j: [x(3) | per frequency f: sin(x 2^f)(3), cos(x 2^f)(3)]; both lane halves' indices are compile-time constants and h selects
                        constexpr auto arg_axis = [](int k) { return k < 3 ? k : ((k - 3) % 6) % 3; };
                        constexpr auto arg_freq = [](int k) { return k < 3 ? 0 : (k - 3) / 6; };
                        constexpr auto kind = [](int k) { return k < 3 ? 0 : k >= 63 ? 3 : (((k - 3) % 6) < 3 ? 1 : 2); };   // 0 raw, 1 sin, 2 cos, 3 pad
                        const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;
                        const float a0 = px[arg_axis(k0)] * __builtin_ldexpf(1.0f, arg_freq(k0));
                        const float a1 = px[arg_axis(k1 < 63 ? k1 : 0)] * __builtin_ldexpf(1.0f, arg_freq(k1 < 63 ? k1 : 0));
                        float sn, cs;
                        nrf_sincosf(h ? a1 : a0, &sn, &cs);
                        const int kd0 = kind(k0), kd1 = kind(k1);
                        const float v0 = kd0 == 0 ? px[arg_axis(k0)] : kd0 == 1 ? sn : kd0 == 2 ? cs : 0.0f;
                        const float v1 = kd1 == 0 ? px[arg_axis(k1 < 63 ? k1 : 0)] : kd1 == 1 ? sn : kd1 == 2 ? cs : 0.0f;
                        v = h ? v1 : v0;
                    } else v = row[16 * s + 8 * h + j];          // index 63 is the first view feature: its weight column is zero
                    _Float16 vh, vl;
                    split_f32(v, vh, vl);
                    pe[s][0][j] = vh; pe[s][1][j] = vl;
                }
        }
        half8 ba[16][2], bb[16][2], none[1][2];
        f32x16 last;
#ifdef NRF_NERF_TRACE
        asm volatile("s_nop 0" : "+v"(pe[3][1]));
        tr[3] += NRF_STAMP() - ti0_;
#endif
        nerf_layer_s<0>(cx, pe, none, ba, last);
        nerf_layer_s<1>(cx, none, ba, bb, last);
        nerf_layer_s<2>(cx, none, bb, ba, last);
        nerf_layer_s<3>(cx, none, ba, bb, last);
        nerf_layer_s<4>(cx, none, bb, ba, last);
        nerf_layer_s<5>(cx, pe, ba, bb, last);
        nerf_layer_s<6>(cx, none, bb, ba, last);
        nerf_layer_s<7>(cx, none, ba, bb, last);
        {
            half8 vw[2][2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if constexpr (FUSED) {
                    const int64_t doff = (int64_t)((uint32_t)q / (uint32_t)in.s) * 32 + 16 * s + 8 * h;
                    vw[s][0] = *reinterpret_cast<const half8 *>(in.dirs + doff);
                    vw[s][1] = *reinterpret_cast<const half8 *>(in.dirs_lo + doff);
                } else {
                    const float *row = in.x + q * in.x_stride;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int k = 16 * s + 8 * h + j;
                        _Float16 vh, vl;
                        split_f32(k < 27 ? row[63 + k] : 0.0f, vh, vl);
                        vw[s][0][j] = vh; vw[s][1][j] = vl;
                    }
                }
            }
            nerf_layer_s<8>(cx, vw, bb, ba, last);          // views_linears_0 o feature_linear -> ba[0..7] (ReLU); tile 4 row 0 = alpha (from the accumulators)
        }
        const float alpha = last[0];
        nerf_layer_s<9>(cx, none, ba, bb, last);             // rgb_linear: one tile, rows 0..2 read from the accumulators
        if (h == 0 && q_raw < npts) {
            float *o = out + q_raw * out_stride;
            o[0] = last[0]; o[1] = last[1]; o[2] = last[2]; o[3] = alpha;
        }
#ifdef NRF_NERF_TRACE
        tr[4] += NRF_STAMP() - ti0_; tr[5] += 1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last two chunks' look-ahead requests are still in flight
#ifdef NRF_NERF_TRACE
    if (tid == 0 && FUSED) for (int i = 0; i < 8; i++) g_nerf_trace[blockIdx.x * 8 + i] += tr[i];      // no other code reads this buffer
#endif
}

#ifdef NRF_NERF_TRACE
extern "C" NRF_API int nrf_dbg_nerf_trace(unsigned long long *host_out, int reset)
{
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nerf_trace), sizeof(unsigned long long) * 256 * 8) != hipSuccess) return NRF_ERR_HIP;
    if (reset) { static unsigned long long z[256 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_nerf_trace), z, sizeof(z)) != hipSuccess) return NRF_ERR_HIP; }
    return NRF_OK;
}
#endif

// per-ray PE(4) of the view direction as (hi, lo) fp16 rows [n, 32] (27 features, zero padded): the layer-8 operand of the fused path
__global__ void k_dirs_pe_split(int64_t n, const float *__restrict__ rays, int stride, __half *__restrict__ out_hi, __half *__restrict__ out_lo)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * 32) return;
    const int64_t i = gid >> 5;
    const int k = (int)(gid & 31);
    const float *dp = rays + i * stride + 8;
    float v = 0.0f;
    if (k < 3) v = dp[k];
    else if (k < 27) {
        const int f = (k - 3) / 6, q = (k - 3) - 6 * f;
        float sn, cs;
        nrf_sincosf(dp[q < 3 ? q : q - 3] * __builtin_ldexpf(1.0f, f), &sn, &cs);
        v = q < 3 ? sn : cs;
    }
    const __half hv = __float2half_rn(v);
    out_hi[gid] = hv;
    out_lo[gid] = __float2half_rn(v - __half2float(hv));
}

int launch_dirs_pe_split(const float *rays, int stride, int64_t n, __half *out_hi, __half *out_lo, hipStream_t st)
{
    if (n == 0) return NRF_OK;
    hipLaunchKernelGGL(k_dirs_pe_split, dim3((unsigned)ceil_div(n * 32, 256)), dim3(256), 0, st, n, rays, stride, out_hi, out_lo);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

int mlp_nerf_split_available(const nrf_mlp *m) { return m && m->family == MLP_NERF && m->d_packed_split != nullptr; }

int mlp_nerf_forward_split(const nrf_mlp *m, const NerfInput &in, bool fused, int64_t p, float *out, int os, hipStream_t st)
{
    if (!mlp_nerf_split_available(m)) {
        set_error("NRF_PREC_F16_SPLIT: this NeRF shape is outside the built matrix-core family (8 x 256, skip 4, PE(10)/PE(4), view directions); use NRF_PREC_F32");
        return NRF_ERR_UNSUPPORTED;
    }
    const size_t img_bytes = (size_t)NerfNetS::total_frags() * 1024;
    if (m->packed_split_bytes != img_bytes + NBIAS * sizeof(float)) { set_error("internal: classic NeRF split image is %zu bytes, kernel expects %zu", m->packed_split_bytes, img_bytes + NBIAS * sizeof(float)); return NRF_ERR_INVALID_ARG; }
    const size_t lds = (size_t)3 * SMAXF * 1024;          // + the static bias array
    const int64_t nblocks = ceil_div(p, SNBLK);
    const unsigned grid = (unsigned)(nblocks < 256 ? nblocks : 256);       // one persistent workgroup per CU
    static PerDeviceOnce attr_set;          // idempotent one-time setup per device (common.h)
    if (attr_set.needed()) {
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_nerf_split<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        NRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_nerf_split<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.done();
    }
    const half8 *packed = reinterpret_cast<const half8 *>(m->d_packed_split);
    const float *biases = reinterpret_cast<const float *>(static_cast<const char *>(m->d_packed_split) + img_bytes);
    if (fused) hipLaunchKernelGGL(k_mlp_nerf_split<true>, dim3(grid), dim3(64 * SNW), lds, st, p, in, packed, biases, out, os);
    else hipLaunchKernelGGL(k_mlp_nerf_split<false>, dim3(grid), dim3(64 * SNW), lds, st, p, in, packed, biases, out, os);
    NRF_LAUNCH_CHECK();
    return NRF_OK;
}

// generic boundary: fp32 rows [p, 90]
int mlp_nerf_forward_split_rows(const nrf_mlp *m, const float *x, int xs, int64_t p, float *out, int os, hipStream_t st)
{
    NerfInput in{x, xs, nullptr, 0, nullptr, 1, nullptr, nullptr};
    return mlp_nerf_forward_split(m, in, false, p, out, os, st);
}

// renderer fast path: points from (rays, z) -- or explicit `pts` [p,3] when not NULL --, PE in registers, per-ray (hi, lo) fp16 direction encodings -> raw [p,4]
int mlp_nerf_forward_split_fused(const nrf_mlp *m, const float *pts, const float *rays, int ray_stride, const float *z, int s, const __half *dirs, const __half *dirs_lo,
                                 int64_t p, float *out, hipStream_t st)
{
    ProfScope prof(NRF_PROF_MLP, st);
    NerfInput in{pts, 3, rays, ray_stride, z, s, dirs, dirs_lo};
    return mlp_nerf_forward_split(m, in, true, p, out, 4, st);
}

}  // namespace nrf
