// MFMA GEMM kernels for gfx950 (MI355X): ga_gemm (NT, forward + dgrad) and ga_wgrad (TN, reduction over rows).
//
// Design (see DESIGN.md "GEMM family"):
//  * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles),
//    K streamed in 128-byte slabs (64 bf16 / 32 fp32), register-staged global->LDS with a 2-deep LDS ring:
//    the global loads of slab t+1 are in flight while slab t is multiplied (one barrier per slab).
//  * bf16: v_mfma_f32_16x16x32_bf16;  fp32 (parity math mode): v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
//  * LDS images are XOR-swizzled at 16-byte granularity so that the ds_read_b128 fragment reads
//    (16 rows x one 16-B chunk per 16-lane group) are bank-conflict free.
//  * The weight fragment is passed as the MFMA "A" operand so each lane ends up with 4 CONSECUTIVE output
//    columns of one output row; the tile is then staged through LDS (fp32) and written with 16-byte stores,
//    which is also where bias / GELU / GELU' / DropPath row-scale / residual / ReLU / BatchNorm column
//    statistics are fused.
//  * wgrad reads both operands "k-strided" (rows = reduction index): bf16 uses ds_read_b64_tr_b16, the gfx950
//    transposing LDS read, on a [rows][128] image whose 32-byte pieces are XOR-swizzled conflict-free.
//  * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each take a contiguous chunk of the tile list so that
//    workgroups sharing an operand panel hit the same L2.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace {

int num_cus() {
    static int n = [] {
        int c = 256;
        ga_device_info(&c, nullptr, nullptr);
        return c;
    }();
    return n;
}

constexpr int kThreads = 256;
constexpr int kRowBytes = 128;          // bytes of K per LDS row (NT kernel)

// bijective XCD-aware remap of a linear workgroup id (guide T1): blocks b, b+8, b+16.. share an XCD
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------------------
// A-operand gather (implicit im2col).  One RowCtx per staged row; one chunk = 16 bytes of T.
// ------------------------------------------------------------------------------------------------
struct RowCtx {
    long base;   // element offset of the row start (plain) / of the patch origin pixel (gather kinds)
    int y, x;    // output pixel coords (CONV3)
    bool valid;
};

__device__ __forceinline__ RowCtx make_row(int kind, long m, long M, long ld, int H, int W, int C) {
    RowCtx r;
    r.valid = m < M;
    r.y = r.x = 0;
    r.base = 0;
    if (!r.valid) return r;
    if (kind == GA_A_PLAIN) {
        r.base = m * ld;
    } else if (kind == GA_A_PATCH2) {
        const int OW = W >> 1, OH = H >> 1;
        const int ox = (int)(m % OW);
        const long t = m / OW;
        const int oy = (int)(t % OH);
        const long b = t / OH;
        r.base = ((b * H + 2 * oy) * W + 2 * ox) * (long)C;
    } else if (kind == GA_A_CONV3 || kind == GA_A_NEIGH2) {
        const int x = (int)(m % W);
        const long t = m / W;
        const int y = (int)(t % H);
        const long b = t / H;
        r.base = b * H * W;  // pixel index of the image start
        r.y = y;
        r.x = x;
    } else if (kind == GA_A_CONV3S2) {   // rows = output pixels of the stride-2 conv; (y, x) = centre tap in the input
        const int OW = (W + 1) >> 1, OH = (H + 1) >> 1;
        const int ox = (int)(m % OW);
        const long t = m / OW;
        const int oy = (int)(t % OH);
        const long b = t / OH;
        r.base = b * H * W;
        r.y = 2 * oy;
        r.x = 2 * ox;
    } else {  // GA_A_STEM4_NCHW: fp32 NCHW input, C == 3 planes
        const int OW = W >> 2, OH = H >> 2;
        const int ox = (int)(m % OW);
        const long t = m / OW;
        const int oy = (int)(t % OH);
        const long b = t / OH;
        r.base = ((b * C) * H + 4 * oy) * (long)W + 4 * ox;  // float index of (b, c=0, 4oy, 4ox)
    }
    return r;
}

// per-thread decomposition of the chunk's first k index (same for all rows a thread stages)
struct KCtx {
    long off;    // element offset added to RowCtx.base
    int dy, dx;  // tap offset (CONV3)
    bool valid;
};

__device__ __forceinline__ KCtx make_k(int kind, int k, int K, int H, int W, int C) {
    KCtx c;
    c.valid = k < K;
    c.off = 0;
    c.dy = c.dx = 0;
    if (!c.valid) return c;
    if (kind == GA_A_PLAIN) {
        c.off = k;
    } else if (kind == GA_A_PATCH2) {
        const int tap = k / C, ch = k - tap * C;
        c.off = ((long)(tap >> 1) * W + (tap & 1)) * C + ch;
    } else if (kind == GA_A_CONV3 || kind == GA_A_CONV3S2) {
        const int tap = k / C, ch = k - tap * C;
        c.dy = tap / 3 - 1;
        c.dx = tap % 3 - 1;
        c.off = ((long)c.dy * W + c.dx) * C + ch;
    } else if (kind == GA_A_NEIGH2) {    // taps (0,0) (0,1) (1,0) (1,1)
        const int tap = k / C, ch = k - tap * C;
        c.dy = tap >> 1;
        c.dx = tap & 1;
        c.off = ((long)c.dy * W + c.dx) * C + ch;
    } else {  // stem: k = (c, ky, kx)
        const int ch = k >> 4, ky = (k >> 2) & 3;
        c.off = ((long)ch * H + ky) * W;  // kx == 0 at a chunk start
    }
    return c;
}

template <typename T>
__device__ __forceinline__ uint4 load_chunk(int kind, const void* base, const RowCtx& r, const KCtx& k, int H, int W,
                                            int C) {
    uint4 z = make_uint4(0, 0, 0, 0);
    if (!(r.valid && k.valid)) return z;
    if (kind == GA_A_CONV3 || kind == GA_A_CONV3S2 || kind == GA_A_NEIGH2) {
        const int yy = r.y + k.dy, xx = r.x + k.dx;
        if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) return z;
        const long pix = r.base + (long)r.y * W + r.x;
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(base) + pix * C + k.off);
    }
    if (kind == GA_A_STEM4_NCHW) {
        const float* p = reinterpret_cast<const float*>(base) + r.base + k.off;
        const float4 a = *reinterpret_cast<const float4*>(p);
        if constexpr (sizeof(T) == 4) {
            return make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w));
        } else {
            const float4 b = *reinterpret_cast<const float4*>(p + W);  // next ky row
            return make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
        }
    }
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(base) + r.base + k.off);
}

template <typename T> __device__ __forceinline__ uint4 act_chunk(uint4 v, int act) {
    if (act == GA_ACT_NONE) return v;
    if constexpr (sizeof(T) == 4) {
        float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = act == GA_ACT_GELU ? gelu_f(f[i]) : fmaxf(f[i], 0.f);
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    } else {
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
            lo = act == GA_ACT_GELU ? gelu_f(lo) : fmaxf(lo, 0.f);
            hi = act == GA_ACT_GELU ? gelu_f(hi) : fmaxf(hi, 0.f);
            w[i] = pack2bf(lo, hi);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// one LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KiB) (dst is
// wave-uniform, carried in M0).  Written as asm so that hipcc does not count it: a builtin glds makes every later
// ds_read wait vmcnt(0) (possible alias), which would drain the prefetch it exists for.  The kernel waits itself.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// two pieces with one M0 set-up: the instruction offset is added to the global address AND to the LDS address, so the second
// piece (LDS dst + 1024) takes its global address minus 1024
__device__ __forceinline__ void glds16x2(const void* g0, const void* g1, unsigned lds_dst) {
    unsigned keep;
    const char* g1m = static_cast<const char*>(g1) - 1024;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g0), "v"(g1m), "s"(lds_dst)
        : "memory");
}

// the same through a raw buffer resource: a 32-bit byte offset per lane + a scalar offset (the K advance) instead of a 64-bit
// address per lane -- half the address traffic of the DMA issue.  Offsets >= the resource's num_records read as zeros.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_s;
__device__ __forceinline__ void blds16x2(u32x4_s rsrc, unsigned voff0, unsigned voff1, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    const unsigned v1m = voff1 - 1024;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %5 offen lds\n\t"
        "buffer_load_dwordx4 %2, %3, %5 offen offset:1024 lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff0), "v"(v1m), "s"(rsrc), "s"(lds_dst), "s"(soff)
        : "memory");
}
__device__ __forceinline__ void blds16(u32x4_s rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ u32x4_s make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    u32x4_s r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}

// lanes whose source row / k chunk is out of range fetch zeros from here
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

// ================================================================================================
// NT kernel.  Tile = 128 rows x BN columns (BN = 32*TNW = 64 / 96 / 128), 256 threads = 4 waves as 2(M) x 2(N),
// each wave 64 x 16*TNW.  One 128-byte K slab of A and B lives in LDS (single buffer, <= 32 KiB) while the NEXT
// slab is already in flight in registers; the fp32 result tile is staged through the same LDS in two 64-row
// halves (<= 34 KiB).  Small LDS + <= 128 VGPRs (TNW <= 3) give 4 resident workgroups per CU, i.e. >= 100 KiB of
// loads in flight per CU -- these GEMMs (K, N <= 3072, M ~ 10^6) are HBM-latency bound, not MFMA bound.
// ================================================================================================
template <int TNW, int NWM> struct NTCfg {
    static constexpr int NTHR = 128 * NWM;                    // NWM x 2 waves
    static constexpr int BM = 64 * NWM;                       // 128 (4 waves) or 256 (8 waves: 2.7x less operand traffic per FLOP)
    static constexpr int BN = 32 * TNW;
    static constexpr int A_BYTES = BM * kRowBytes;
    static constexpr int RSTEP = NTHR / 8;                    // rows staged per pass of the workgroup
    static constexpr int NB = BN * 8 / NTHR > 0 ? BN * 8 / NTHR : 1;   // B chunks per thread per slab (register staging)
    static constexpr int LDCS = BN + 4;                       // padded fp32 row stride of the staged 64-row piece
    static constexpr int SMEM_AB = A_BYTES + BN * kRowBytes;
    static constexpr int SMEM_C = 64 * LDCS * 4;
    static constexpr int SMEM = SMEM_AB > SMEM_C ? SMEM_AB : SMEM_C;
    static constexpr int P8 = BN / 8;                         // 8-column pieces per row
    static constexpr int RG = NTHR / P8;                      // row groups in the coalesced store phase
    // LDS-DMA form: [fp32 staging | ring of NS slots].  With 128-column tiles of 8 waves the staging area shrinks to
    // 32 unpadded (XOR-swizzled) rows = 16 KiB and THREE slots fit into the 160 KiB of a CU: two slabs (96 KiB) in
    // flight while one is multiplied.  Otherwise two slots behind the padded 64-row staging area (which must also hold
    // the [2][RG][BN] column-sum scratch).
    static constexpr bool DMA3 = TNW == 4 && NWM == 4;
    static constexpr bool SWZ = TNW == 4 || TNW == 8;         // 128 / 256-column tiles: XOR-swizzled staging pieces of 16 KiB
    static constexpr int NS = DMA3 ? 3 : 2;
    static constexpr int PR = TNW == 8 ? 16 : (SWZ ? 32 : 64);   // rows per staged piece (LDS-DMA form)
    static constexpr int CSUM_BYTES = 2 * RG * BN * 4;
    // 128 x 128 / 4 waves: 16 KiB + two 32 KiB slots = 80 KiB -> TWO workgroups per CU (one's epilogue under the other's K loop)
    // 256 x 256 / 8 waves (TNW = 8): 16 KiB + two 64 KiB slots = 144 KiB, half the operand bytes per FLOP of 128 x 128
    static constexpr int STAGE_DMA = SWZ ? PR * BN * 4 : (SMEM_C > CSUM_BYTES ? SMEM_C : CSUM_BYTES);
    static constexpr int RING0 = (STAGE_DMA + 1023) & ~1023;
    static constexpr int SMEM_DMA = RING0 + NS * SMEM_AB;
    static constexpr int NW = 2 * NWM;                        // waves
    static constexpr int NBP = (4 * TNW + NW - 1) / NW;       // B pieces (8 rows) per wave, the last may be idle
};

// EPI >= 0 fixes the epilogue at compile time (dead paths are not even emitted: the all-runtime generic kernel is
// ~14k instructions, most of them never executed by the hot launches, and thrashes the instruction cache):
//   0 plain (bias, optional column sums)      1 fc1: bias + GELU, second output GELU'
//   2 fc2: bias (+ DropPath row scale) + residual      3 dgrad2: multiply by the stored GELU' (+ column sums)
enum { EPI_GENERIC = -1, EPI_PLAIN = 0, EPI_FC1 = 1, EPI_FC2 = 2, EPI_DG2 = 3 };

// DMA (bf16, plain A): the K slabs are brought in by LDS-DMA (glds16) into a 2-slot ring that sits BEHIND the fp32
// staging area of the epilogue, one barrier per slab; the stream of slabs runs across tile boundaries, so the first
// slab of the next tile lands while this tile's epilogue runs.  No staging registers and no ds_write pass: with
// register staging the 8 ds_write_b128 per thread and slab (~79 B/clk/CU) cost more LDS time than the MFMAs take.
// 96-column tiles without the epilogue prefetch fit 168 VGPRs without spilling: three resident workgroups per CU
// instead of two (same-box A/B: -0.2 ms/step; the 128-column forms spill at that bound and lose)
template <typename T, int TNW, int NWM, bool PLAIN, bool PRE, int EPI, bool DMA = false>
__global__ __launch_bounds__(128 * NWM, (NWM == 2 && !DMA && !PRE && EPI >= 0 && sizeof(T) == 2 && TNW == 3) ? 3 : (NWM == 2 && DMA) ? 2 : 1) void gemm_nt_kernel(const ga_gemm_desc d) {
#define F_GELU (EPI < 0 ? d.act == GA_ACT_GELU : EPI == EPI_FC1)
#define F_RELU (EPI < 0 ? d.act == GA_ACT_RELU : false)
#define F_C2 (EPI < 0 ? d.C2 != nullptr : (EPI == EPI_FC1 && d.C2 != nullptr))   // eval: fc1 without the GELU' output
#define F_H (EPI < 0 ? Hb != nullptr : EPI == EPI_DG2)
#define F_HDERIV (EPI < 0 ? d.h_is_deriv != 0 : true)
#define F_RS (EPI < 0 ? d.rowscale != nullptr : (EPI == EPI_FC2 && d.rowscale != nullptr))
#define F_R (EPI < 0 ? Rb != nullptr : EPI == EPI_FC2)
#define F_RELUA (EPI < 0 ? d.relu_after != 0 : false)
#define F_CSUM (EPI < 0 ? d.colsum != nullptr : ((EPI == EPI_DG2 || EPI == EPI_PLAIN) && d.colsum != nullptr))
#define F_UNPATCH (EPI < 0 ? d.c_kind == GA_C_UNPATCH2 : false)
#define F_CF32 (EPI < 0 ? d.c_f32 != 0 : false)
#define F_AACT (EPI < 0 ? d.a_act : GA_ACT_NONE)
    using CF = NTCfg<TNW, NWM>;
    constexpr int EPC = elt<T>::EPC;
    constexpr int BK = kRowBytes / (int)sizeof(T);
    constexpr int BN = CF::BN;
    static_assert(!DMA || (PLAIN && sizeof(T) == 2), "the LDS-DMA form needs plain bf16 operands");
    static_assert(DMA || (CF::BN * 8) % CF::NTHR == 0, "B slab must split evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RING0 = DMA ? CF::RING0 : 0;
    unsigned char* As = smem + RING0;
    unsigned char* Bs = As + CF::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = DMA ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (d.N + BN - 1) / BN, tiles_m = (d.M + CF::BM - 1) / CF::BM;
    const int nwg = tiles_n * tiles_m;
    const int z = blockIdx.z;
    const int za = d.a_batch_mod > 0 ? z % d.a_batch_mod : z;

    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(d.A) +
                              (d.a_kind == GA_A_STEM4_NCHW ? 4 : (long)sizeof(T)) * za * d.strideA;
    const T* Bb = reinterpret_cast<const T*>(d.B) + z * d.strideB;

    // ---- staging roles: thread stages rows r0+RSTEP*i of A (i<4) and of B (i<NB), chunk column kc
    const int kc = tid & 7, r0 = tid >> 3;
    const int nk = (d.K + BK - 1) / BK;
    RowCtx arow[PLAIN ? 1 : 4];   // gather kinds keep a full context per staged row; PLAIN only needs m0
    int m0 = 0, n0 = 0;
    // PERSISTENT over output tiles: workgroup b takes tiles b, b+G, b+2G, ... (through the XCD-aware remap, so
    // the workgroups of one XCD walk neighbouring tiles and share A panels in its L2); the first K slab of the
    // NEXT tile is already in flight while the current tile's epilogue runs.
    auto set_tile = [&](int vt) {
        const int bid = xcd_remap(vt, nwg);
        const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
        m0 = tile_m * CF::BM;
        n0 = tile_n * BN;
        if constexpr (!PLAIN) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                arow[i] = make_row(d.a_kind, (long)m0 + r0 + CF::RSTEP * i, d.M, d.lda, d.a_H, d.a_W, d.a_C);
        }
    };

    uint4 ra[4], rb[CF::NB];
    auto g_load = [&](int kt) {
        const int k = kt * BK + kc * EPC;
        const bool kv = k < d.K;
        if constexpr (PLAIN) {
            const T* Ap = reinterpret_cast<const T*>(Ab) + k;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = (long)m0 + r0 + CF::RSTEP * i;
                ra[i] = (kv && m < d.M) ? *reinterpret_cast<const uint4*>(Ap + m * d.lda) : make_uint4(0, 0, 0, 0);
            }
        } else {
            const KCtx kx = make_k(d.a_kind, k, d.K, d.a_H, d.a_W, d.a_C);
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = load_chunk<T>(d.a_kind, Ab, arow[i], kx, d.a_H, d.a_W, d.a_C);
        }
#pragma unroll
        for (int i = 0; i < CF::NB; ++i) {
            const long n = (long)n0 + r0 + CF::RSTEP * i;
            rb[i] = (kv && n < d.N) ? *reinterpret_cast<const uint4*>(Bb + n * d.ldb + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto s_store = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + CF::RSTEP * i;
            *reinterpret_cast<uint4*>(As + row * kRowBytes + ((kc ^ (row & 7)) << 4)) = act_chunk<T>(ra[i], F_AACT);
        }
#pragma unroll
        for (int i = 0; i < CF::NB; ++i) {
            const int row = r0 + CF::RSTEP * i;
            *reinterpret_cast<uint4*>(Bs + row * kRowBytes + ((kc ^ (row & 7)) << 4)) = rb[i];
        }
    };

    // ---- LDS-DMA roles: wave w brings A pieces 4w..4w+3 and B pieces w, w+NW, ... (a piece = 8 rows x 128 B);
    // lane: row lane>>3 of the piece, LDS chunk lane&7, which holds SOURCE chunk (lane&7) ^ (lane>>3)
    // (through buffer resources: a 32-bit byte offset per lane, fixed for the tile, + the slab's K offset as the scalar offset;
    // rows beyond M / N and chunks beyond K carry an offset outside the resource and read as zeros)
    unsigned aoff[DMA ? 4 : 1], boff[DMA ? CF::NBP : 1];
    constexpr unsigned kOob = 0x80000000u;
    const int dchunk = ((lane & 7) ^ (lane >> 3)) * 8;
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + RING0;
    const u32x4_s rsA = make_rsrc(Ab, kOob), rsB = make_rsrc(Bb, kOob);
    const int ktail = d.K % BK;                                  // > 0: the last slab is ragged
    auto dma_rows = [&](int tm0, int tn0) {
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = (long)tm0 + (wave * 4 + i) * 8 + (lane >> 3);
                aoff[i] = m < d.M ? (unsigned)(m * d.lda + dchunk) * 2u : kOob;
            }
#pragma unroll
            for (int i = 0; i < CF::NBP; ++i) {
                const long n = (long)tn0 + (i * CF::NW + wave) * 8 + (lane >> 3);
                boff[i] = n < d.N ? (unsigned)(n * d.ldb + dchunk) * 2u : kOob;
            }
        }
    };
    auto dma_issue = [&](int kt, int slot) {
        if constexpr (DMA) {
            const bool kdead = ktail && kt == nk - 1 && dchunk >= ktail;      // this lane's chunk of the ragged last slab
            const unsigned soff = (unsigned)kt * (unsigned)kRowBytes;
            const unsigned dst = lds_ring + slot * CF::SMEM_AB;
#pragma unroll
            for (int i = 0; i < 4; ++i) blds16(rsA, kdead ? kOob : aoff[i], soff, dst + (wave * 4 + i) * 1024);
#pragma unroll
            for (int i = 0; i < CF::NBP; ++i) {
                const int p = i * CF::NW + wave;
                if (p < 4 * TNW) blds16(rsB, kdead ? kOob : boff[i], soff, dst + CF::A_BYTES + p * 1024);
            }
        }
    };

    const T* Hb = d.H ? reinterpret_cast<const T*>(d.H) + z * d.strideH : nullptr;
    const T* Rb = d.R ? reinterpret_cast<const T*>(d.R) + z * d.strideR : nullptr;
    float* Cs = reinterpret_cast<float*>(smem);
    const int c8 = tid % CF::P8, rg = tid / CF::P8;   // this thread's 8-column piece / first row of the half
    const bool t_active = rg < CF::RG;

    // Epilogue operand prefetch (bf16): the GELU'-source H (dgrad) or the residual R (forward) pieces this thread
    // will need are requested at the START of the last K slab, so their HBM latency hides under the MFMAs and the
    // LDS staging instead of being paid once per row in the store loop.
    constexpr int NR = (64 + CF::RG - 1) / CF::RG;
    constexpr bool kPre = PRE && sizeof(T) == 2;
    uint4 pre[kPre ? NWM : 1][kPre ? NR : 1];
    const T* Pb = F_H ? Hb : Rb;
    const long ldp = F_H ? d.ldh : d.ldr;

    auto pre_sel = [&](int hf, int it) -> const uint4& { return pre[kPre ? hf : 0][kPre ? it : 0]; };

    int vt = blockIdx.x;
    set_tile(vt);
    // LDS-DMA form: the slabs of all the workgroup's tiles form ONE stream; the issue side runs NS-1 slabs ahead of the
    // consuming side, across tile boundaries (issue-side state: tile ivt, slab ikt, ring slot islot)
    int cslot = 0, ivt = blockIdx.x, ikt = 0, islot = 0, inflight = 0;
    auto issue_next = [&]() {
        if constexpr (DMA) {
            if (ivt >= nwg) return;
            if (ikt == 0) {
                const int bid = xcd_remap(ivt, nwg);
                const int tile_m = bid / tiles_n;
                dma_rows(tile_m * CF::BM, (bid - tile_m * tiles_n) * BN);
            }
            dma_issue(ikt, islot);
            islot = islot + 1 == CF::NS ? 0 : islot + 1;
            if (++ikt == nk) {
                ikt = 0;
                ivt += gridDim.x;
            }
            ++inflight;
        }
    };
    if constexpr (DMA) {
#pragma unroll
        for (int j = 0; j < CF::NS - 1; ++j) issue_next();
    } else {
        g_load(0);
    }
    for (; vt < nwg; ) {
    const int cm0 = m0, cn0 = n0;   // the tile being computed (m0/n0 move on to the prefetched tile below)
    const int n = cn0 + c8 * 8;
    const bool n_ok = t_active && n < d.N;
    const bool full = n + 8 <= d.N;
    const bool use_pre = kPre && Pb != nullptr && n_ok && full;
    f32x4_t acc[TNW][4];  // [tn][tm]
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if constexpr (!DMA) {
        s_store();
        __syncthreads();
    }

    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (DMA) {
            // this wave's pieces of the slab about to be consumed have landed; with three slots the NEXT slab's
            // 6 loads (4 A + 2 B pieces per wave) may stay in flight across the barrier: a counted wait.  Younger
            // stores / loads of the epilogue only make the wait stricter, never weaker (loads return in order).
            if (CF::NS == 3 && inflight >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // ... everyone's; the slot consumed last is free
            issue_next();
            --inflight;
            As = smem + RING0 + cslot * CF::SMEM_AB;
            Bs = As + CF::A_BYTES;
            cslot = cslot + 1 == CF::NS ? 0 : cslot + 1;
        } else {
            if (kt + 1 < nk) g_load(kt + 1);
        }
        if constexpr (kPre) {
            if (kt == nk - 1 && use_pre) {
#pragma unroll
                for (int hf = 0; hf < NWM; ++hf)
#pragma unroll
                    for (int it = 0; it < NR; ++it) {
                        const int row = rg + it * CF::RG;
                        const long m = (long)cm0 + hf * 64 + row;
                        if (row < 64 && m < d.M) pre[hf][it] = load16_nt(Pb + m * ldp + n);   // read once: do not displace the operand panels in L2
                    }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + (lane >> 4);
            uint4 af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ra_ = wm * 64 + t * 16 + (lane & 15);
                af[t] = *reinterpret_cast<const uint4*>(As + ra_ * kRowBytes + ((chunk ^ (ra_ & 7)) << 4));
            }
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn) {
                const int rb_ = wn * (16 * TNW) + tn * 16 + (lane & 15);
                const uint4 bf = *reinterpret_cast<const uint4*>(Bs + rb_ * kRowBytes + ((chunk ^ (rb_ & 7)) << 4));
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            *reinterpret_cast<const bf16x8_t*>(&bf), *reinterpret_cast<const bf16x8_t*>(&af[tm]),
                            acc[tn][tm], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int tm = 0; tm < 4; ++tm)
                            acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                reinterpret_cast<const float*>(&bf)[j], reinterpret_cast<const float*>(&af[tm])[j],
                                acc[tn][tm], 0, 0, 0);
                }
            }
        }
        if constexpr (!DMA) {
            __syncthreads();  // every wave is done reading this slab
            if (kt + 1 < nk) {
                s_store();
                __syncthreads();
            }
        }
    }
    // start fetching the next tile's first slab; it lands while this tile's epilogue runs
    vt += gridDim.x;
    if (vt < nwg) {
        set_tile(vt);
        if constexpr (!DMA) g_load(0);
    }

    // ---- fused epilogue, two 64-row halves staged through LDS as fp32 [64][BN+4]
    float bias[8], csum[8], csq[8];
    {
        // unconditional loads from a clamped index + selects: as `cond ? d.bias[...] : 0` this was one exec-masked branch with a
        // single dword load per element, eight L2 round trips in series at the top of EVERY tile's epilogue
        const float* bsrc = d.bias ? d.bias + (long)z * d.strideBias : reinterpret_cast<const float*>(Bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = d.bias && n_ok && n + j < d.N;
            const float bl = bsrc[live ? n + j : 0];
            bias[j] = live ? bl : 0.f;
            csum[j] = csq[j] = 0.f;
        }
    }
    long c_off = 0;
    if (F_UNPATCH && n_ok) {
        const int tap = n / d.c_C, ch = n - tap * d.c_C;
        c_off = ((long)(tap >> 1) * d.c_W + (tap & 1)) * d.c_C + ch;
    }
    // staged piece: 64 padded rows, or (3-slot LDS-DMA form) 32 unpadded rows whose 16-byte units are XOR-swizzled with
    // the row (conflict-free fragment writes, 2-way on the row reads)
    constexpr bool SW = DMA && CF::SWZ;
    constexpr int PRr = SW ? CF::PR : 64, NSUB = 64 / PRr, TMP = PRr / 16, NRP = SW ? PRr / CF::RG : NR;
#pragma unroll
    for (int half = 0; half < NWM; ++half) {
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        if (half || sub) __syncthreads();  // previous piece fully read
        if (wm == half) {
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn)
#pragma unroll
                for (int tml = 0; tml < TMP; ++tml) {
                    const int row = tml * 16 + (lane & 15), col = wn * (16 * TNW) + tn * 16 + (lane >> 4) * 4;
                    float* dst = SW ? Cs + row * BN + ((((col >> 2) ^ (row & 15))) << 2) : Cs + row * CF::LDCS + col;
                    *reinterpret_cast<f32x4_t*>(dst) = acc[tn][sub * TMP + tml];
                }
        }
        __syncthreads();
        if (n_ok) {
#pragma unroll
            for (int itp = 0; itp < NRP; ++itp) {
                const int it = sub * NRP + itp;       // index of this row among the thread's rows of the 64-row half
                const int row = rg + itp * CF::RG;
                if (row >= PRr) break;
                const long m = (long)cm0 + half * 64 + sub * PRr + row;
                if (m >= d.M) break;
                float v[8];
                {
                    const float* pa = SW ? Cs + row * BN + (((2 * c8) ^ (row & 15)) << 2) : Cs + row * CF::LDCS + c8 * 8;
                    const float* pb = SW ? Cs + row * BN + (((2 * c8 + 1) ^ (row & 15)) << 2) : Cs + row * CF::LDCS + c8 * 8 + 4;
                    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(pa);
                    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(pb);
                    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
                    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (EPI < 0 ? v[j] * d.alpha : v[j]) + bias[j];
                if (F_C2) {   // second output: pre-activation (mode 1) or GELU'(pre-activation) (mode 2)
                    float w[8];
                    if (EPI >= 0 || (d.c2_mode == 2 && d.act == GA_ACT_GELU)) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {   // v := gelu, w := gelu'
                            if constexpr (sizeof(T) == 2 && EPI == EPI_FC1) gelu_both_fast(v[j], v[j], w[j]);
                            else gelu_both_f(v[j], v[j], w[j]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) w[j] = d.c2_mode == 2 ? gelu_grad_f(v[j]) : v[j];
                        if (d.act == GA_ACT_GELU) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
                        }
                    }
                    if (F_RELU) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    T* C2p = reinterpret_cast<T*>(d.C2) + z * d.strideC + m * d.ldc + n;
                    if (full) {
                        store8_nt(C2p, w);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (n + j < d.N) elt<T>::st(C2p + j, w[j]);
                    }
                } else if (F_GELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if constexpr (sizeof(T) == 2 && EPI == EPI_FC1) {
                            float unused;
                            gelu_both_fast(v[j], v[j], unused);
                        } else {
                            v[j] = gelu_f(v[j]);
                        }
                    }
                } else if (F_RELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (F_H) {
                    float h[8];
                    if (use_pre) {
                        if constexpr (kPre) unpack8(pre_sel(half, it), h);
                    } else if (full) {
                        load8_nt(Hb + m * d.ldh + n, h);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) h[j] = n + j < d.N ? elt<T>::ld(Hb + m * d.ldh + n + j) : 0.f;
                    }
                    if (F_HDERIV) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= h[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(h[j]);
                    }
                }
                if (F_RS) {
                    const float s = d.rowscale[(unsigned)m / (unsigned)d.rows_per_scale];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= s;
                }
                if (F_R) {
                    float r[8];
                    if (use_pre && !F_H) {
                        if constexpr (kPre) unpack8(pre_sel(half, it), r);
                    } else if (full) {
                        load8_nt(Rb + m * d.ldr + n, r);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) r[j] = n + j < d.N ? elt<T>::ld(Rb + m * d.ldr + n + j) : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += r[j];
                }
                if (F_RELUA) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (F_CSUM) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        csum[j] += v[j];
                        csq[j] += v[j] * v[j];
                    }
                }
                long off;
                if (F_UNPATCH) {
                    const unsigned OW = d.c_W >> 1, OH = d.c_H >> 1;
                    const unsigned ox = (unsigned)m % OW;
                    const unsigned t = (unsigned)m / OW;
                    const unsigned oy = t % OH;
                    const long b = t / OH;
                    off = ((b * d.c_H + 2 * oy) * d.c_W + 2 * ox) * (long)d.c_C + c_off;
                } else {
                    off = z * d.strideC + m * d.ldc + n;
                }
                if (F_CF32) {
                    float* Cp = reinterpret_cast<float*>(d.C) + off;
                    if (full) {
                        store8(Cp, v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (n + j < d.N) Cp[j] = v[j];
                    }
                } else {
                    T* Cp = reinterpret_cast<T*>(d.C) + off;
                    if (full) {
                        store8_nt(Cp, v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (n + j < d.N) elt<T>::st(Cp + j, v[j]);
                    }
                }
            }
        }
    }
    }
    if (F_CSUM) {  // workgroup-level column reduction over the RG row groups, then one atomic per column
        float* red = reinterpret_cast<float*>(smem);  // [2][RG][BN]; the 16 KiB staging area takes one half at a time
        constexpr int NPASS = SW ? 2 : 1;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (SW && pass == 1 && !d.colsumsq) break;
            __syncthreads();
            if (t_active) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (SW) {
                        red[rg * BN + c8 * 8 + j] = pass == 0 ? csum[j] : csq[j];
                    } else {
                        red[rg * BN + c8 * 8 + j] = csum[j];
                        red[CF::RG * BN + rg * BN + c8 * 8 + j] = csq[j];
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < (SW ? BN : 2 * BN); i += CF::NTHR) {
                const int which = SW ? pass : i / BN, col = SW ? i : i - which * BN;
                if (cn0 + col < d.N && (which == 0 || d.colsumsq)) {
                    float s = 0.f;
                    for (int r = 0; r < CF::RG; ++r) s += red[(SW ? 0 : which * CF::RG * BN) + r * BN + col];
                    atomicAdd((which ? d.colsumsq : d.colsum) + z * d.strideCol + cn0 + col, s);
                }
            }
        }
    }
    __syncthreads();  // LDS (C staging / reduction scratch) is free for the next tile's slab
    }  // persistent tile loop
#undef F_GELU
#undef F_RELU
#undef F_C2
#undef F_H
#undef F_HDERIV
#undef F_RS
#undef F_R
#undef F_RELUA
#undef F_CSUM
#undef F_UNPATCH
#undef F_CF32
#undef F_AACT
}

// ================================================================================================
// NT kernel, 8-wave ping-pong form (bf16, plain operands, one of the compile-time epilogues).
// 256 x 256 output tile, 512 threads = 2 (m) x 4 (n) waves of 128 x 64 (acc: 8 x 4 MFMA tiles = 128 registers), K tiles of
// 64 in a ring of two 64 KiB buffers, each four 16 KiB half-tiles {A rows 0-127, A rows 128-255, B rows 0-127, B rows
// 128-255} in the [row][128 B] XOR layout of the other NT forms, filled by LDS-DMA (every wave brings 16 rows of each).
// A K tile is multiplied in four PHASES of 16 MFMAs (one 64 x 32 quadrant of the wave's tile over K = 64); each phase is
//     L: ds_read the fragments the phase needs | issue LDS-DMA | s_waitcnt | barrier      M: 16 MFMA | barrier
// and the two wave groups (m halves; one wave of each group per SIMD) run ONE barrier apart, so that in every barrier
// interval one group multiplies while the other reads / issues DMA.  Fragment reads: A(rows 0-63) 8 in L of phase 1,
// A(rows 64-127) 8 in L of phase 3; the 4 B reads of phase 2 and the 4 of the NEXT K tile's phase 1 are issued at the head of
// the M steps of phases 1 / 3, under MFMAs that do not use those registers.  All reads of a phase are retired BEFORE its first barrier, so a half-tile
// may be refilled from the phase after its last read.  Issue schedule (an LDS-DMA piece costs 60-180 clocks of issue, so they
// are spread, one half-tile = 2 pieces per wave and phase): phases 3, 4 of tile g: the B halves of tile g+2; phases 1, 2 of
// tile g+1: the A halves of g+2 -- 2 to 5 phases ahead of their first use; counted waits (vmcnt(6) / vmcnt(4)) in phases 3 / 4
// retire the B / A halves of tile g+1 one phase before they are read.  The K-tile stream runs across the workgroup's output tiles
// (persistent), so the first two K tiles of the next output tile land while this tile's epilogue runs.
// Epilogue: per wave, 16 rows at a time through a private 4 KiB fp32 staging piece -> 8 columns per lane, 16-byte stores.
// ================================================================================================
constexpr int kPPThreads = 512, kPPHalf = 16384, kPPBuf = 4 * kPPHalf, kPPSmem = 2 * kPPBuf + 8 * 4096;

template <int EPI>
__global__ __launch_bounds__(kPPThreads) void gemm_nt_pp_kernel(const ga_gemm_desc d, const int dbg_arg) {
#ifdef GAEXT_DEBUG
    const int dbg = dbg_arg;            // timing experiments (results deliberately wrong): -DGAEXT_DEBUG builds only
#else
    constexpr int dbg = 0;              // release build: every `dbg` branch below is dead code and is not emitted
    (void)dbg_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (d.N + 255) / 256, tiles_m = (d.M + 255) / 256;
    const int nwg = tiles_n * tiles_m;
    const int z = blockIdx.z;
    const int za = d.a_batch_mod > 0 ? z % d.a_batch_mod : z;
    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(d.A) + (long)za * d.strideA;
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(d.B) + (long)z * d.strideB;
    const int nk = (d.K + 63) / 64;
    const int my_tiles = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * nk;                               // K tiles in this workgroup's stream

    // ---- fragment addresses (bytes from the buffer start): lane reads row (lane & 15) of a 16-row fragment, 16-byte chunk
    // ks*4 + (lane >> 4); the chunk sits at chunk ^ (row & 7), and ks = 1 is the same address with bit 6 flipped
    const int fa = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const unsigned char* aF[2] = {smem + wr * kPPHalf + fa, smem + wr * kPPHalf + (fa ^ 64)};
    const unsigned char* bF[2] = {smem + (2 + (wc >> 1)) * kPPHalf + (wc & 1) * 8192 + fa,
                                  smem + (2 + (wc >> 1)) * kPPHalf + (wc & 1) * 8192 + (fa ^ 64)};

    // ---- LDS-DMA roles: of every half-tile this wave brings pieces 2w and 2w+1 (a piece = 8 rows x 128 B = one glds16).
    // Rows beyond M / N are clamped to the last row (their products are never stored); the K tail fetches the zero page.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int dchunk = ((lane & 7) ^ (lane >> 3)) * 8;
    // two issue-side cursors: the A rows 128-255 half of a K tile is issued one phase-group later than its other three halves
    struct Cursor { int vt, kt; };
    Cursor cm = {(int)blockIdx.x, 0}, cl = {(int)blockIdx.x, 0};
    int off_a0[2], off_b[2][2], off_a1[2];                     // element offsets of this lane's rows in the cursor's output tile
    auto rows_of = [&](int t, int h, bool is_a, int (&o)[2]) {
        const int bid = xcd_remap(t, nwg);
        const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = h * 128 + 16 * wave + 8 * i + (lane >> 3);
            o[i] = is_a ? min(tm * 256 + r, d.M - 1) * (int)d.lda : min(tn * 256 + r, d.N - 1) * (int)d.ldb;
        }
    };
    auto set_main = [&]() {
        rows_of(cm.vt, 0, false, off_b[0]);
        rows_of(cm.vt, 1, false, off_b[1]);
    };
    auto set_lag = [&]() {
        rows_of(cl.vt, 0, true, off_a0);
        rows_of(cl.vt, 1, true, off_a1);
    };
    const u32x4_s rsA = make_rsrc(Ab, 0x80000000u), rsB = make_rsrc(Bb, 0x80000000u);
    const int ktail = d.K & 63;                                 // > 0: the last K tile is ragged; its chunks beyond K read as zeros
    auto glds_half = [&](const bf16_t* base, const int (&o)[2], int kt_, int half, int buf) {
        if (dbg & 2) return;                                    // timing experiment: no operand traffic
        const unsigned dst = lds0 + buf * kPPBuf + half * kPPHalf + 2 * wave * 1024;
        if (dbg & 16) {                                         // the 64-bit-address form (A/B experiments: 7-11 % slower)
            const int k = kt_ * 64 + dchunk;
            const bool kv = k < d.K && !(dbg & 8);              // (dbg 8: every piece comes from the zero page -- no operand traffic, same issue)
            glds16x2(kv ? static_cast<const void*>(base + o[0] + k) : static_cast<const void*>(g_zero_page),
                     kv ? static_cast<const void*>(base + o[1] + k) : static_cast<const void*>(g_zero_page), dst);
            return;
        }
        unsigned v0 = (unsigned)(o[0] + dchunk) * 2u, v1 = (unsigned)(o[1] + dchunk) * 2u;
        if (ktail && kt_ == nk - 1 && dchunk >= ktail) v0 = v1 = 0x80000400u;   // out of the resource's range: zeros
        blds16x2(base == Ab ? rsA : rsB, v0, v1, (unsigned)kt_ * 128u, dst);
    };
    auto issue_B0 = [&](int buf) { glds_half(Bb, off_b[0], cm.kt, 2, buf); };
    auto issue_B1 = [&](int buf) {
        glds_half(Bb, off_b[1], cm.kt, 3, buf);
        if (++cm.kt == nk) {
            cm.kt = 0;
            cm.vt += gridDim.x;
            if (cm.vt < nwg) set_main();
        }
    };
    auto issue_A0 = [&](int buf) { glds_half(Ab, off_a0, cl.kt, 0, buf); };
    auto issue_A1 = [&](int buf) {
        glds_half(Ab, off_a1, cl.kt, 1, buf);
        if (++cl.kt == nk) {
            cl.kt = 0;
            cl.vt += gridDim.x;
            if (cl.vt < nwg) set_lag();
        }
    };

    f32x4_t acc[4][8];                                         // [tn][tm]: C[m = tm*16 + (lane&15)][n = tn*16 + (lane>>4)*4 + r]
    bf16x8_t af[4][2], bc0[2][2], b1[2][2], bn0[2][2];
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = zero4;

#define PP_READ_A(mh, bufp)                                                                                         \
    _Pragma("unroll") for (int tm = 0; tm < 4; ++tm) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)               \
        af[tm][ks] = *reinterpret_cast<const bf16x8_t*>(aF[ks] + (bufp) + ((mh) * 4 + tm) * 2048);
#define PP_READ_B(dst, nh, bufp)                                                                                    \
    _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)               \
        dst[tn][ks] = *reinterpret_cast<const bf16x8_t*>(bF[ks] + (bufp) + ((nh) * 2 + tn) * 2048);
#define PP_MMA(bq, mh, nh)                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)               \
        _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                                            \
            acc[(nh) * 2 + tn][(mh) * 4 + tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[tn][ks], af[tm][ks],     \
                                                                                      acc[(nh) * 2 + tn][(mh) * 4 + tm], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
// the same, 4 MFMAs per A fragment pair, which is then re-read (rows of quadrant row `mh2`) while the remaining MFMAs run
#define PP_MMA_RELOAD(bq, mh, nh, mh2, bufp)                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    _Pragma("unroll") for (int tm = 0; tm < 4; ++tm) {                                                              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)           \
            acc[(nh) * 2 + tn][(mh) * 4 + tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[tn][ks], af[tm][ks],     \
                                                                                      acc[(nh) * 2 + tn][(mh) * 4 + tm], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
            af[tm][ks] = *reinterpret_cast<const bf16x8_t*>(aF[ks] + (bufp) + ((mh2) * 4 + tm) * 2048);             \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }                                                                                                               \
    __builtin_amdgcn_s_setprio(0);
#define PP_LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_BAR                            \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0)

    // ---- prologue: K tiles 0 and 1 of the stream
    set_main();
    set_lag();
    issue_B0(0);
    issue_B1(0);
    issue_A0(0);
    issue_A1(0);
    if (S > 1) {
        issue_B0(1);
        issue_B1(1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_BAR;
    PP_READ_B(bn0, 0, 0);
    PP_LGKM0;

    const bf16_t* Hb = d.H ? reinterpret_cast<const bf16_t*>(d.H) + (long)z * d.strideH : nullptr;
    const bf16_t* Rb = d.R ? reinterpret_cast<const bf16_t*>(d.R) + (long)z * d.strideR : nullptr;
    float* stage = reinterpret_cast<float*>(smem + 2 * kPPBuf + wave * 4096);
    int vt = blockIdx.x, kt = 0;
#pragma unroll 1
    for (int g = 0; g < S; ++g) {
        const int bufp = (g & 1) * kPPBuf, nbufp = kPPBuf - bufp;
        const bool more1 = g + 1 < S, more2 = g + 2 < S;
        // the second wave group runs one barrier behind the first through the K loop of an output tile; the groups re-align for
        // the epilogue (run side by side, it is paid once, not once per group)
        if (kt == 0 && wr == 1) { PP_BAR; }
        // ---- phase 1: quadrant (rows 0-63, cols 0-31).  L: A rows 0-63 + DMA;  M: the B fragments of phase 2 ride under the MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) bc0[i][j] = bn0[i][j];
        PP_READ_A(0, bufp);
        if (more1) issue_A0((g + 1) & 1);                       // A halves of K tile g+1 (free since phase 4 of tile g-1)
        PP_LGKM0;
        PP_BAR;
        PP_READ_B(b1, 1, bufp);
        PP_MMA(bc0, 0, 0);
        PP_BAR;
        // ---- phase 2: (rows 0-63, cols 32-63).  L: DMA only
        if (more1) {
            issue_A1((g + 1) & 1);
            if (!(dbg & 4)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // B halves of K tile g+1 have landed (this wave's share)
        }
        PP_LGKM0;
        PP_BAR;
        PP_MMA_RELOAD(b1, 0, 1, 1, bufp);                       // ... and A rows 64-127 replace rows 0-63 as their registers die
        PP_BAR;
        // ---- phase 3: (rows 64-127, cols 32-63); the B halves of this buffer are free: K tile g+2
        if (more2) issue_B0(g & 1);
        PP_LGKM0;
        PP_BAR;
        if (more1) { PP_READ_B(bn0, 0, nbufp); }               // first B fragments of K tile g+1, under the MFMAs
        PP_MMA(b1, 1, 1);
        PP_BAR;
        // ---- phase 4: (rows 64-127, cols 0-31).  L: DMA only
        if (more2) issue_B1(g & 1);
        if (more1 && !(dbg & 4)) {
            if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // A halves of K tile g+1
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_LGKM0;
        PP_BAR;
        PP_MMA(bc0, 1, 0);
        PP_BAR;
        if (++kt < nk) continue;
        kt = 0;
        if (wr == 0) { PP_BAR; }
        if (dbg & 1) {                                          // timing experiment: no epilogue
            vt += gridDim.x;
            continue;
        }
        // ================= epilogue of output tile vt (no workgroup barrier in here: the staging piece is private) =================
        {
            const int bid = xcd_remap(vt, nwg);
            const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
            const int q = lane & 7, rr = lane >> 3;
            const int n = tile_n * 256 + wc * 64 + q * 8;
            const bool n_ok = n < d.N;                          // N % 8 == 0: a piece is wholly in or out
            float bias[8], csum[8], csq[8];
            {
                // (unconditional loads from a clamped index + selects: see gemm_nt_kernel's epilogue)
                const bool live = d.bias && n_ok;
                const float* bsrc = d.bias ? d.bias + (long)z * d.strideBias : reinterpret_cast<const float*>(Bb);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float bl = bsrc[live ? n + j : 0];
                    bias[j] = live ? bl : 0.f;
                    csum[j] = csq[j] = 0.f;
                }
            }
            const bool do_csum = (EPI == EPI_PLAIN || EPI == EPI_DG2) && d.colsum != nullptr;
            // the epilogue operand (dgrad2: stored GELU'; fc2: shortcut) of row piece tm+1 is requested while piece tm is staged
            constexpr bool kOp = EPI == EPI_DG2 || EPI == EPI_FC2;
            const bf16_t* Pb = EPI == EPI_DG2 ? Hb : Rb;
            const long ldp = EPI == EPI_DG2 ? d.ldh : d.ldr;
            const long mrow0 = (long)tile_m * 256 + wr * 128 + rr;
            uint4 opn[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
            auto fetch_op = [&](int tm) {
                if constexpr (kOp) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const long m = mrow0 + tm * 16 + 8 * p;
                        if (n_ok && m < d.M) opn[p] = load16_nt(Pb + m * ldp + n);
                    }
                }
            };
            fetch_op(0);
#pragma unroll
            for (int tm = 0; tm < 8; ++tm) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the previous piece has been read
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const int row = lane & 15, u = tn * 4 + (lane >> 4);
                    *reinterpret_cast<f32x4_t*>(stage + row * 64 + ((u ^ row) << 2)) = acc[tn][tm];
                    acc[tn][tm] = zero4;
                }
                const uint4 opc[2] = {opn[0], opn[1]};
                if (tm + 1 < 8) fetch_op(tm + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int row = 8 * p + rr;
                    const long m = (long)tile_m * 256 + wr * 128 + tm * 16 + row;
                    const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(stage + row * 64 + (((2 * q) ^ row) << 2));
                    const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(stage + row * 64 + (((2 * q + 1) ^ row) << 2));
                    if (!n_ok || m >= d.M) continue;
                    float v[8] = {lo[0] + bias[0], lo[1] + bias[1], lo[2] + bias[2], lo[3] + bias[3],
                                  hi[0] + bias[4], hi[1] + bias[5], hi[2] + bias[6], hi[3] + bias[7]};
                    if constexpr (EPI == EPI_FC1) {
                        float w[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) gelu_both_fast(v[j], v[j], w[j]);
                        if (d.C2) store8_nt(reinterpret_cast<bf16_t*>(d.C2) + (long)z * d.strideC + m * d.ldc + n, w);
                    }
                    if constexpr (EPI == EPI_DG2) {
                        float h[8];
                        unpack8(opc[p], h);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= h[j];
                    }
                    if constexpr (EPI == EPI_FC2) {
                        if (d.rowscale) {
                            const float sc = d.rowscale[(unsigned)m / (unsigned)d.rows_per_scale];
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] *= sc;
                        }
                        float r[8];
                        unpack8(opc[p], r);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += r[j];
                    }
                    if (do_csum) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            csum[j] += v[j];
                            csq[j] += v[j] * v[j];
                        }
                    }
                    store8_nt(reinterpret_cast<bf16_t*>(d.C) + (long)z * d.strideC + m * d.ldc + n, v);
                }
            }
            if (do_csum) {                                      // lanes q, q+8, .. hold the same 8 columns over different rows
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = csum[j], b = csq[j];
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) {
                        a += __shfl_xor(a, o, 64);
                        b += __shfl_xor(b, o, 64);
                    }
                    if (rr == 0 && n_ok) {
                        atomicAdd(d.colsum + (long)z * d.strideCol + n + j, a);
                        if (d.colsumsq) atomicAdd(d.colsumsq + (long)z * d.strideCol + n + j, b);
                    }
                }
            }
        }
        vt += gridDim.x;
    }
#undef PP_READ_A
#undef PP_READ_B
#undef PP_MMA
#undef PP_MMA_RELOAD
#undef PP_LGKM0
#undef PP_BAR
}

// ================================================================================================
// NT kernel, 3-slot ring form ("r3", round 3; bf16, plain operands, one of the compile-time epilogues).
//
// Why another body: on the K = 192 .. 768 launches of the ConvNeXt / CSWin trunks (M = 12,544 .. 200,704 token rows) the
// 8-wave forms run ONE workgroup per CU, so a tile's epilogue (bias / GELU + GELU' / x stored GELU' / + shortcut, 128 values
// per lane) is paid un-overlapped after a K loop of only 3 .. 12 slabs, and the 4-wave 128 x 128 LDS-DMA form reads 0.5
// fragments per MFMA with one slab in flight.  This form is built for TWO INDEPENDENT workgroups per CU:
//   * 256 x 128 output tile per 256-thread workgroup, 4 waves as 2 (m) x 2 (n), 128 x 64 per wave (32 accumulator tiles =
//     128 VGPRs; 12 fragment reads per 32 MFMAs), exactly 80 KiB of LDS -> two workgroups per CU whose K loops and epilogues
//     drift apart: one's epilogue (VALU, stores) runs under the other's MFMAs on the same SIMDs;
//   * K stages of 32 (64-byte rows): a stage = A 256 rows + B 128 rows = 24 KiB, ring of THREE stages, filled by LDS-DMA
//     (buffer_load ... lds, 6 pieces of 1 KiB per wave and stage); two stages are in flight while one is multiplied, ONE
//     barrier per stage, counted s_waitcnt vmcnt(6).  The stage stream runs across the workgroup's output tiles, so the
//     first two stages of the next tile land during the epilogue.  Every stage top issues exactly 6 pieces -- beyond the end
//     of the stream they carry out-of-range offsets (zeros into a free slot) -- so the counted wait is exact everywhere;
//   * 64-byte rows: lane l of a fragment read takes row l & 15, 16-byte chunk l >> 4; chunk c of row r sits at
//     c ^ f(r >> 2), f = (0, 3, 2, 1): conflict-free for the four 16-lane groups ds_read_b128 is served in.  The DMA writes
//     LDS lane-linearly, so the permutation is applied to the SOURCE chunk each lane fetches;
//   * epilogue per wave, no workgroup barrier: 8 rows x 64 columns at a time through a private 2 KiB fp32 staging piece
//     (XOR-swizzled) -> 8 consecutive columns per lane, 16-byte stores of whole 128-byte row segments.  The epilogue operands
//     (bias; stored GELU' of dgrad2 / shortcut of fc2, 16 x 16 B per lane) are requested by asm buffer loads at the top of the
//     tile's LAST stage and right after its MFMAs; one s_waitcnt vmcnt(0) before the first staging piece retires them
//     together with the two stages in flight, which is why the first two stages of a tile need no wait of their own and the
//     epilogue's stores are never waited for before the third stage.
// ================================================================================================
constexpr int kR3Threads = 256, kR3Slot = (256 + 128) * 64, kR3Stage = 2048, kR3Smem = 3 * kR3Slot + 4 * kR3Stage;   // 81,920 B
static_assert(kR3Smem == 81920, "two workgroups per CU need exactly 80 KiB each");

__device__ __forceinline__ u32x4_t r3_bload16(u32x4_s rs, unsigned voff) {        // read-once operand: streaming
    u32x4_t v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}
__device__ __forceinline__ u32x4_t r3_bload16_c(u32x4_s rs, unsigned voff) {      // re-read by other tiles (bias): cached
    u32x4_t v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}

__device__ __forceinline__ float r3_bload4(u32x4_s rs, unsigned voff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}
__device__ __forceinline__ void r3_bstore16(u32x4_s rs, unsigned voff, u32x4_t v) {    // offsets >= num_records are dropped
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ u32x4_t r3_pack8(const float v[8]) {
    u32x4_t a;
    a[0] = pack2bf(v[0], v[1]); a[1] = pack2bf(v[2], v[3]); a[2] = pack2bf(v[4], v[5]); a[3] = pack2bf(v[6], v[7]);
    return a;
}

// arrival tickets per CU: the two co-resident workgroups of a CU draw consecutive tickets, so ticket parity tells them apart
// (placement is never relied on for correctness: the parity only decides which of the two starts half a tile late)
__device__ int g_r3_ticket[4096];

// DBG != 0: timing experiments (RESULTS DELIBERATELY WRONG), instantiated by -DGAEXT_DEBUG builds only: 1 no epilogue, 2 no global
// stores, 4 no GELU, 8 no operand loads, 16 no DMA, 32 wait-free K loop.  Compile-time on purpose: a run-time switch kept every
// variant's registers live at once, hipcc spilled the destination registers of in-flight asm loads (guide 5.7 item 1) and a
// garbage store offset inside the then 2 GiB buffer window faulted.
// AK = 1: the instantiation that also fetches the gather kinds GA_A_NEIGH2 / GA_A_CONV3S2 (their edge bookkeeping costs the plain
// instantiation 25 spilled SGPRs when it is compiled in)
template <int EPI, int DBG = 0, int AK = 0>
__global__ __launch_bounds__(kR3Threads, 2) void gemm_nt_r3_kernel(const ga_gemm_desc d, const int stagger) {
    constexpr int dbg = DBG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (d.N + 127) / 128, tiles_m = (d.M + 255) / 256;
    const int nwg = tiles_n * tiles_m;
    const int z = blockIdx.z;
    const int za = d.a_batch_mod > 0 ? z % d.a_batch_mod : z;
    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(d.A) + (long)za * d.strideA;
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(d.B) + (long)z * d.strideB;
    const int nk = (d.K + 31) / 32;
    const int my_tiles = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * nk;                                   // stages in this workgroup's stream

    // ---- fragment read addresses: row (lane & 15) of a 16-row piece (1 KiB), chunk (lane >> 4) ^ f(row >> 2)
    const int frow = lane & 15;
    const unsigned fa = frow * 64 + ((((lane >> 4) ^ ((4 - (frow >> 2)) & 3))) << 4);
    const unsigned char* aF = smem + wm * 8192 + fa;               // + tm * 1024 (+ slot)
    const unsigned char* bF = smem + 16384 + wn * 4096 + fa;       // + tn * 1024 (+ slot)

    // ---- LDS-DMA roles: wave w brings A pieces 4w .. 4w+3 and B pieces 2w, 2w+1 of every stage (a piece = 16 rows x 64 B).
    // Lane l writes row l >> 2, LDS chunk l & 3 of its piece, which must hold SOURCE chunk (l & 3) ^ f((l >> 2) >> 2)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int dchunk = ((lane & 3) ^ ((4 - (lane >> 4)) & 3)) * 8;  // element offset of this lane's chunk inside the 32-wide stage
    constexpr unsigned kOob = 0x80000000u;                         // >= num_records: the piece reads as zeros
    // buffer windows = the exact extent of each matrix: an offset beyond it (kOob, or anything a bug produces) reads as zeros or
    // drops the store instead of touching memory that is not the operand's
    auto extent = [](long rows, long ld, long cols, int esz) { return (unsigned)(((rows - 1) * ld + cols) * esz); };
    // A-operand kinds: plain rows, or the 2 x 2 / stride-2 patches of an NHWC map (GA_A_PATCH2, the downsample convs): a patch row
    // is TWO runs of 2C contiguous elements (taps (0,0)(0,1) and (1,0)(1,1)), so the lane offsets point at the patch origin and
    // the second half of the K stages adds one image row (W * C elements) through the scalar offset -- no gather code in the loop
    // GA_A_NEIGH2 (data gradient of the 3 x 3 / stride-2 convs, plain epilogue only): row m = pixel (y, x) of the a_H x a_W map, its
    // K = 4C elements the pixels (y, x) (y, x+1) (y+1, x) (y+1, x+1) -- the same two runs of 2C elements one image row apart, but
    // a neighbour beyond the right / lower edge is ZERO: two bits per staged row (has a right / a lower neighbour) against the
    // quarter of K the stage lies in turn the lane's offset out of range
    const bool patch2 = d.a_kind == GA_A_PATCH2;
    const bool neigh2 = AK == 1 && EPI == EPI_PLAIN && d.a_kind == GA_A_NEIGH2;
    // GA_A_CONV3S2 (the 3 x 3 / stride-2 convs themselves, even maps, plain epilogue only): row m = output pixel, its K = 9C elements
    // three runs of 3C (taps kx = 0, 1, 2 of one tap row are neighbouring pixels) one map row apart, starting one pixel up and left
    // of the centre (2 oy, 2 ox).  The lane offset is the CENTRE pixel (always inside the map: the range check sees the lane offset
    // only), the buffer base is moved W + 1 pixels down, and the scalar offset of a stage counts from the upper-left tap:
    // linear k + ky (W - 3) C.  Taps left of column 0 / above row 0 turn the lane offset out of range (two bits per staged row).
    const bool conv3s2 = AK == 1 && EPI == EPI_PLAIN && d.a_kind == GA_A_CONV3S2;
    const unsigned p2_row = (patch2 || neigh2) ? (unsigned)(d.a_W * d.a_C - 2 * d.a_C) * 2u : 0u;   // bytes added from stage nk/2 on
    const unsigned c3_row = conv3s2 ? (unsigned)((d.a_W - 3) * d.a_C) * 2u : 0u;                    // bytes added per tap row
    const unsigned c3_shift = conv3s2 ? (unsigned)((d.a_W + 1) * d.a_C) * 2u : 0u;
    const unsigned a_bytes = (patch2 || conv3s2) ? extent(4L * d.M, d.a_C, d.a_C, 2) : neigh2 ? extent(d.M, d.a_C, d.a_C, 2) : extent(d.M, d.lda, d.K, 2);
    const u32x4_s rsA = make_rsrc(reinterpret_cast<const unsigned char*>(Ab) - c3_shift, a_bytes + c3_shift),
                  rsB = make_rsrc(Bb, extent(d.N, d.ldb, d.K, 2));
    unsigned n2_have = 0;      // NEIGH2: bits 2i, 2i+1 = row of piece i has a right / lower neighbour; CONV3S2: a left / upper one
    const int ktail = d.K & 31;                                    // > 0: the last stage is ragged
    unsigned aoff[4], boff[2];
    int ivt = blockIdx.x, ikt = 0;                                 // issue-side cursor: output tile, stage inside it
    auto dma_rows = [&]() __attribute__((always_inline)) {
        if (ivt < nwg) {
            const int bid = xcd_remap(ivt, nwg);
            const int tm_ = bid / tiles_n, tn_ = bid - tm_ * tiles_n;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = (long)tm_ * 256 + (wave * 4 + i) * 16 + (lane >> 2);
                if (patch2) {
                    const unsigned OW = (unsigned)d.a_W >> 1, OH = (unsigned)d.a_H >> 1;
                    const unsigned ox = (unsigned)m % OW, t = (unsigned)m / OW, oy = t % OH, b = t / OH;
                    aoff[i] = m < d.M ? (unsigned)((((long)b * d.a_H + 2 * oy) * d.a_W + 2 * ox) * d.a_C + dchunk) * 2u : kOob;
                } else if (conv3s2) {
                    const unsigned OW = (unsigned)d.a_W >> 1, OH = (unsigned)d.a_H >> 1;
                    const unsigned ox = (unsigned)m % OW, t = (unsigned)m / OW, oy = t % OH, b = t / OH;
                    aoff[i] = m < d.M ? (unsigned)((((long)b * d.a_H + 2 * oy) * d.a_W + 2 * ox) * d.a_C + dchunk) * 2u : kOob;
                    const unsigned have = (ox > 0 ? 1u : 0u) | (oy > 0 ? 2u : 0u);
                    n2_have = (n2_have & ~(3u << (2 * i))) | (have << (2 * i));
                } else if (neigh2) {
                    const unsigned x = (unsigned)m % (unsigned)d.a_W, y = ((unsigned)m / (unsigned)d.a_W) % (unsigned)d.a_H;
                    aoff[i] = m < d.M ? (unsigned)(m * d.a_C + dchunk) * 2u : kOob;
                    const unsigned have = (x + 1 < (unsigned)d.a_W ? 1u : 0u) | (y + 1 < (unsigned)d.a_H ? 2u : 0u);
                    n2_have = (n2_have & ~(3u << (2 * i))) | (have << (2 * i));
                } else {
                    aoff[i] = m < d.M ? (unsigned)(m * d.lda + dchunk) * 2u : kOob;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long n = (long)tn_ * 128 + (wave * 2 + i) * 16 + (lane >> 2);
                boff[i] = n < d.N ? (unsigned)(n * d.ldb + dchunk) * 2u : kOob;
            }
        } else {                                                   // beyond the stream: keep the op count, fetch nothing
#pragma unroll
            for (int i = 0; i < 4; ++i) aoff[i] = kOob;
            boff[0] = boff[1] = kOob;
        }
    };
    // a stage = EXACTLY 6 vector-memory operations per wave, issued as three pairs (part 0, 1, 2) that the K loop spreads between
    // its MFMA groups: a burst of 6 at the stage top cost ~600 clocks of issue against the 512 the stage's 32 MFMAs take
    // (no-epilogue timing variant: 990 TFLOP/s with the burst against 1440 without any DMA)
    auto dma_part = [&](int slot, int part) __attribute__((always_inline)) {
        const bool kdead = ktail && ikt == nk - 1 && dchunk >= ktail;
        const unsigned soff = (unsigned)ikt * 64u;
        unsigned soffA = soff + (2 * ikt >= nk ? p2_row : 0u);
        const unsigned dst = lds0 + slot * kR3Slot;
        unsigned a0 = aoff[part == 0 ? 0 : 2], a1 = aoff[part == 0 ? 1 : 3];
        if (conv3s2 && part < 2) {
            // stage -> tap (ky, kx): nk = 9 C / 32 stages, nk / 9 per tap (C % 32 == 0)
            const int third = nk / 3, spt = nk / 9;
            const int ky = (ikt >= third ? 1 : 0) + (ikt >= 2 * third ? 1 : 0);
            const int r = ikt - ky * third;
            const unsigned need = (r < spt ? 1u : 0u) | (ky == 0 ? 2u : 0u);       // kx == 0 needs a left, ky == 0 an upper neighbour
            soffA = soff + (unsigned)ky * c3_row;
            if ((need & ~(n2_have >> (4 * part))) & 3u) a0 = kOob;
            if ((need & ~(n2_have >> (4 * part + 2))) & 3u) a1 = kOob;
        }
        if (neigh2 && part < 2) {
            // quarter of K = tap (dy, dx): bit 0 needs a right neighbour, bit 1 a lower one (nk % 4 == 0: C % 32 == 0)
            const unsigned need = ((4 * ikt >= nk && 4 * ikt < 2 * nk) || 4 * ikt >= 3 * nk ? 1u : 0u) | (2 * ikt >= nk ? 2u : 0u);
            const unsigned miss = need & ~(n2_have >> (4 * part));
            if (miss & 3u) a0 = kOob;
            if ((need & ~(n2_have >> (4 * part + 2))) & 3u) a1 = kOob;
        }
        if (part == 0) blds16x2(rsA, kdead ? kOob : a0, kdead ? kOob : a1, soffA, dst + (wave * 4) * 1024);
        if (part == 1) blds16x2(rsA, kdead ? kOob : a0, kdead ? kOob : a1, soffA, dst + (wave * 4 + 2) * 1024);
        if (part == 2) {
            blds16x2(rsB, kdead ? kOob : boff[0], kdead ? kOob : boff[1], soff, dst + 16384 + (wave * 2) * 1024);
            if (++ikt == nk) {
                ikt = 0;
                ivt += gridDim.x;
                dma_rows();
            }
        }
    };
    auto dma_issue = [&](int slot) __attribute__((always_inline)) {
        dma_part(slot, 0);
        dma_part(slot, 1);
        dma_part(slot, 2);
    };

    f32x4_t acc[4][8];                                             // [tn][tm]: C[m = tm*16 + (lane&15)][n = tn*16 + (lane>>4)*4 + r]
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = zero4;

    // ---- epilogue operands
    constexpr bool kOp = EPI == EPI_DG2 || EPI == EPI_FC2;
    const bf16_t* Pb = EPI == EPI_DG2 ? (d.H ? reinterpret_cast<const bf16_t*>(d.H) + (long)z * d.strideH : nullptr)
                                      : (d.R ? reinterpret_cast<const bf16_t*>(d.R) + (long)z * d.strideR : nullptr);
    const long ldp = EPI == EPI_DG2 ? d.ldh : d.ldr;
    const u32x4_s rsP = make_rsrc(Pb, (kOp && Pb && !(dbg & 8)) ? extent(d.M, ldp, d.N, 2) : 0u);
    const u32x4_s rsBias = make_rsrc(d.bias ? d.bias + (long)z * d.strideBias : nullptr, d.bias ? (unsigned)d.N * 4u : 0u);
    u32x4_t opn[kOp ? 4 : 1][2], biasv[2];                  // operand pieces tm, tm+4 share registers (rolling prefetch)
    float rsc[EPI == EPI_FC2 ? 4 : 1][2];                          // DropPath row scales (fc2), requested with the operand pieces
    const u32x4_s rsRS = make_rsrc(EPI == EPI_FC2 ? d.rowscale : nullptr,
                                   (EPI == EPI_FC2 && d.rowscale) ? (unsigned)(((d.M - 1) / d.rows_per_scale + 1) * 4) : 0u);
    auto rs_off = [&](long m) -> unsigned { return m < d.M ? ((unsigned)m / (unsigned)d.rows_per_scale) * 4u : kOob; };
    const int q = lane & 7, rr = lane >> 3;
    float* stage = reinterpret_cast<float*>(smem + 3 * kR3Slot + wave * kR3Stage);

    int vt = blockIdx.x, kt = 0, cslot = 0;
    int tile_m = 0, tile_n = 0;
    bool first = true;
    dma_rows();
    dma_issue(0);
    dma_issue(1);
    // Two workgroups that start together on one CU run the same program in lockstep -- K loops together, epilogues together --
    // and the timing variants showed their phases simply ADD (K loop + staging + GELU + stores = the measured time).  The
    // second arrival on a CU therefore starts `stagger` x 64 clocks late (about half a tile), so that one workgroup's epilogue
    // (VALU, LDS staging, stores) runs beside the other's MFMAs for the rest of the launch.
    if (stagger > 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID [3:0]
        const unsigned key = ((xcc & 15u) << 8) | ((hw >> 8) & 255u);
        int ticket = 0;
        if (tid == 0) ticket = atomicAdd(&g_r3_ticket[key], 1);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        // only wave 0 knows the ticket; every wave sleeps on the first barrier behind it anyway
        if (wave == 0 && (ticket & 1)) {
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
        }
    }
#pragma unroll 1
    for (int g = 0; g < S; ++g) {
        if (kt == 0) {
            const int bid = xcd_remap(vt, nwg);
            tile_m = bid / tiles_n;
            tile_n = bid - tile_m * tiles_n;
        }
        // this wave's pieces of stage g have landed: the workgroup's first tile waits with the next stage's 6 pieces in flight;
        // later tiles retired stages 0 and 1 in the previous epilogue's drain and wait from stage 2 on (which also retires that
        // epilogue's stores -- two stages after they were issued)
        if ((first || kt >= 2) && !(dbg & 32)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // ... everyone's; nobody reads the slot of stage g-1 any more
        const int n = tile_n * 128 + wn * 64 + q * 8;
        const bool n_ok = n < d.N;                                 // N % 8 == 0: a piece of 8 columns is wholly in or out
        const long mrow0 = (long)tile_m * 256 + wm * 128 + rr;
        if (kt == nk - 1) {                                        // the tile's last stage: request bias and operand pieces 0-3
            biasv[0] = r3_bload16_c(rsBias, n_ok ? (unsigned)n * 4u : kOob);
            biasv[1] = r3_bload16_c(rsBias, n_ok ? (unsigned)n * 4u + 16u : kOob);
            if constexpr (kOp) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const long m = mrow0 + tm * 16 + 8 * p;
                        opn[tm][p] = r3_bload16(rsP, (n_ok && m < d.M) ? (unsigned)(m * ldp + n) * 2u : kOob);
                        if constexpr (EPI == EPI_FC2) rsc[tm][p] = r3_bload4(rsRS, rs_off(m));
                    }
            }
        }
        const int islot = cslot >= 1 ? cslot - 1 : 2;              // stage g+2 -> slot (g+2) % 3 = (g-1) % 3, free since the barrier
        {
            const unsigned sb = cslot * kR3Slot;
            bf16x8_t af[8], bfr[4];
            // fragment reads run two row tiles ahead of the MFMAs that use them (left alone, hipcc re-uses two fragment
            // registers and waits for every pair of reads right before its first MFMA)
#define R3_RA(i) af[i] = *reinterpret_cast<const bf16x8_t*>(aF + sb + (i) * 1024)
#define R3_MMA(i)                                                                                               \
    _Pragma("unroll") for (int tn = 0; tn < 4; ++tn)                                                             \
        acc[tn][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tn], af[i], acc[tn][i], 0, 0, 0)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bfr[tn] = *reinterpret_cast<const bf16x8_t*>(bF + sb + tn * 1024);
            R3_RA(0); R3_RA(1); R3_RA(2); R3_RA(3);
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 16)) dma_part(islot, 0);
            R3_MMA(0); R3_MMA(1);
            __builtin_amdgcn_sched_barrier(0);
            R3_RA(4); R3_RA(5);
            if (!(dbg & 16)) dma_part(islot, 1);
            __builtin_amdgcn_sched_barrier(0);
            R3_MMA(2); R3_MMA(3);
            __builtin_amdgcn_sched_barrier(0);
            R3_RA(6); R3_RA(7);
            if (!(dbg & 16)) dma_part(islot, 2);
            __builtin_amdgcn_sched_barrier(0);
            R3_MMA(4); R3_MMA(5); R3_MMA(6); R3_MMA(7);
#undef R3_RA
#undef R3_MMA
        }
        cslot = cslot == 2 ? 0 : cslot + 1;
        if (++kt < nk) continue;
        kt = 0;
        first = false;
        if (dbg & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(acc[i][j]));
            // (every register an asm load targets must be named by the wait: a register the compiler believes dead is re-used,
            // e.g. for an address, while the load is still in flight)
            if constexpr (kOp)
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(opn[0][0]), "+v"(opn[0][1]), "+v"(opn[1][0]), "+v"(opn[1][1]), "+v"(opn[2][0]), "+v"(opn[2][1]),
                               "+v"(opn[3][0]), "+v"(opn[3][1]), "+v"(biasv[0]), "+v"(biasv[1]), "+v"(rsc[0][0]), "+v"(rsc[0][EPI == EPI_FC2]),
                               "+v"(rsc[EPI == EPI_FC2][0]), "+v"(rsc[EPI == EPI_FC2][EPI == EPI_FC2]), "+v"(rsc[2 * (EPI == EPI_FC2)][0]),
                               "+v"(rsc[2 * (EPI == EPI_FC2)][EPI == EPI_FC2]), "+v"(rsc[3 * (EPI == EPI_FC2)][0]), "+v"(rsc[3 * (EPI == EPI_FC2)][EPI == EPI_FC2])
                             :: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(biasv[0]), "+v"(biasv[1])::"memory");
            vt += gridDim.x;
            continue;
        }
        // ================= epilogue of output tile vt: per wave, no workgroup barrier =================
        // every vector-memory operation from here to the end of the operand pipeline is an UNCONDITIONAL asm statement (rows /
        // columns beyond the matrix carry out-of-range buffer offsets), so the counted waits below are exact
        if constexpr (kOp) {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(opn[0][0]), "+v"(opn[0][1]), "+v"(opn[1][0]), "+v"(opn[1][1]), "+v"(opn[2][0]), "+v"(opn[2][1]),
                           "+v"(opn[3][0]), "+v"(opn[3][1]), "+v"(biasv[0]), "+v"(biasv[1])
                         :: "memory");
            if constexpr (EPI == EPI_FC2) {                        // (same drain: names the row-scale registers too)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) asm volatile("" : "+v"(rsc[tm][0]), "+v"(rsc[tm][1]));
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(biasv[0]), "+v"(biasv[1]) :: "memory");
        }
        {
            float bias[8], csum[8], csq[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bias[j] = __uint_as_float(biasv[0][j]);
                bias[4 + j] = __uint_as_float(biasv[1][j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) csum[j] = csq[j] = 0.f;
            const bool do_csum = (EPI == EPI_PLAIN || EPI == EPI_DG2) && d.colsum != nullptr;
            // C-output kinds: plain rows, or GA_C_UNPATCH2 (the downsample conv's data gradient): row m = (b, oy, ox), the 8-column
            // piece n = (tap, channel) goes to pixel (2 oy + tap / 2, 2 ox + tap % 2) of the NHWC gradient map
            const bool unpatch2 = EPI == EPI_PLAIN && d.c_kind == GA_C_UNPATCH2;
            const u32x4_s rsC = make_rsrc(reinterpret_cast<bf16_t*>(d.C) + (long)z * d.strideC,
                                          unpatch2 ? extent(4L * d.M, d.c_C, d.c_C, 2) : extent(d.M, d.ldc, d.N, 2));
            unsigned up_col = 0;
            if (unpatch2 && n_ok) {
                const int tap = n / d.c_C, ch = n - tap * d.c_C;
                up_col = (unsigned)(((tap >> 1) * d.c_W + (tap & 1)) * d.c_C + ch);
            }
            const u32x4_s rsC2 = make_rsrc((EPI == EPI_FC1 && d.C2) ? reinterpret_cast<bf16_t*>(d.C2) + (long)z * d.strideC : nullptr,
                                           (EPI == EPI_FC1 && d.C2) ? extent(d.M, d.ldc, d.N, 2) : 0u);
#pragma unroll
            for (int tm = 0; tm < 8; ++tm) {
                if constexpr (kOp) {                               // operand piece tm >= 4 was requested 4 pieces ago, after the stores of
                    if (tm >= 4) {                                 // piece tm-4: younger = 2 stores per later piece + 2 loads per later request
                        // fc2 requests 4 loads per piece (2 operand + 2 row-scale), dgrad2 2; every piece stores twice
                        constexpr int L = EPI == EPI_FC2 ? 4 : 2;
                        constexpr int ST = (dbg & 2) ? 0 : 2;    // (the no-store timing variant issues none)
                        constexpr int kYoung[4] = {3 * ST + 3 * L, 3 * ST + 2 * L, 3 * ST + L, 3 * ST};
                        constexpr int R = EPI == EPI_FC2 ? 1 : 0;
                        if (tm == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(opn[0][0]), "+v"(opn[0][1]), "+v"(rsc[0][0]), "+v"(rsc[0][R]) : "n"(kYoung[0]) : "memory");
                        if (tm == 5) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(opn[1][0]), "+v"(opn[1][1]), "+v"(rsc[R][0]), "+v"(rsc[R][R]) : "n"(kYoung[1]) : "memory");
                        if (tm == 6) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(opn[2][0]), "+v"(opn[2][1]), "+v"(rsc[2 * R][0]), "+v"(rsc[2 * R][R]) : "n"(kYoung[2]) : "memory");
                        if (tm == 7) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(opn[3][0]), "+v"(opn[3][1]), "+v"(rsc[3 * R][0]), "+v"(rsc[3 * R][R]) : "n"(kYoung[3]) : "memory");
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {                      // rows 8p .. 8p+7 of the 16-row accumulator tiles
                    // (no waits around the staging piece: LDS operations of ONE wave execute in issue order, so these writes follow
                    // the previous piece's reads and precede this piece's; hipcc keeps the order of may-alias LDS accesses)
                    if (((lane >> 3) & 1) == p) {
                        const int row = lane & 7;
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn) {
                            const int u = tn * 4 + (lane >> 4);
                            *reinterpret_cast<f32x4_t*>(stage + row * 64 + ((u ^ row) << 2)) = acc[tn][tm];
                        }
                    }
                    const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(stage + rr * 64 + (((2 * q) ^ rr) << 2));
                    const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(stage + rr * 64 + (((2 * q + 1) ^ rr) << 2));
                    const long m = mrow0 + tm * 16 + 8 * p;
                    float v[8] = {lo[0] + bias[0], lo[1] + bias[1], lo[2] + bias[2], lo[3] + bias[3],
                                  hi[0] + bias[4], hi[1] + bias[5], hi[2] + bias[6], hi[3] + bias[7]};
                    const bool live = n_ok && m < d.M;
                    unsigned coff = live ? (unsigned)(m * d.ldc + n) * 2u : kOob;
                    if (unpatch2 && live) {
                        const unsigned OW = (unsigned)d.c_W >> 1, OH = (unsigned)d.c_H >> 1;
                        const unsigned ox = (unsigned)m % OW, t = (unsigned)m / OW, oy = t % OH, b = t / OH;
                        coff = (unsigned)((((long)b * d.c_H + 2 * oy) * d.c_W + 2 * ox) * d.c_C + up_col) * 2u;
                    }
                    if constexpr (EPI == EPI_FC1) {
                        float w[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (dbg & 4) w[j] = v[j] * 0.5f;
                            else gelu_both_fast(v[j], v[j], w[j]);
                        }
                        if (!(dbg & 2)) r3_bstore16(rsC2, coff, r3_pack8(w));
                        else asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
                    }
                    if constexpr (EPI == EPI_DG2) {
                        float h[8];
                        unpack8(make_uint4(opn[tm & 3][p][0], opn[tm & 3][p][1], opn[tm & 3][p][2], opn[tm & 3][p][3]), h);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= h[j];
                    }
                    if constexpr (EPI == EPI_FC2) {
                        if (d.rowscale) {
                            const float sc = rsc[tm & 3][p];
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] *= sc;
                        }
                        float r[8];
                        unpack8(make_uint4(opn[tm & 3][p][0], opn[tm & 3][p][1], opn[tm & 3][p][2], opn[tm & 3][p][3]), r);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += r[j];
                    }
                    if (do_csum && live) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            csum[j] += v[j];
                            csq[j] += v[j] * v[j];
                        }
                    }
                    if (!(dbg & 2)) r3_bstore16(rsC, coff, r3_pack8(v));
                    else asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
                }
                if constexpr (kOp) {
                    if (tm < 4) {                                  // request piece tm+4 into the registers piece tm has just left
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const long m = mrow0 + (tm + 4) * 16 + 8 * p;
                            opn[tm][p] = r3_bload16(rsP, (n_ok && m < d.M) ? (unsigned)(m * ldp + n) * 2u : kOob);
                            if constexpr (EPI == EPI_FC2) rsc[tm][p] = r3_bload4(rsRS, rs_off(m));
                        }
                    }
                }
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tn][tm] = zero4;
            }
            if (do_csum) {                                         // lanes q, q+8, .. hold the same 8 columns over different rows
                // reduce over the 8 row lanes, then pass the wave's 64 column sums through the (idle) staging piece so that lane
                // l owns column l: ONE 256-byte atomic per statistic and tile instead of 8 instructions of 8 lanes each (the
                // narrow form made dgrad2 at M = 200,704 3x slower than its no-epilogue time: 150 k contended atomic instructions)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) {
                        csum[j] += __shfl_xor(csum[j], o, 64);
                        csq[j] += __shfl_xor(csq[j], o, 64);
                    }
                }
                if (rr == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        stage[q * 8 + j] = csum[j];
                        stage[64 + q * 8 + j] = csq[j];
                    }
                }
                const float a = stage[lane], b = stage[64 + lane];
                const int nc = tile_n * 128 + wn * 64 + lane;
                if (nc < d.N) {
                    atomicAdd(d.colsum + (long)z * d.strideCol + nc, a);
                    if (d.colsumsq) atomicAdd(d.colsumsq + (long)z * d.strideCol + nc, b);
                }
            }
        }
        vt += gridDim.x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (S == 0 cannot happen: grid <= tiles)
}

// ================================================================================================
// TN (wgrad) kernel: dW[n][k] = sum_m Y[m][n] X[m][k]
// LDS images are [rows = m][128 columns]; bf16: 64 rows x 256 B, fp32: 32 rows x 512 B (16 KiB each)
// ================================================================================================
template <typename T> struct TN {
    static constexpr int EPC = elt<T>::EPC;
    static constexpr int RB = 128 * (int)sizeof(T);   // row bytes
    static constexpr int ROWS = 16384 / RB;           // m rows per slab: 64 (bf16) / 32 (fp32)
    static constexpr int CPR = RB / 16;               // chunks per row: 16 / 32
    static constexpr int RSTEP = kThreads / CPR;      // rows covered by one pass of the 256 threads: 16 / 8
};

__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }

template <typename T>
__global__ __launch_bounds__(kThreads, 2) void gemm_tn_kernel(const ga_wgrad_desc d) {
    using P = TN<T>;
    constexpr int EPC = P::EPC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    // 1-D grid over (row split, output tile), walked XCD-aware: the output tiles of ONE row split re-read the same
    // Y / X row panels, so they are given to workgroups of one XCD (ids b, b+8, ...) and share them in its L2.
    const int tiles_k = (d.K + 127) / 128, tiles_n = (d.N + 127) / 128;
    const int ntiles = tiles_k * tiles_n;
    const int bid = xcd_remap(blockIdx.x, ntiles * d.split_m);
    const int split = bid / ntiles, tile = bid - split * ntiles;
    const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int z = blockIdx.z;
    // m range of this split, in whole slabs
    const long slabs = (d.M + P::ROWS - 1) / P::ROWS;
    const long per = (slabs + d.split_m - 1) / d.split_m;
    const long s_begin = (long)split * per;
    const long s_end = s_begin + per < slabs ? s_begin + per : slabs;
    if (s_begin >= s_end) return;

    const T* Yb = reinterpret_cast<const T*>(d.Y) + z * d.strideY;
    const int zx = d.x_batch_mod > 0 ? z % d.x_batch_mod : z;
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X) +
                              (d.x_kind == GA_A_STEM4_NCHW ? 4 : (long)sizeof(T)) * zx * d.strideX;

    const int cc = tid % P::CPR, rr = tid / P::CPR;  // chunk column, first row
    const int ny = n0 + cc * EPC;                    // Y column of this thread's chunks
    const bool yv = ny < d.N;
    const KCtx kx = make_k(d.x_kind, k0 + cc * EPC, d.K, d.x_H, d.x_W, d.x_C);

    uint4 ry[4], rx[4];
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    const bool do_bias = d.dbias != nullptr && tile_k == 0;

    auto g_load = [&](long slab) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = slab * P::ROWS + rr + P::RSTEP * i;
            ry[i] = (yv && m < d.M) ? *reinterpret_cast<const uint4*>(Yb + m * d.ldy + ny) : make_uint4(0, 0, 0, 0);
            const RowCtx r = make_row(d.x_kind, m, d.M, d.ldx, d.x_H, d.x_W, d.x_C);
            rx[i] = load_chunk<T>(d.x_kind, Xb, r, kx, d.x_H, d.x_W, d.x_C);
        }
    };
    auto s_store = [&](int buf) {
        unsigned char* Ys = smem + buf * 32768;
        unsigned char* Xs = Ys + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rr + P::RSTEP * i;
            const int off = sizeof(T) == 2 ? row * P::RB + ((cc ^ tn_swz(row)) << 4) : row * P::RB + (cc << 4);
            *reinterpret_cast<uint4*>(Ys + off) = ry[i];
            *reinterpret_cast<uint4*>(Xs + off) = act_chunk<T>(rx[i], d.x_act);
            if (do_bias) {
                if constexpr (sizeof(T) == 2) {
                    const unsigned w[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bsum[2 * j] += __uint_as_float(w[j] << 16);
                        bsum[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
                    }
                } else {
                    bsum[0] += __uint_as_float(ry[i].x); bsum[1] += __uint_as_float(ry[i].y);
                    bsum[2] += __uint_as_float(ry[i].z); bsum[3] += __uint_as_float(ry[i].w);
                }
            }
        }
    };

    f32x4_t acc[4][4];  // [tn][tk]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    g_load(s_begin);
    s_store(0);
    __syncthreads();
    for (long s = s_begin; s < s_end; ++s) {
        const int cur = (int)((s - s_begin) & 1);
        if (s + 1 < s_end) g_load(s + 1);
        const unsigned char* Ys = smem + cur * 32768;
        const unsigned char* Xs = Ys + 16384;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // 32 reduction rows per MFMA
                s16x4_t yf[4][2], xf[4][2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int row = ks * 32 + 8 * (lane >> 4) + 4 * hf + ((lane & 15) >> 2);
                    const int sw = tn_swz(row);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int ch_y = ((wn * 64 + t * 16) >> 3) + ((lane & 3) >> 1);
                        const int ch_x = ((wk * 64 + t * 16) >> 3) + ((lane & 3) >> 1);
                        const unsigned ay = row * P::RB + ((ch_y ^ sw) << 4) + 8 * (lane & 1);
                        const unsigned ax = row * P::RB + ((ch_x ^ sw) << 4) + 8 * (lane & 1);
                        yf[t][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4_t*)(Ys + ay));
                        xf[t][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4_t*)(Xs + ax));
                    }
                }
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int tk = 0; tk < 4; ++tk)
                        acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            *reinterpret_cast<const bf16x8_t*>(&yf[tn][0]), *reinterpret_cast<const bf16x8_t*>(&xf[tk][0]),
                            acc[tn][tk], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {  // 4 reduction rows per MFMA
                const int row = ks * 4 + (lane >> 4);
                float yf[4], xf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    yf[t] = *reinterpret_cast<const float*>(Ys + row * P::RB + (wn * 64 + t * 16 + (lane & 15)) * 4);
                    xf[t] = *reinterpret_cast<const float*>(Xs + row * P::RB + (wk * 64 + t * 16 + (lane & 15)) * 4);
                }
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int tk = 0; tk < 4; ++tk)
                        acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[tn], xf[tk], acc[tn][tk], 0, 0, 0);
            }
        }
        if (s + 1 < s_end) s_store(cur ^ 1);
        __syncthreads();
    }

    // ---- write out: D[n][k]; lane: k = lane&15 (col), n = (lane>>4)*4 + r (rows)
    float* W = d.dW + z * d.strideW;
    const bool atomic = d.accumulate || d.split_m > 1;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + tn * 16 + (lane >> 4) * 4 + r;
                const int k = k0 + wk * 64 + tk * 16 + (lane & 15);
                if (n < d.N && k < d.K) {
                    const float v = acc[tn][tk][r] * d.alpha;
                    if (atomic) atomicAdd(W + (long)n * d.ldw + k, v);
                    else W[(long)n * d.ldw + k] = v;
                }
            }
    if (do_bias) {  // reduce the per-thread column sums over the threads that share a chunk column
        float* red = reinterpret_cast<float*>(smem);  // [RSTEP][128]
#pragma unroll
        for (int j = 0; j < EPC; ++j) red[rr * 128 + cc * EPC + j] = bsum[j];
        __syncthreads();
        if (tid < 128 && n0 + tid < d.N) {
            float s = 0.f;
            for (int r = 0; r < P::RSTEP; ++r) s += red[r * 128 + tid];
            atomicAdd(d.dbias + z * d.strideDbias + n0 + tid, s * d.alpha);
        }
    }
}

// ================================================================================================
// TN kernel, wide form (bf16, plain operands, M % 32 == 0): 256 x 256 output tile, 8 waves (2 over n x 4 over k,
// 128 x 64 per wave), operands brought in by LDS-DMA (global_load_lds_dwordx4) into a ring of four 32 KiB stages:
// the loads of stages s+1..s+3 are in flight while stage s is multiplied, one barrier per stage.
// A stage = 32 reduction rows x {Y0, Y1, X0, X1}, each a [32][256 B] image in the same XOR layout as the narrow
// kernel (the DMA writes LDS lane-linearly, so the swizzle is applied to the SOURCE chunk each lane fetches).
// vs the 128 x 128 form: half the operand bytes per FLOP from L2 and no staging registers / ds_writes.
// Column sums of Y (bias gradient) ride along as one extra MFMA per fragment against a constant ones operand.
// ================================================================================================
constexpr int kTn2Threads = 512, kTn2Stage = 32768, kTn2Smem = 4 * kTn2Stage;

__global__ __launch_bounds__(kTn2Threads) void gemm_tn2_kernel(const ga_wgrad_desc d, const int split_m,
                                                               float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform roles stay in SGPRs
    const int wn = wave >> 2, wk = wave & 3;
    const int tiles_k = (d.K + 255) / 256, tiles_n = (d.N + 255) / 256;
    const int ntiles = tiles_k * tiles_n;
    const int bid = xcd_remap(blockIdx.x, ntiles * split_m);
    const int split = bid / ntiles, tile = bid - split * ntiles;
    const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
    const int n0 = tile_n * 256, k0 = tile_k * 256;
    const int z = blockIdx.z;
    const int stages = d.M / 32;
    const int per = (stages + split_m - 1) / split_m;
    const int s_begin = split * per;
    const int s_end = s_begin + per < stages ? s_begin + per : stages;
    if (s_begin >= s_end) return;

    // ---- DMA role: wave w fills sub-image w>>1 (0,1 = Y halves; 2,3 = X halves), rows 16*(w&1) + 4*i + (lane>>4)
    const int sub = wave >> 1;
    const bool is_y = sub < 2;
    const int ncols = is_y ? d.N : d.K;
    const int col0 = (is_y ? n0 : k0) + (sub & 1) * 128;          // first column of the sub-image
    const bool sub_live = col0 < ncols;                            // wholly out of range: never fetched, never used
    const long ld = is_y ? d.ldy : d.ldx;
    const int zx = d.x_batch_mod > 0 ? z % d.x_batch_mod : z;
    const bf16_t* gsrc = is_y ? reinterpret_cast<const bf16_t*>(d.Y) + z * d.strideY
                              : reinterpret_cast<const bf16_t*>(d.X) + zx * d.strideX;
    const int lrow = 16 * (wave & 1) + (lane >> 4);                // + 4*i
    const int pc = lane & 15;
    // source chunk for i even / odd pairs: swizzle = ((row&3)<<1) | (((row>>3)&1)<<3), row&3 = (lane>>4)&3, row bit 3 = (i>>1)&1
    const int sw_lo = ((lane >> 4) & 3) << 1;
    long goff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int c = col0 + ((pc ^ (sw_lo | (b << 3))) << 3);
        if (c >= ncols) c = 0;                                     // feeds output columns that are never written
        goff[b] = (long)lrow * ld + c;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // LDS-DMA through a buffer resource: 32-bit lane offsets (constant over the launch) + the stage's row offset as scalar
    const u32x4_s rs = make_rsrc(gsrc, 0xffffffffu);
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voff[i] = (unsigned)(goff[(i >> 1) & 1] + (long)(4 * i) * ld) * 2u;
    auto stage_load = [&](int s, int buf) {
        if (!sub_live) return;
        const unsigned soff = (unsigned)s * 64u * (unsigned)ld;                    // bytes of 32 rows
        const unsigned dst = lds0 + buf * kTn2Stage + sub * 8192 + (wave & 1) * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) blds16(rs, voff[i], soff, dst + i * 1024);
    };

    // ---- MFMA role
    int nv = (d.N - n0 - wn * 128 + 15) / 16; nv = nv < 0 ? 0 : (nv > 8 ? 8 : nv);   // live n fragments of this wave
    int kv = (d.K - k0 - wk * 64 + 15) / 16;  kv = kv < 0 ? 0 : (kv > 4 ? 4 : kv);   // live k fragments
    const bool do_bias = d.dbias != nullptr && tile_k == 0 && wk == 0;
    f32x4_t acc[8][4];
    float bsum[8];                             // column sums of Y over this lane's share of the reduction rows
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const bf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};

    const unsigned y_img = wn * 8192, x_img = 16384 + (wk >> 1) * 8192;
    const int xc0 = (wk & 1) * 8;                                  // first chunk of this wave's 64 X columns
    // fragment addresses: the XOR term depends on the lane only (row & 3 and bit 3 of the row are the same for both
    // 4-row halves and both k-steps), so one address per fragment column + immediates for half / k-step / buffer
    unsigned yaddr[8], xaddr[4];
    {
        const int row = 8 * (lane >> 4) + ((lane & 15) >> 2);
        const int sw = tn_swz(row);
        const unsigned rb = row * 256 + 8 * (lane & 1);
#pragma unroll
        for (int t = 0; t < 8; ++t) yaddr[t] = y_img + rb + (((2 * t + ((lane & 3) >> 1)) ^ sw) << 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) xaddr[t] = x_img + rb + (((xc0 + 2 * t + ((lane & 3) >> 1)) ^ sw) << 4);
    }
    // one stage = one k-step of 32 reduction rows.  Fragments beyond N / K are multiplied too (their inputs are
    // clamped loads, their outputs are never written): no predicates in the loop.
    auto compute = [&](int cur) {
        const unsigned char* st = smem + cur * kTn2Stage;
        s16x4_t yf[8][2], xf[4][2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                xf[t][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(st + xaddr[t] + hf * 1024));
#pragma unroll
            for (int t = 0; t < 8; ++t)
                yf[t][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4_t*)(st + yaddr[t] + hf * 1024));
        }
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) {
            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(&yf[tn][0]);
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
                acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    a, *reinterpret_cast<const bf16x8_t*>(&xf[tk][0]), acc[tn][tk], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int tn = 0; tn < 8; ++tn) {
                const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(&yf[tn][0]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bsum[tn] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{a[2 * j], a[2 * j + 1]}, ones2, bsum[tn], false);
            }
        }
    };

    // 4-slot ring, three stages in flight: at iteration `it` the wave's own DMA of stages it+1 and it+2 (4 loads each) may
    // still be outstanding when stage `it` is consumed -- a COUNTED wait, the loads span the barrier
    const int nst = s_end - s_begin;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (j < nst) stage_load(s_begin + j, j);
    for (int it = 0; it < nst; ++it) {
        const int ahead = nst - 1 - it;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // stage `it` has landed for every wave; slot (it-1)&3 is no longer read
        if (it + 3 < nst) stage_load(s_begin + it + 3, (it + 3) & 3);
        if (nv > 0 && kv > 0) compute(it & 3);
    }

    // ---- write out: D[n][k]; lane: k = lane&15 (col), n = (lane>>4)*4 + r (rows).  With row splits every workgroup
    // stores its partial tile to part[split][N][K] (plain stores; tn2_reduce sums the splits): 20 splits adding 256 KiB
    // tiles into the same 2 MB with fp32 atomics took as long as the multiplication itself.
    float* W = part ? part + ((long)z * split_m + split) * d.N * d.K : d.dW + z * d.strideW;
    const long ldw = part ? d.K : d.ldw;
    const float alpha = part ? 1.f : d.alpha;
    const bool atomic = !part && (d.accumulate || split_m > 1);
#pragma unroll
    for (int tn = 0; tn < 8; ++tn)
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 128 + tn * 16 + (lane >> 4) * 4 + r;
                const int k = k0 + wk * 64 + tk * 16 + (lane & 15);
                if (n < d.N && k < d.K) {
                    const float v = acc[tn][tk][r] * alpha;
                    if (atomic) atomicAdd(W + (long)n * ldw + k, v);
                    else W[(long)n * ldw + k] = v;
                }
            }
    if (do_bias) {                            // lane holds column n = tn*16 + (lane&15); the 4 lane groups hold row subsets
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) {
            float v = bsum[tn];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = n0 + wn * 128 + tn * 16 + lane;
            if (lane < 16 && n < d.N) atomicAdd(d.dbias + z * d.strideDbias + n, v * d.alpha);
        }
    }
}

// dW[n][k] += alpha * sum_s part[s][n][k]   (N*K multiple of 4: N, K are multiples of 8)
__global__ __launch_bounds__(256) void tn2_reduce_kernel(const float* __restrict__ part, int nsplit, long nk, int K,
                                                         float alpha, float* __restrict__ dW, long ldw, long strideW) {
    const long z = blockIdx.y;
    const float* p = part + z * nsplit * nk;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < nk; i += (long)gridDim.x * 1024) {
        float4 a = *reinterpret_cast<const float4*>(p + i);
#pragma unroll 4
        for (int s = 1; s < nsplit; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(p + (long)s * nk + i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const long n = i / K, k = i - n * K;
        float* dst = dW + z * strideW + n * ldw + k;
        dst[0] += alpha * a.x; dst[1] += alpha * a.y; dst[2] += alpha * a.z; dst[3] += alpha * a.w;
    }
}

// the wide form needs whole 32-row stages, plain bf16 operands, and an output that is accumulated into (so that it
// may choose its own row split); it pays once the reduction is long enough to amortise the 256 x 256 tile
bool tn2_eligible(const ga_wgrad_desc* d) {
    return GA_KNOB("TN2", 1) && d->dtype == GA_BF16 && d->x_kind == GA_A_PLAIN && d->x_act == GA_ACT_NONE && d->M % 32 == 0 &&
           d->M >= 8192 && (d->accumulate || d->split_m > 1) && (long)d->M * d->ldy < (1L << 31) && (long)d->M * d->ldx < (1L << 31);   // 32-bit byte offsets
}

// row split of the wide form and the bytes of partial-tile workspace it wants (0: combine with atomics)
size_t tn2_plan(const ga_wgrad_desc* d, int* split_out) {
    const int tiles = cdiv(d->N, 256) * cdiv(d->K, 256) * d->batch;
    const int stages = d->M / 32;
    const int wg_budget = GA_KNOB("TN2_WGS", 0);
    // 3/4 of the CUs: in the train step these launches share the chip with the dgrad chain (asynchronous lane), and
    // fewer row splits mean fewer partial tiles to write and reduce (same-box A/B: 192 vs 256 workgroups -0.13 ms/step)
    const int cus = wg_budget > 0 ? wg_budget : num_cus() * 3 / 4;
    int split = std::max(1, std::min(stages / 8, cus / tiles));               // one workgroup per CU
    split = cdiv(stages, cdiv(stages, split));                                // no empty row range
    *split_out = split;
    const long nk = (long)d->N * d->K;
    const long nk_min = GA_KNOB("TN2_PART_MIN", 65536);      // 256 x 256 outputs (CSWin proj) included: -0.2 ms/step there, neutral elsewhere
    return (split > 1 && nk >= nk_min) ? (size_t)d->batch * split * nk * sizeof(float) : 0;   // small outputs: atomics are cheaper
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// persistent grid = exactly the number of workgroups that are resident at once (occupancy query per variant),
// rounded down to a multiple of 8 so every XCD gets the same share of the tile walk
template <typename T, int TNW, int NWM, bool PLAIN, bool PRE, int EPI, bool DMA = false>
void launch_nt_(const ga_gemm_desc* d, hipStream_t s) {
    using CF = NTCfg<TNW, NWM>;
    constexpr int smem = DMA ? CF::SMEM_DMA : CF::SMEM;
    auto kern = gemm_nt_kernel<T, TNW, NWM, PLAIN, PRE, EPI, DMA>;
    static const int per_cu = [&] {
        int n = 0;
        if (smem > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, CF::NTHR, smem) != hipSuccess || n < 1) n = 1;
        return n;
    }();
    if (per_cu == 0) {
        ga_set_error("ga_gemm: cannot reserve %d bytes of LDS", smem);
        return;
    }
    const int tiles = cdiv(d->M, CF::BM) * cdiv(d->N, CF::BN);
    const int cap = std::max(8, (per_cu * num_cus() / d->batch) / 8 * 8);
    dim3 grid(std::min(tiles, cap), 1, d->batch), block(CF::NTHR);
    hipLaunchKernelGGL(kern, grid, block, smem, s, *d);
}

template <int EPI>
void launch_nt_pp(const ga_gemm_desc* d, hipStream_t s) {
    auto kern = gemm_nt_pp_kernel<EPI>;
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kPPSmem) == hipSuccess;
    if (!ok) {
        ga_set_error("ga_gemm: cannot reserve %d bytes of LDS", kPPSmem);
        return;
    }
    const int tiles = cdiv(d->M, 256) * cdiv(d->N, 256);
    const int cap = std::max(8, (num_cus() / d->batch) / 8 * 8);
    dim3 grid(std::min(tiles, cap), 1, d->batch), block(kPPThreads);
#ifdef GAEXT_DEBUG
    const int dbg = GA_KNOB("PP_DBG", 0);         // timing experiments of the body (RESULTS DELIBERATELY WRONG): debug builds only
#else
    const int dbg = 0;
#endif
    hipLaunchKernelGGL(kern, grid, block, kPPSmem, s, *d, dbg);
}

template <int EPI, int DBG = 0, int AK = 0>
void launch_nt_r3_(const ga_gemm_desc* d, hipStream_t s) {
    auto kern = gemm_nt_r3_kernel<EPI, DBG, AK>;
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kR3Smem) == hipSuccess;
    if (!ok) {
        ga_set_error("ga_gemm: cannot reserve %d bytes of LDS", kR3Smem);
        return;
    }
    const int tiles = cdiv(d->M, 256) * cdiv(d->N, 128);
    const int cap = std::max(8, (2 * num_cus() / d->batch) / 8 * 8);       // two workgroups per CU
    dim3 grid(std::min(tiles, cap), 1, d->batch), block(kR3Threads);
    // start skew of the second workgroup of a CU, in units of 64 clocks.  Off by default: measured 0, 40, 80, 160, 240 on the
    // MLP shapes of stages 1-3 (tools/r3_ab.py, R3_ARMS): within +-3 % of each other, no trend -- the two workgroups of a CU do
    // not run in lockstep, their phases add up because each is bound by the CU's shared issue / LDS paths
    const int stagger = std::max(0, GA_KNOB("R3_STAGGER", 0));
    hipLaunchKernelGGL(kern, grid, block, kR3Smem, s, *d, stagger);
}
template <int EPI>
void launch_nt_r3(const ga_gemm_desc* d, hipStream_t s) {
#ifdef GAEXT_DEBUG
    switch (GA_KNOB("R3_DBG", 0)) {     // the timing-experiment variants exist in debug builds only
        case 1: return launch_nt_r3_<EPI, 1>(d, s);
        case 2: return launch_nt_r3_<EPI, 2>(d, s);
        case 4: return launch_nt_r3_<EPI, 4>(d, s);
        case 6: return launch_nt_r3_<EPI, 6>(d, s);
        case 8: return launch_nt_r3_<EPI, 8>(d, s);
        case 10: return launch_nt_r3_<EPI, 10>(d, s);
        case 17: return launch_nt_r3_<EPI, 17>(d, s);
        case 33: return launch_nt_r3_<EPI, 33>(d, s);
        default: break;
    }
#endif
    launch_nt_r3_<EPI, 0>(d, s);
}

// 3-slot ring form (256 x 128 tiles, two workgroups per CU): NT_R3 = bit mask of epilogues (1 plain, 2 fc1, 4 fc2, 8 dgrad2);
// -1 = the heuristic below
bool want_pp(const ga_gemm_desc* d, int epi);
bool want_r3(const ga_gemm_desc* d, int epi) {
    const int r3 = GA_KNOB("NT_R3", -1);
    const bool forced = r3 >= 0;
    const int mask = forced ? r3 : 15;
    const bool neigh2 = d->a_kind == GA_A_NEIGH2;       // 2 x 2 neighbourhoods (data gradient of the 3 x 3 / stride-2 convs): plain epilogue only
    const bool conv3s2 = d->a_kind == GA_A_CONV3S2;     // the 3 x 3 / stride-2 convs on even maps: plain epilogue only
    const bool patch2 = d->a_kind == GA_A_PATCH2 || neigh2 || conv3s2;   // 2 x 2 / stride-2 patches (downsample convs): plain epilogue only
    if (patch2 && (epi != EPI_PLAIN || d->a_C % 16 != 0 || d->K != (conv3s2 ? 9 : 4) * d->a_C || 4L * d->M * d->a_C >= (1L << 30) ||
                   d->a_batch_mod || d->batch != 1))
        return false;
    if (neigh2 && (d->a_C % 32 != 0 || (long)d->M % ((long)d->a_H * d->a_W) != 0 || !GA_KNOB("NT_R3_NEIGH2", 1))) return false;
    if (conv3s2 && (d->a_C % 32 != 0 || d->a_H % 2 != 0 || d->a_W % 2 != 0 || d->a_W < 4 ||
                    (long)d->M % ((long)(d->a_H / 2) * (d->a_W / 2)) != 0 || !GA_KNOB("NT_R3_CONV3S2", 1)))
        return false;
    if (!mask || d->dtype != GA_BF16 || (d->a_kind != GA_A_PLAIN && !patch2) || epi == EPI_GENERIC || !((mask >> epi) & 1)) return false;
    if (d->N % 8 != 0 || d->K % 8 != 0 || d->K < 64 || (!patch2 && d->lda % 8 != 0) || d->ldb % 8 != 0 || d->ldc % 8 != 0) return false;
    if ((!patch2 && (long)d->M * d->lda >= (1L << 30)) || (long)d->N * d->ldb >= (1L << 30)) return false;   // 32-bit byte offsets
    if ((reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B) | reinterpret_cast<uintptr_t>(d->C)) & 15) return false;
    const bool unpatch2 = d->c_kind == GA_C_UNPATCH2;   // scatter of the downsample conv's data gradient: plain epilogue only
    if (unpatch2 && (epi != EPI_PLAIN || d->colsum || d->c_C % 8 != 0 || d->N != 4 * d->c_C || 4L * d->M * d->c_C >= (1L << 30) || d->batch != 1))
        return false;
    if ((!unpatch2 && (long)d->M * d->ldc >= (1L << 30)) || (d->c_kind != GA_C_PLAIN && !unpatch2) || d->c_f32) return false;
    if (epi == EPI_FC1 && d->C2 && (reinterpret_cast<uintptr_t>(d->C2) & 15)) return false;
    if (epi == EPI_DG2 && (d->ldh % 8 != 0 || (reinterpret_cast<uintptr_t>(d->H) & 15) || (long)d->M * d->ldh >= (1L << 30))) return false;
    if (epi == EPI_FC2 && (d->ldr % 8 != 0 || (reinterpret_cast<uintptr_t>(d->R) & 15) || (long)d->M * d->ldr >= (1L << 30))) return false;
    if (d->bias && (reinterpret_cast<uintptr_t>(d->bias) & 3)) return false;
    if (forced) return true;
    // heuristic from same-process A/B rounds against the other forms (tools/r3_ab.py, gpurun_out/r03/r3_ab*.log; MI355X):
    //   fc1 / fc2 / dgrad2 epilogues at M = 6,272 .. 200,704, K = 192 .. 3072: x1.04 .. 1.76 everywhere measured
    //   plain: ahead for K <= 512 (x1.13 .. 1.18) and for the K = 768 .. 2208 launches of the heads (x1.03 .. 1.10); behind the
    //   8-wave ping-pong body on very wide / very long / very tall launches (N 2208: x0.91, K 3072: x0.83, 8192^3: x0.88,
    //   M 73,856 of the ViT trunk: x0.91 .. 0.97) and behind the 128-column forms at N < 384 with a mid-length K (x0.95)
    // gather kinds: the alternative is the register-staged gather (110-240 TFLOP/s on these launches, 0.105 ms for the 100 tiles
    // of merge3's half-batch forward against 0.03 here)
    if (neigh2 || conv3s2) return (long)cdiv(d->M, 256) * cdiv(d->N, 128) >= 16;
    if ((long)cdiv(d->M, 256) * cdiv(d->N, 128) * d->batch < num_cus() / 2) return false;      // too few tiles to fill the chip
    const bool pp = want_pp(d, epi);
    if (d->M >= 65536 && pp) return false;
    if (epi != EPI_PLAIN) return true;
    if (d->K <= 512) return true;
    if (pp) return d->N < 2048 && d->K < 3072;
    return d->N >= 384 ? d->K < 3072 : d->K >= 1024;
}

// 8-wave ping-pong form: GAEXT_NT_PP = bit mask of epilogues (1 plain, 2 fc1, 4 fc2, 8 dgrad2); unset: plain / fc1 / fc2, for
// launches whose K loop is long enough to carry the un-overlapped epilogue (K >= GAEXT_NT_PP_MINK, default 512) and whose
// last column tile is not mostly empty
bool want_pp(const ga_gemm_desc* d, int epi) {
    const int pp = GA_KNOB("NT_PP", -1);    // -1: heuristic
    const bool e = pp >= 0;
    const int mask = e ? pp : 7;            // (dgrad2: its stored-GELU' operand is read inside the un-overlapped epilogue: measured slower)
    const int mink = GA_KNOB("NT_PP_MINK", 512);
    if (!mask || d->dtype != GA_BF16 || d->a_kind != GA_A_PLAIN || epi == EPI_GENERIC || !((mask >> epi) & 1)) return false;
    if (d->N % 8 != 0 || d->K % 8 != 0 || d->K < (e ? 256 : mink) || d->lda % 8 != 0 || d->ldb % 8 != 0 || d->ldc % 8 != 0) return false;
    if ((long)d->M * d->lda >= (1L << 30) || (long)d->N * d->ldb >= (1L << 30) || d->lda < 64 || d->ldb < 64) return false;   // 32-bit byte offsets
    if ((reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B) | reinterpret_cast<uintptr_t>(d->C)) & 15) return false;
    if (epi == EPI_FC1 && d->C2 && (reinterpret_cast<uintptr_t>(d->C2) & 15)) return false;
    if (epi == EPI_DG2 && (d->ldh % 8 != 0 || (reinterpret_cast<uintptr_t>(d->H) & 15))) return false;
    if (epi == EPI_FC2 && (d->ldr % 8 != 0 || (reinterpret_cast<uintptr_t>(d->R) & 15))) return false;
    const int tn = cdiv(d->N, 256);
    if (!e && tn * 256 - d->N > tn * 256 / 8) return false;        // > 12.5 % of the column tiles' MFMA work on columns that do not exist
    return (long)cdiv(d->M, 256) * tn * d->batch >= num_cus() / 2;
}

// pick the compile-time epilogue when the launch matches one of the hot shapes of the training step
int classify_epilogue(const ga_gemm_desc* d, bool allow_patch2 = false) {
    if ((d->a_kind != GA_A_PLAIN && !(allow_patch2 && (d->a_kind == GA_A_PATCH2 || d->a_kind == GA_A_NEIGH2 || d->a_kind == GA_A_CONV3S2))) ||
        d->a_act != GA_ACT_NONE ||
        d->alpha != 1.0f ||
        (d->c_kind != GA_C_PLAIN && !(allow_patch2 && d->c_kind == GA_C_UNPATCH2)) || d->c_f32 ||
        d->relu_after)
        return EPI_GENERIC;
    const bool act0 = d->act == GA_ACT_NONE;
    if (d->act == GA_ACT_GELU && (!d->C2 || d->c2_mode == 2) && !d->H && !d->R && !d->rowscale && !d->colsum) return EPI_FC1;
    if (act0 && !d->C2 && !d->H && d->R && !d->colsum) return EPI_FC2;
    if (act0 && !d->C2 && d->H && d->h_is_deriv && !d->R && !d->rowscale) return EPI_DG2;
    if (act0 && !d->C2 && !d->H && !d->R && !d->rowscale) return EPI_PLAIN;
    return EPI_GENERIC;
}

// 256-row tiles (8 waves, one workgroup per CU) for the two epilogues that carry a prefetched epilogue operand
// (fc2: + shortcut, dgrad2: * gelu'): their 4-wave form sits at 160-170 VGPRs = 2 workgroups per CU, and the wide
// tile reads the weight slab once per 256 rows.  Measured on MI355X (tools/gemm_bench.py): dgrad2 1.35-1.45x,
// fc2 1.1-1.2x; the plain / fc1 epilogues (120 VGPRs, 4 workgroups per CU) are 5-15 % SLOWER with it.
bool want_big_tile(const ga_gemm_desc* d, int epi) {
    const int force = GA_KNOB("NT_BIG", -1);      // 0 / 1 override for experiments; -1 = heuristic
    if (d->dtype != GA_BF16 || d->a_kind != GA_A_PLAIN || (epi != EPI_FC2 && epi != EPI_DG2)) return false;
    if (force >= 0) return force != 0;
    return (long)cdiv(d->M, 256) * cdiv(d->N, 128) * d->batch >= 2L * num_cus();
}

// LDS-DMA form: 256-row tiles, plain bf16 operands, one of the compile-time epilogues
// the LDS-DMA forms address their operands with 32-bit byte offsets from the matrix base
bool dma_offsets_fit(const ga_gemm_desc* d) { return (long)d->M * d->lda < (1L << 30) && (long)d->N * d->ldb < (1L << 30); }

bool want_dma(const ga_gemm_desc* d, int epi, int tnw) {
    const int mode = GA_KNOB("NT_DMA", 1);        // 0 off, 1 heuristic (default), 2 every eligible launch
    if (!mode || d->dtype != GA_BF16 || d->a_kind != GA_A_PLAIN || epi == EPI_GENERIC || (tnw != 4 && tnw != 3) || !dma_offsets_fit(d)) return false;
    if ((long)cdiv(d->M, 256) * cdiv(d->N, 32 * tnw) * d->batch < num_cus()) return false;
    // measured (tools/gemm_bench.py): ahead only for the fc2 epilogue with a long reduction (K >= 1024, +8..20 %);
    // mode 2 forces it on every eligible launch (tests, experiments)
    return mode == 2 || (epi == EPI_FC2 && tnw == 4 && d->K >= 1024);
}

// 128 x 128 tile, 4 waves, LDS-DMA into a 2-slot ring, 80 KiB: two workgroups per CU without the ds_write staging pass
bool want_dma2(const ga_gemm_desc* d, int epi) {
    // NT_DMA2: bit mask of epilogues (1 plain, 2 fc1, 4 fc2, 8 dgrad2)
    // unset: fc1 and dgrad2 (stage-2 shapes, same box, after the DMA went through buffer resources: fc1 0.126 -> 0.119 ms,
    // dgrad2 0.137 -> 0.118; fc2 / dgrad1 are 3-5 % slower with it and keep the register-staged form)
    const int mask = GA_KNOB("NT_DMA2", 10);
    if (!mask || d->dtype != GA_BF16 || d->a_kind != GA_A_PLAIN || epi == EPI_GENERIC || !dma_offsets_fit(d)) return false;
    if (d->K < GA_KNOB("NT_DMA2_MINK", 256)) return false;
    return (mask >> epi) & 1;
}

// 256 x 256 tile, 8 waves (64 x 128 each), LDS-DMA into a 2-slot ring: wide-N launches
bool want_t256(const ga_gemm_desc* d, int epi) {
    const int t256 = GA_KNOB("NT_T256", -1);      // bit mask of epilogues (1 plain, 2 fc1, 4 fc2, 8 dgrad2); -1: heuristic
    // unset: every epilogue, but only for the very tall launches (M >= 65536: the ViT trunk's 73,856 token rows, -4.8 % on the
    // MAP-ViT-B/384 step); on the ConvNeXt / CSWin stage-2/3 shapes (M = 50,176) the form measured -12 .. +5 % and stays off
    const int mask = t256 >= 0 ? t256 : ((d->M >= 65536 && d->N >= 768) ? 15 : 0);      // (N >= 768: the CSWin stem's N = 256 launches lose 2 %)
    if (!mask || d->dtype != GA_BF16 || d->a_kind != GA_A_PLAIN || epi == EPI_GENERIC || !dma_offsets_fit(d)) return false;
    if (d->N % 256 != 0 || d->K < 256) return false;
    if ((long)cdiv(d->M, 256) * (d->N / 256) * d->batch < num_cus()) return false;
    return (mask >> epi) & 1;
}

template <typename T, int TNW, int NWM>
void launch_nt(const ga_gemm_desc* d, hipStream_t s) {
    constexpr bool BF = sizeof(T) == 2;   // the epilogue-operand prefetch exists for bf16 only (32 extra VGPRs)
    if (d->a_kind != GA_A_PLAIN) {
        launch_nt_<T, TNW, NWM, false, false, EPI_GENERIC>(d, s);
        return;
    }
    switch (classify_epilogue(d)) {
        case EPI_PLAIN: launch_nt_<T, TNW, NWM, true, false, EPI_PLAIN>(d, s); break;
        case EPI_FC1: launch_nt_<T, TNW, NWM, true, false, EPI_FC1>(d, s); break;
        case EPI_FC2: launch_nt_<T, TNW, NWM, true, BF, EPI_FC2>(d, s); break;
        case EPI_DG2: launch_nt_<T, TNW, NWM, true, BF, EPI_DG2>(d, s); break;
        default:
            if (BF && (d->H || d->R)) launch_nt_<T, TNW, NWM, true, BF, EPI_GENERIC>(d, s);
            else launch_nt_<T, TNW, NWM, true, false, EPI_GENERIC>(d, s);
    }
}

}  // namespace

extern "C" int ga_gemm(const ga_gemm_desc* d, ga_stream_t stream) {
    GA_REQUIRE(d && d->A && d->B && d->C, "ga_gemm: null operand");
    GA_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1, "ga_gemm: bad shape M=%d N=%d K=%d batch=%d", d->M,
               d->N, d->K, d->batch);
    GA_REQUIRE(d->dtype == GA_F32 || d->dtype == GA_BF16, "ga_gemm: bad dtype %d", d->dtype);
    const int epc = d->dtype == GA_BF16 ? 8 : 4;
    GA_REQUIRE(aligned16(d->A) && aligned16(d->B) && aligned16(d->C), "ga_gemm: operands must be 16-byte aligned");
    GA_REQUIRE(d->ldb % epc == 0 && d->strideB % epc == 0, "ga_gemm: ldb/strideB must be multiples of %d", epc);
    GA_REQUIRE(d->K % epc == 0, "ga_gemm: K=%d must be a multiple of %d (pad the operand)", d->K, epc);
    if (d->a_kind == GA_A_PLAIN) {
        GA_REQUIRE(d->lda % epc == 0 && d->strideA % epc == 0, "ga_gemm: lda/strideA must be multiples of %d", epc);
    } else if (d->a_kind == GA_A_PATCH2) {
        GA_REQUIRE(d->a_C % epc == 0 && d->a_H % 2 == 0 && d->a_W % 2 == 0 && d->K == 4 * d->a_C &&
                       (long)d->M % ((d->a_H / 2) * (d->a_W / 2)) == 0,
                   "ga_gemm: PATCH2 needs C%%%d==0, even H,W, K==4C", epc);
    } else if (d->a_kind == GA_A_CONV3) {
        GA_REQUIRE(d->a_C % epc == 0 && d->K == 9 * d->a_C && (long)d->M % (d->a_H * d->a_W) == 0,
                   "ga_gemm: CONV3 needs C%%%d==0, K==9C", epc);
    } else if (d->a_kind == GA_A_CONV3S2) {
        GA_REQUIRE(d->a_C % epc == 0 && d->K == 9 * d->a_C && (long)d->M % (((d->a_H + 1) / 2) * ((d->a_W + 1) / 2)) == 0,
                   "ga_gemm: CONV3S2 needs C%%%d==0, K==9C", epc);
    } else if (d->a_kind == GA_A_NEIGH2) {
        GA_REQUIRE(d->a_C % epc == 0 && d->K == 4 * d->a_C && (long)d->M % (d->a_H * d->a_W) == 0,
                   "ga_gemm: NEIGH2 needs C%%%d==0, K==4C", epc);
    } else if (d->a_kind == GA_A_STEM4_NCHW) {
        GA_REQUIRE(d->a_C == 3 && d->K == 48 && d->a_H % 4 == 0 && d->a_W % 4 == 0, "ga_gemm: STEM4 needs C=3,K=48");
    } else {
        GA_REQUIRE(false, "ga_gemm: bad a_kind %d", d->a_kind);
    }
    if (d->c_kind == GA_C_UNPATCH2) {
        GA_REQUIRE(d->c_C % 8 == 0 && d->N == 4 * d->c_C && !d->c_f32, "ga_gemm: UNPATCH2 needs N==4*c_C, c_C%%8==0");
    } else {
        GA_REQUIRE(d->c_kind == GA_C_PLAIN, "ga_gemm: bad c_kind");
    }
    if (d->H) GA_REQUIRE(aligned16(d->H) && d->ldh % 8 == 0, "ga_gemm: H alignment");
    if (d->C2) GA_REQUIRE(aligned16(d->C2) && d->c_kind == GA_C_PLAIN && !d->c_f32 && (d->c2_mode == 1 || d->c2_mode == 2),
                          "ga_gemm: C2 needs a plain, dtype-typed C and c2_mode 1|2");
    if (d->R) GA_REQUIRE(aligned16(d->R) && d->ldr % 8 == 0, "ga_gemm: R alignment");
    if (d->rowscale) GA_REQUIRE(d->rows_per_scale > 0, "ga_gemm: rows_per_scale");
    // vector stores need an 8-element aligned leading dimension; otherwise every piece takes the scalar path,
    // which the kernel selects per piece only at the N edge -> require it here.
    GA_REQUIRE(d->c_kind != GA_C_PLAIN || d->ldc % 8 == 0, "ga_gemm: ldc=%ld must be a multiple of 8", (long)d->ldc);
    // N-tile width: 128 when it divides N, else 96 (stage-0 C = 96, concat 2208 = 23*96), else 64; ragged N -> least waste
    int tnw;
    if (d->N % 128 == 0) tnw = 4;
    else if (d->N % 96 == 0) tnw = 3;
    else if (d->N % 64 == 0) tnw = 2;
    else {
        const long w4 = (long)cdiv(d->N, 128) * 128, w3 = (long)cdiv(d->N, 96) * 96, w2 = (long)cdiv(d->N, 64) * 64;
        tnw = (w4 <= w3 && w4 <= w2) ? 4 : (w3 <= w2 ? 3 : 2);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool bf = d->dtype == GA_BF16;
#define GA_LAUNCH_NT(TNW, NWM)                          \
    do {                                               \
        if (bf) launch_nt<bf16_t, TNW, NWM>(d, s);     \
        else launch_nt<float, TNW, NWM>(d, s);         \
    } while (0)
    if (d->a_kind == GA_A_CONV3 && ga_conv3_c64_try(d, s)) return ga_check_launch("ga_gemm");     // 64 -> 64 channels: direct convolution
    if (d->a_kind == GA_A_CONV3S2 && d->a_C == 8 && ga_conv0_c8_try(d, s)) return ga_check_launch("ga_gemm");   // 3 (8) -> 64: first conv of the deep stem
    const int epi = classify_epilogue(d);
    if ((d->a_kind == GA_A_PATCH2 || d->a_kind == GA_A_NEIGH2 || d->a_kind == GA_A_CONV3S2 || d->c_kind == GA_C_UNPATCH2) &&
        classify_epilogue(d, true) == EPI_PLAIN &&
        want_r3(d, EPI_PLAIN)) {
        if (d->a_kind == GA_A_NEIGH2 || d->a_kind == GA_A_CONV3S2) launch_nt_r3_<EPI_PLAIN, 0, 1>(d, s);
        else launch_nt_r3<EPI_PLAIN>(d, s);          // downsample conv (2 x 2 / stride 2) straight from the NHWC map / its data gradient
        return ga_check_launch("ga_gemm");
    }
    if (want_r3(d, epi)) {
        switch (epi) {
            case EPI_PLAIN: launch_nt_r3<EPI_PLAIN>(d, s); break;
            case EPI_FC1: launch_nt_r3<EPI_FC1>(d, s); break;
            case EPI_FC2: launch_nt_r3<EPI_FC2>(d, s); break;
            default: launch_nt_r3<EPI_DG2>(d, s); break;
        }
    } else if (want_dma(d, epi, tnw)) {
        if (tnw == 4) {
            switch (epi) {
                case EPI_PLAIN: launch_nt_<bf16_t, 4, 4, true, false, EPI_PLAIN, true>(d, s); break;
                case EPI_FC1: launch_nt_<bf16_t, 4, 4, true, false, EPI_FC1, true>(d, s); break;
                case EPI_FC2: launch_nt_<bf16_t, 4, 4, true, true, EPI_FC2, true>(d, s); break;
                default: launch_nt_<bf16_t, 4, 4, true, true, EPI_DG2, true>(d, s); break;
            }
        } else {
            switch (epi) {
                case EPI_PLAIN: launch_nt_<bf16_t, 3, 4, true, false, EPI_PLAIN, true>(d, s); break;
                case EPI_FC1: launch_nt_<bf16_t, 3, 4, true, false, EPI_FC1, true>(d, s); break;
                case EPI_FC2: launch_nt_<bf16_t, 3, 4, true, true, EPI_FC2, true>(d, s); break;
                default: launch_nt_<bf16_t, 3, 4, true, true, EPI_DG2, true>(d, s); break;
            }
        }
    } else if (want_pp(d, epi)) {
        switch (epi) {
            case EPI_PLAIN: launch_nt_pp<EPI_PLAIN>(d, s); break;
            case EPI_FC1: launch_nt_pp<EPI_FC1>(d, s); break;
            case EPI_FC2: launch_nt_pp<EPI_FC2>(d, s); break;
            default: launch_nt_pp<EPI_DG2>(d, s); break;
        }
    } else if (want_t256(d, epi)) {
        switch (epi) {
            case EPI_PLAIN: launch_nt_<bf16_t, 8, 4, true, false, EPI_PLAIN, true>(d, s); break;
            case EPI_FC1: launch_nt_<bf16_t, 8, 4, true, false, EPI_FC1, true>(d, s); break;
            case EPI_FC2: launch_nt_<bf16_t, 8, 4, true, false, EPI_FC2, true>(d, s); break;
            default: launch_nt_<bf16_t, 8, 4, true, false, EPI_DG2, true>(d, s); break;
        }
    } else if (tnw == 4 && want_dma2(d, epi)) {
        switch (epi) {
            case EPI_PLAIN: launch_nt_<bf16_t, 4, 2, true, false, EPI_PLAIN, true>(d, s); break;
            case EPI_FC1: launch_nt_<bf16_t, 4, 2, true, false, EPI_FC1, true>(d, s); break;
            case EPI_FC2: launch_nt_<bf16_t, 4, 2, true, true, EPI_FC2, true>(d, s); break;
            default: launch_nt_<bf16_t, 4, 2, true, true, EPI_DG2, true>(d, s); break;
        }
    } else if (tnw == 4) {
        if (want_big_tile(d, epi)) {
            if (epi == EPI_FC2) launch_nt_<bf16_t, 4, 4, true, true, EPI_FC2>(d, s);
            else launch_nt_<bf16_t, 4, 4, true, true, EPI_DG2>(d, s);
        } else {
            GA_LAUNCH_NT(4, 2);
        }
    } else if (tnw == 3) {
        GA_LAUNCH_NT(3, 2);
    } else {
        GA_LAUNCH_NT(2, 2);
    }
#undef GA_LAUNCH_NT
    return ga_check_launch("ga_gemm");
}

extern "C" int ga_wgrad(const ga_wgrad_desc* d, ga_stream_t stream) {
    GA_REQUIRE(d && d->Y && d->X && d->dW, "ga_wgrad: null operand");
    GA_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1 && d->split_m >= 1, "ga_wgrad: bad shape");
    GA_REQUIRE(d->dtype == GA_F32 || d->dtype == GA_BF16, "ga_wgrad: bad dtype %d", d->dtype);
    const int epc = d->dtype == GA_BF16 ? 8 : 4;
    GA_REQUIRE(aligned16(d->Y) && aligned16(d->X), "ga_wgrad: operands must be 16-byte aligned");
    GA_REQUIRE(d->N % epc == 0 && d->ldy % epc == 0 && d->strideY % epc == 0, "ga_wgrad: N/ldy must be multiples of %d",
               epc);
    if (d->x_kind == GA_A_PLAIN) {
        // K (an OUTPUT dim here) may be ragged as long as the X rows are padded to a chunk multiple
        GA_REQUIRE(d->ldx % epc == 0 && d->strideX % epc == 0 && d->ldx >= (d->K + epc - 1) / epc * epc,
                   "ga_wgrad: ldx must be a multiple of %d and cover K rounded up", epc);
    } else if (d->x_kind == GA_A_PATCH2) {
        GA_REQUIRE(d->x_C % epc == 0 && d->K == 4 * d->x_C, "ga_wgrad: PATCH2 needs K==4C");
    } else if (d->x_kind == GA_A_CONV3) {
        GA_REQUIRE(d->x_C % epc == 0 && d->K == 9 * d->x_C, "ga_wgrad: CONV3 needs K==9C");
    } else if (d->x_kind == GA_A_CONV3S2) {
        GA_REQUIRE(d->x_C % epc == 0 && d->K == 9 * d->x_C, "ga_wgrad: CONV3S2 needs K==9C");
    } else if (d->x_kind == GA_A_NEIGH2) {
        GA_REQUIRE(d->x_C % epc == 0 && d->K == 4 * d->x_C, "ga_wgrad: NEIGH2 needs K==4C");
    } else if (d->x_kind == GA_A_STEM4_NCHW) {
        GA_REQUIRE(d->x_C == 3 && d->K == 48, "ga_wgrad: STEM4 needs C=3,K=48");
    } else {
        GA_REQUIRE(false, "ga_wgrad: bad x_kind %d", d->x_kind);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->x_kind == GA_A_CONV3 && ga_conv3_c64_wgrad_try(d, s)) return ga_check_launch("ga_wgrad");   // 64 -> 64 channels: direct form
    if (d->x_kind == GA_A_CONV3S2 && d->x_C == 64 && ga_conv3s2_c64_wgrad_try(d, s)) return ga_check_launch("ga_wgrad");   // ... stride 2
    if (d->x_kind == GA_A_CONV3S2 && d->x_C == 8 && ga_conv0_c8_wgrad_try(d, s)) return ga_check_launch("ga_wgrad");        // 3 (8) -> 64, stride 2
    if (tn2_eligible(d)) {
        static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn2_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, kTn2Smem) == hipSuccess;
        GA_REQUIRE(attr_ok, "ga_wgrad: cannot reserve %d bytes of LDS", kTn2Smem);
        int split;
        const size_t need = tn2_plan(d, &split);
        dim3 grid2(cdiv(d->N, 256) * cdiv(d->K, 256) * split, 1, d->batch), block2(kTn2Threads);
        // partial tiles + one reduce launch when the caller provided the workspace ga_wgrad_workspace() asks for; fp32 atomics
        // into dW otherwise (slower for wide outputs, same result up to summation order)
        float* part = (need && d->workspace && (size_t)d->ws_bytes >= need) ? reinterpret_cast<float*>(d->workspace) : nullptr;
        GA_REQUIRE(!part || aligned16(part), "ga_wgrad: workspace must be 16-byte aligned");
        const long nk = (long)d->N * d->K;
        hipLaunchKernelGGL(gemm_tn2_kernel, grid2, block2, kTn2Smem, s, *d, split, part);
        if (part)
            hipLaunchKernelGGL(tn2_reduce_kernel, dim3((unsigned)std::min<long>(2048, cdiv(nk, 1024)), d->batch), dim3(256),
                               0, s, part, split, nk, d->K, d->alpha, d->dW, d->ldw, d->strideW);
        return ga_check_launch("ga_wgrad");
    }
    dim3 grid(cdiv(d->N, 128) * cdiv(d->K, 128) * d->split_m, 1, d->batch), block(kThreads);
    if (d->dtype == GA_BF16)
        hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, grid, block, 65536, s, *d);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, block, 65536, s, *d);
    return ga_check_launch("ga_wgrad");
}

extern "C" size_t ga_wgrad_workspace(const ga_wgrad_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch < 1) return 0;
    if (d->x_kind == GA_A_CONV3 || (d->x_kind == GA_A_CONV3S2 && (d->x_C == 64 || d->x_C == 8))) return ga_conv3_c64_wgrad_workspace(d);
    if (!tn2_eligible(d)) return 0;
    int split;
    return tn2_plan(d, &split);
}
