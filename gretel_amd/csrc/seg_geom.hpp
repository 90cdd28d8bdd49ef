// seg_geom.hpp -- how the segment-parallel path extension (segwalk.hpp) cuts a window: states, segments, groups, LDS
// sizes.  Plain constexpr / inline functions shared by the host (buffer and grid sizes) and the kernels; the state
// radix R is only known on the device (st->ranked), so the host sizes for both.
#pragma once

#define SEG_THREADS 1024
// diagnostic builds only (-DSEG_STAMPS): s_memtime at the phase boundaries of k_seg's workgroup 0 into st->dbg8
// (-DRWS_STAMPS: k_rwseg's workgroup 100 -- the reweight phases in slots 0..5, its k_seg phases behind them)
#if defined(RWS_STAMPS_ALL)
// (-DRWS_STAMPS_ALL: EVERY workgroup of k_rwseg leaves its stamps, 16 per workgroup, in the (otherwise unused) smin buffer: which
// workgroup a launch waits for, and in which phase -- gh_debug_segment_stamps reads them)
#define RWS_STAMP(i) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) P.smin[(size_t)blockIdx.x * 16 + (i)] = (double)__builtin_amdgcn_s_memtime(); } while (0)
#define SEG_STAMP(i) RWS_STAMP(6 + (i))
#elif defined(RWS_STAMPS)
#define RWS_STAMP(i) do { __builtin_amdgcn_s_waitcnt(0); if (blockIdx.x == 100 && threadIdx.x == 0) P.st->dbg8[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define SEG_STAMP(i) RWS_STAMP(6 + (i))
#elif defined(SEG_STAMPS)
#define RWS_STAMP(i)
#define SEG_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) P.st->dbg8[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RWS_STAMP(i)
#define SEG_STAMP(i)
#endif
// -DRW_STAMPS: the same inside k_rw's workgroup 100 (every stamp waits for the memory operations issued before it)
#ifdef RW_STAMPS
#ifndef RW_STAMP_BLOCK
#define RW_STAMP_BLOCK 100
#endif
#define RW_STAMP(i) do { __builtin_amdgcn_s_waitcnt(0); if (blockIdx.x == RW_STAMP_BLOCK && threadIdx.x == 0) st->dbg8[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RW_STAMP(i)
#endif
#define SEG_MIN_LEN 8          /* shortest segment (positions) */
#define SEG_MAX_L 5            /* 5^5 = 3125 states still fit; beyond that the serial walkers run */
#define SEG_MAX_L_NARROW 6     /* ... and 4^6 = 4096 in a window whose positions all have at most four candidates */
// the five-symbol radix at this lag count?  (R = 4: always)
__host__ __device__ constexpr bool seg_radix_ok(int R, int L) { return R == 4 ? L <= SEG_MAX_L_NARROW : L <= SEG_MAX_L; }

// MIXED radix (class 6, segmix.hpp): a window of the five-symbol layout in which only a few positions offer five candidates.
// A state is the last L picks as candidate RANKS in a mixed-radix number -- the digit of position p has radix R_p = the number
// of candidates there -- so the states entering a target number prod R_p over its L predecessors (1024 where all have four,
// 1280 behind one position with five) instead of 5^L = 3125 everywhere.  The window qualifies when that product stays within
// SEGM_NS at every target (k_classify, dev_state.maxstates); segments and groups are cut like the ranked layout's.
#define SEG_CLS_MIXED 6
#define SEGM_NS 2048           /* state budget per target (2 states per thread) */
#define SEGM_L 5               /* the lag count that has the instantiation: 5^5 against 4^5 is where the radix costs 3x */
#define SEGM_CH 40             /* targets per LDS chunk at most (whole words of 10 picks) */
#define SEGM_NXCAP (24 * 1024) /* Next entries (2 bytes) per chunk; a target has at most SEGM_NS of them: 10 targets always fit */
#define SEGM_MAXITEMS 512      /* (target, block of 64 tasks) work items per chunk */
#define SEGM_ROW 6              /* doubles per staged table row: five columns and one of padding, so that rows start 16-byte aligned */

template <int R> struct seg_radix;
template <> struct seg_radix<4> { typedef uint8_t next_t;  static constexpr int BITS = 2, DPW = 16; };   // 4 picks of 2 bits per entry
template <> struct seg_radix<5> { typedef uint16_t next_t; static constexpr int BITS = 3, DPW = 10; };   // 5 picks of 3 bits
template <> struct seg_radix<6> { typedef uint16_t next_t; static constexpr int BITS = 3, DPW = 10; };   // mixed: up to 5 picks of 3 bits
// (DPW: picks per 32-bit word of hist)
__host__ __device__ constexpr int seg_dpw(int R) { return R == 4 ? 16 : 10; }      // (mixed: 10)

__host__ __device__ constexpr int seg_ipow(int b, int e) { int r = 1; for (int i = 0; i < e; i++) r *= b; return r; }
// states per target the buffers of a class are laid out for
__host__ __device__ constexpr int seg_ns(int R, int L) { return R == SEG_CLS_MIXED ? SEGM_NS : seg_ipow(R, L); }

// positions per LDS chunk of k_seg: the slice of G ((c + L - 1) sources x L lags x R x R doubles) and the chunk's
// Next tables (c x R^(L-1) entries) within 96 KB, at most 64
__host__ __device__ constexpr int seg_chunk(int R, int L)
{
    if (R == SEG_CLS_MIXED) return SEGM_CH;
    const int NI = seg_ipow(R, L - 1), sz = R == 4 ? 1 : 2;
    int c = 64;
    while (c > 8 && ((c + L - 1) * L * R * R * 8 + c * NI * sz + c * R * 8) > 96 * 1024) c -= 8;
    return c / seg_dpw(R) * seg_dpw(R);        // whole words of hist per chunk
}

struct seg_geom {
    int NS, NI;         // states, entries per position (= NS / R: one entry holds the picks of all R oldest digits)
    int seglen, S;      // positions per segment, segments
    int G1, G2;         // groups, segments per group
    int NW;             // 32-bit words of hist per (segment, entry state)
};

// the same on host and device; R is only known on the device (st->ranked), the host sizes for both
__host__ __device__ inline seg_geom seg_geometry(int N, int L, int R)
{
    seg_geom g;
    g.NS = seg_ns(R, L);
    g.NI = R == SEG_CLS_MIXED ? g.NS : g.NS / R;
    int g2 = (g.NS > 3125 ? 65536 : 32768) / g.NS;      // one group's maps (G2 x NS x 2 bytes) within 64 KB of LDS (128 KB for 4^6 states)
    if (g2 > 16) g2 = 16;
    if (g2 < 1) g2 = 1;
    // groups: k_emit keeps the maps of the groups in front of it in LDS ((G1 - 1) x NS x 2 bytes + one prefix map <= 156 KB)
    const int g1max = (g.NS > 2048 && g.NS <= 3125) ? 25 : 16;
    const int smax = g1max * g2;
    int len = (N + smax - 1) / smax;
    if (len < SEG_MIN_LEN) len = SEG_MIN_LEN;
    g.seglen = len;
    g.S = (N + len - 1) / len;
    g.G2 = g2;
    g.G1 = (g.S + g2 - 1) / g2;
    // whole chunks except the last, each a whole number of words
    g.NW = (len / seg_chunk(R, L)) * (seg_chunk(R, L) / seg_dpw(R)) + (len % seg_chunk(R, L) + seg_dpw(R) - 1) / seg_dpw(R);
    return g;
}

// the slice of G and the Next tables (rounded up to 8 bytes: the marginal table of the chunk follows)
__host__ __device__ constexpr size_t seg_lds_bytes(int R, int L)
{
    // mixed: the ranked slice (five rows x five columns per source and lag), the Next entries, the per-target tables (segmix.hpp)
    if (R == SEG_CLS_MIXED) return (size_t)(SEGM_CH + L - 1) * L * 5 * SEGM_ROW * 8 + (size_t)SEGM_NXCAP * 2 + (size_t)SEGM_CH * 64 + 1024 + 64;
    return ((size_t)(seg_chunk(R, L) + L - 1) * L * R * R * 8 + (size_t)seg_chunk(R, L) * seg_ipow(R, L - 1) * (R == 4 ? 1 : 2) + 7) & ~(size_t)7;
}
__host__ __device__ constexpr size_t seg_lds_total(int R, int L) { return seg_lds_bytes(R, L) + (R == SEG_CLS_MIXED ? 0 : (size_t)seg_chunk(R, L) * R * 8); }
__host__ __device__ inline size_t scan_lds_bytes(int N, int L, int R)
{
    const seg_geom g = seg_geometry(N, L, R);
    return (size_t)g.G2 * g.NS * 2;
}
__host__ __device__ inline size_t emit_lds_bytes(int N, int L, int R)
{
    const seg_geom g = seg_geometry(N, L, R);
    return (size_t)g.G1 * g.NS * 2;             // the group maps in front (<= G1 - 1) and the segment's prefix map
}


// k_emit_small: the group maps in front (<= G1 x NS) and every segment map (S x NS), 2 bytes each
__host__ __device__ inline size_t emit_small_lds_bytes(int N, int L, int R)
{
    const seg_geom g = seg_geometry(N, L, R);
    return ((size_t)g.G1 + (size_t)g.S) * g.NS * 2 + 16;
}
