// race_kernel_reg.hip.h -- register-resident whole-race kernel, field size N fixed at compile time.
//
// Same algorithm and same results as race_kernel (race_kernel.hip.h), laid out for the
// CDNA4 register file instead of LDS: one lane = one simulation, and the N cars of that
// simulation live in VGPR arrays indexed by TIME RANK over all cars
//
//     cum[r]  f64   cumulative time of the car at rank r      (reference CarState.cumulative_time)
//     pk[r]   u32   grid slot | tyre age | driver | compound | dirty | drs | dnf | dry compounds used
//
// Every loop over cars is fully unrolled, so all register indices are compile-time
// constants; "the car ahead" is simply rank r-1.  Ordering work is a fixed compare-exchange
// network (93 comparators for N = 20: SortNetwork below) on (cum, pk) after the lap
// times have been added, and one forward + one backward bubble pass after an overtake pass
// (which only moves a few cars by 0.1-0.3 s).  Ties compare pk as an integer: the grid slot sits
// in its top bits, which is Python's stable-sort order for the reference's grid-ordered lists.
//
// Occupancy is what this kernel is short of (one wave issues an instruction every ~5 cycles whatever
// it is; a SIMD needs 3+ waves to fill its issue slots), so the per-lane footprint is kept to what three
// waves per SIMD admit at N = 20: <= 168 VGPRs and 192 B of LDS per lane
//     LAST [N][B]  f64   last lap time per DRIVER                 (reference CarState.last_lap_time)
//     W    [8][B]  u32   draw words of the overtake attempts of one pass (8 at a time)
// with row stride = the block size B (bank = f(lane) only: conflict-free gathers by driver index), plus the
// block-shared tables: inverse-normal cubic, per-driver {variance, base pace}, {base pace, degradation} x 2^31,
// per-(compound, driver) {degradation x factor, pit threshold}, per-compound pace delta, the pit rule, the
// n x n histogram (u32) and the transposed grid matrix.
//
// Random draws of a lap are addressed by the car's PLACE in the field order (its register index): the Philox block of
// places 4j .. 4j+3 is computed right where the four cars' laps are, and never touches LDS.
#pragma once
#include "race_common.hip.h"
#include "race_isa.hip.h"
#include "sort_networks.h"

// Diagnostic builds only (tools/ablate.sh): bit k set = run section k twice (each section is
// idempotent, results unchanged) so its cost shows up as a time difference.  0 in the product build.
#ifndef MCGP_DUP
#define MCGP_DUP 0
#endif
// Diagnostic: bit k set = leave section k out (results become wrong; timing only).
#ifndef MCGP_SKIP
#define MCGP_SKIP 0
#endif

#include <utility>

// Statistics hook of the host debugging build (tools/emu): nothing in the product build.
#ifndef MCGP_STAT
#define MCGP_STAT(what, value)
#endif
#ifndef MCGP_TRACE_PASS
#define MCGP_TRACE_PASS(sim, lap, pass, n_cand)
#endif
// Race-interrupting events handled by the wave with lane = car (1) or by every lane for itself (0: the same
// results; kept for A/B runs and for the host debugging build, whose threads run one at a time).
// Diagnostic (host build, tests): 1 = _sample_grid takes its exact, dividing path for every draw.
#ifndef MCGP_GRID_EXACT
#define MCGP_GRID_EXACT 0
#endif
#ifndef MCGP_COOPERATIVE_EVENTS
#define MCGP_COOPERATIVE_EVENTS 1
#endif
// Diagnostic (host build, tests): 1 = the reference-width build takes its exact 53-bit path for every event draw and every
// overtake pass (the path a wave otherwise takes for one draw in 2^32: a word equal to the leading word of its threshold),
// and reads the table row of every deviate from device memory (the path of a row the block does not hold in LDS).
#ifndef MCGP_WIDE_EXACT
#define MCGP_WIDE_EXACT 0
#endif
// Diagnostic: a draw word counts as "equal to the leading word of its threshold" when the two agree in all but their
// lowest MCGP_WIDE_TIE_SHIFT bits (0 in the product: equality).  A wider net sends a share of the waves down the exact
// path with only SOME of their lanes needing it -- the mix a real tie produces, at a rate a test can see.
#ifndef MCGP_WIDE_TIE_SHIFT
#define MCGP_WIDE_TIE_SHIFT 0
#endif

namespace mcgp {

// pk word of the register kernel.  Field positions are chosen so that LDS addresses fall out of ONE mask wherever a
// table is indexed three times a lap (the overtake passes) and of a shift + mask elsewhere:  pk & 0x1F0 = 16 driver;
// pk & 0x3F0 = 16 (32 dnf + driver);  pk & 8 = 8 drs;  (pk & 0x1C00) >> 1 = 512 compound;  (pk >> 6) & 0x70 = 16 compound;
// and (pk & k3AgeMask) + (1 << 16) = (tyre age + 1) << 16, which compares directly with the pit-stop word.
constexpr uint32_t k3UsedMask = 7u;            // [0..2] dry compounds used so far (SOFT, MEDIUM, HARD)
constexpr uint32_t k3Drs = 1u << 3;            // (pk & 8) = 8 drs: byte offset into the {0.0, drs_delta} table
constexpr int k3IdShift = 4;                   // [4..8] driver index: (pk & k3IdMask) = 16 driver
constexpr uint32_t k3IdMask = 31u << k3IdShift;
constexpr uint32_t k3Dnf = 1u << 9;            // directly above the driver index: pk & k3SlotMask = 16 (32 dnf + driver)
constexpr uint32_t k3SlotMask = k3IdMask | k3Dnf;
constexpr int k3CompShift = 10;                // [10..12] tyre compound
constexpr uint32_t k3CompMask = 7u << k3CompShift;
constexpr uint32_t k3Dirty = 1u << 13;         // 0 < time_behind_leader < dirty_air_threshold   ([14..15] unused)
constexpr int k3AgeShift = 16;                 // [16..26] tyre age; lap of retirement once dnf is set
constexpr uint32_t k3AgeMask = 0x7FFu << k3AgeShift;
constexpr int k3GposShift = 27;                // [27..31] grid slot (most significant: tie-break)
__host__ __device__ constexpr uint32_t pk_driver16(uint32_t p) { return p & k3IdMask; }                  // 16 x driver
__host__ __device__ constexpr uint32_t pk_slot16(uint32_t p) { return p & k3SlotMask; }                  // 16 x (32 dnf + driver)
__host__ __device__ constexpr uint32_t pk_comp512(uint32_t p) { return (p & k3CompMask) >> (k3CompShift - 9); }   // 512 x compound
__host__ __device__ constexpr uint32_t pk_comp16(uint32_t p) { return (p >> (k3CompShift - 4)) & 0x70u; }        // 16 x compound
constexpr double kAgeFieldUnit = 1.0 / 65536.0;   // tables that multiply (pk & k3AgeMask) carry this factor

// ---- launch geometry and LDS map, fixed per field size ----
constexpr size_t kLdsPerCu = 160 * 1024;       // gfx950
__host__ __device__ constexpr size_t align16(size_t x) { return (x + 15) / 16 * 16; }
// Rows of the per-lane W plane: the draw words of 8 overtake attempts (two Philox blocks).  A pass with more
// attempts than that in some lane -- 0.4 % of wave-passes on the 20-car benchmark fields -- goes through the
// plane 8 attempts at a time.  The same 32 bytes per lane hold the sampled grid (one byte per slot) before lap 1.
constexpr int kWordRows = 8;
__host__ __device__ constexpr size_t per_thread_lds_bytes_reg(int n) { return (size_t)kWordRows * 4 + (size_t)n * 8; }
// block-shared tables: inverse-normal rows; per-driver {var, base}; {base, deg} x 2^31 with a second copy of
// NaNs 32 entries further on, which the pk word of a RETIRED car indexes: its pace is NaN, so both pairs it
// belongs to fail every overtake test without a flag test; per-(compound, driver) {eff f64, pit word u32, DNF
// threshold u32} (compound stride 32 entries); per-compound {delta, pad}; {0.0, drs_delta} and the same x 2^31;
// n x n histogram (u32); transposed grid matrix (f64).  The driver index varies fastest in every table (16-byte
// entries): lanes holding different drivers hit different banks, the same driver broadcasts.  Tables end at the
// last driver (not at 32): the LDS saved is what lets the N = 20 block hold three waves per SIMD.
constexpr int kNumCompounds = 5;
__host__ __device__ constexpr size_t shared_lds_bytes_reg(int n)
{
    return (size_t)kNormalRows * 16 + (size_t)n * 16 + (size_t)(kMaxCars + n) * 16 +
           (size_t)((kNumCompounds - 1) * kMaxCars + n) * 16 + kCompStride * 16 + 64 + 128 + align16((size_t)n * n * 4) +
           align16((size_t)n * n * 8);      // (16-byte multiple: the per-lane planes that follow are accessed 16 bytes at a time)
}
// Waves per SIMD the kernel is compiled for (__launch_bounds__: register budget 512 / this), chosen by measurement
// for every field size (tools/sweep_waves.sh, one box, 4x10^6 simulations, profiles/r3_sweep_waves.txt): e.g. N = 10
// 13.2 ms at 4 waves, 13.6 at 3, 18.1 at 5; N = 20 32.3 at 3, 51.6 at 4 (spills); N = 22 37.0 at 3, 39.9 at 2; N = 23
// 42.3 at 2, 51.1 at 3 (the LDS rows of 23 cars leave room for 10 waves per CU only).  N = 4 is the one irregular
// entry: above 4 waves the compiler moves its arrays to scratch (288 B per lane) and it runs 2-4x slower.
#ifdef MCGP_MIN_WAVES
__host__ __device__ constexpr int reg_min_waves(int) { return MCGP_MIN_WAVES; }
#else
__host__ __device__ constexpr int reg_min_waves(int n) { return n <= 6 ? (n == 4 ? 4 : 6) : n <= 11 ? 4 : n <= 22 ? 3 : 2; }
#endif
// Waves per block: the (waves per block, blocks per CU) pair that keeps the most waves resident within the LDS
// budget and the kernel's waves per SIMD; among equals a multiple of 4 waves per block (a block's waves go round the
// 4 SIMDs: two blocks of 6 load them 4, 4, 2, 2), then the block nearest to 8 waves (1024-thread blocks make the
// register allocator spill more -- N = 10: 164 B per lane against 48 --, many small ones copy the tables more often).
#ifndef MCGP_MAX_BLOCK_WAVES
#define MCGP_MAX_BLOCK_WAVES 16
#endif
constexpr size_t kLdsReserve = 256;             // kept free: the block must fit beside what the runtime itself may take
__host__ __device__ constexpr int block_shape_rank(int w) { return (w % 4 == 0 ? 0 : 100) + (w <= 8 ? 2 * (8 - w) : w - 8); }
__host__ __device__ constexpr int reg_block_waves(int n)
{
    const int cap = 4 * reg_min_waves(n);
    int best = 0, waves = 1;
    for (int w = 1; w <= MCGP_MAX_BLOCK_WAVES; ++w) {
        int b = (int)((kLdsPerCu - kLdsReserve) / (shared_lds_bytes_reg(n) + (size_t)w * 64 * per_thread_lds_bytes_reg(n)));
        if (b * w > cap) b = cap / w;
        if (b < 1) continue;
        if (b * w > best || (b * w == best && block_shape_rank(w) < block_shape_rank(waves))) { best = b * w; waves = w; }
    }
    return waves;
}

// device memory the kernel wants for the lanes' retirement lists: (n + 1) words per lane of the launch
__host__ __device__ constexpr size_t reg_retire_ws_bytes(int n, size_t lanes) { return (size_t)(n + 1) * lanes * 4; }

// (WAVES: the default is the block shape measured best for the field size; the reference-width build runs at 2 waves per
//  SIMD -- 256 registers instead of 168, which its binary64 deviates want -- in blocks of 8 waves)
// The reference-width build at 3 waves per SIMD (168 registers, as the default build at up to 22 cars): its lap step is
// ordered so that a batch's four deviates are FINISHED before the batch's slots issue their LDS gathers -- the sixteen
// registers of a table row and the sixty of the gathered values are then never alive together --, table rows are fetched
// one at a time, a batch whose rows the block does not hold reads them from device memory one deviate at a time, and the
// exact 53-bit overtake pass is out of line (wide_pass_exact_fn), so that its registers are not the race loop's to pay for.  Same-box A/B against blocks of 8 waves at 2 per SIMD with the default ordering (profiles/r5_ab.txt): 10 cars -18 %,
// 16 -12 %, 18 -9.5 %, 19 -7 %, 20 -6.5 % (S78 -4 %), 21 +0.4 %, 22 -2.8 %.  MCGP_WIDE_LEAN=0 (diagnostic builds) keeps every
// field size at 2 waves per SIMD, gathers first, two rows at a time.
#ifndef MCGP_WIDE_LEAN
#define MCGP_WIDE_LEAN 1
#endif
// MCGP_WIDE_BLOCK_WAVES / MCGP_WIDE_MIN_WAVES (diagnostic builds): a fixed block shape / register budget for every field size
// instead of the choice of wide_block_waves() / wide_min_waves() below.
#ifdef MCGP_WIDE_BLOCK_WAVES
constexpr int kWideBlockWaves = MCGP_WIDE_BLOCK_WAVES;
#else
constexpr int kWideBlockWaves = 0;             // chosen per field size
#endif
// The reference-width build keeps the binary64 inverse-normal table (normal53_table.h: 784 rows of 8 coefficients, 64 B
// each) in LDS, behind the per-lane planes, instead of the 7 KB binary32 table it has no use for: as many of the table's
// LAST rows -- the cells of the largest tail indices -- as fit beside the block's other data, in whole octaves of 16 rows,
// and at most kWideLdsRowsMax of them: every octave left out halves the share of deviates that go to the copy in device
// memory instead (normal53_rows, race_common.hip.h), which at 2^-32 per draw (a few wave-batches per 10^7 races) costs
// nothing and keeps that path exercised.
constexpr int kWideLdsRowsMax = 32 * 16;
// A row takes 80 bytes of LDS, not 64: its four 16-byte pieces are fetched by four ds_read_b128, each lane its own row,
// and a 16-lane group of that instruction is conflict-free only if its lanes hit 16 different 16-byte slots of the 256-byte
// bank row.  At a stride of 64 bytes piece k of ANY row lies in one of 4 slots (4 r + k mod 16): 16 lanes in 4 slots,
// 6 LDS cycles per group on average and 48 % of the kernel's LDS cycles spent on conflicts (profiles/r5_counters.json, first
// version); at 80 bytes (5 r + k mod 16) every slot is reachable and the average is 3.
constexpr int kNorm53LdsStride = 80;
__host__ __device__ constexpr size_t wide_fixed_lds_bytes(int n, int waves)
{
    return shared_lds_bytes_reg(n) - (size_t)kNormalRows * 16 + (size_t)waves * 64 * per_thread_lds_bytes_reg(n);
}
__host__ __device__ constexpr int wide_lds_rows(int n, int waves)
{
    const size_t fixed = wide_fixed_lds_bytes(n, waves);
    const size_t room = kLdsPerCu - kLdsReserve > fixed ? kLdsPerCu - kLdsReserve - fixed : 0;
    int rows = (int)(room / (size_t)kNorm53LdsStride) / 16 * 16;
    return rows > kWideLdsRowsMax ? kWideLdsRowsMax : rows;
}
__host__ __device__ constexpr size_t wide_lds_bytes(int n, int waves)
{
    return wide_fixed_lds_bytes(n, waves) + (size_t)wide_lds_rows(n, waves) * kNorm53LdsStride;
}
// ... and, behind the table rows, the 64-bit survival thresholds of the retirement draw (S_k per lap and driver, the same
// for every race of the launch: reg_load_tables) when the race is short enough for them to fit -- a 20-car block has room
// for about 150 laps; longer races work the chain out per wave on the scalar unit, as the default build does in 32 bits.
__host__ __device__ constexpr size_t wide_chain_bytes(int n, int total_laps)
{
    return total_laps >= 2 ? (size_t)(total_laps - 1) * (size_t)((n + 3) & ~3) * 8 : 0;
}
__host__ __device__ constexpr size_t wide_launch_lds_bytes(int n, int waves, int total_laps)
{
    const size_t base = wide_lds_bytes(n, waves), chain = wide_chain_bytes(n, total_laps);
    return base + chain + kLdsReserve <= kLdsPerCu ? base + chain : base;
}
template <int N, int WAVES = reg_block_waves(N), bool WIDE = false>
struct RegGeo {
    static constexpr int kWaves = WAVES;
    static constexpr bool kWide = WIDE;
    static constexpr int B = 64 * kWaves;                         // threads per block
    // block-shared tables first: their bases (and the row bases below) fit the 16-bit immediate offset of a DS
    // instruction, so an address is just the bit field taken from pk
    static constexpr uint32_t oDrvA = 0;                          // {var, base} x N drivers, 16 B each   (lap step)
    static constexpr uint32_t oDrvB = oDrvA + N * 16;             // {base, deg} 2^31 x N drivers; at +32 entries {NaN, NaN} x N  (overtake pace)
    static constexpr uint32_t oIc = oDrvB + (kMaxCars + N) * 16;  // [compound][driver] {eff f64, pit word u32, DNF threshold u32}
    static constexpr uint32_t oComp = oIc + ((kNumCompounds - 1) * kMaxCars + N) * 16;   // {delta f64, pad} x 8, 16 B each
    static constexpr uint32_t oDrs = oComp + kCompStride * 16;    // {0.0, drs_delta}                          (lap time)
    static constexpr uint32_t oPit = oDrs + 16;                   // {0.0, pit_loss}                           (the stop's time, by flag)
    static constexpr uint32_t oDrsB = oDrs + 32;                  // {0.0, drs_delta 2^31}                     (overtake pace)
    static constexpr uint32_t oLut = oDrsB + 32;                  // pit rule: u32 [4 regimes][8 used-sets]
    static constexpr uint32_t oNorm = oLut + 128;                 // float4[kNormalRows]; addressed from kNormalRowBias rows before it (not WIDE)
    static constexpr uint32_t oHist = oNorm + (WIDE ? 0 : kNormalRows * 16);   // u32[N x N]
    static_assert(WIDE || oNorm >= kNormalRowBias * 16, "the inverse-normal rows are addressed through a base 48 rows before the table");
    static constexpr uint32_t oGrid = oHist + (uint32_t)align16((size_t)N * N * 4);   // f64 [slot][driver]
    static constexpr uint32_t oW = oGrid + (uint32_t)align16((size_t)N * N * 8);   // [kWordRows][B] u32, 16-byte aligned (odd N: padded)
    static constexpr uint32_t oLast = oW + (uint32_t)kWordRows * B * 4;      // [N][B] f64
    // WIDE: rows kNorm53First .. 783 of the binary64 inverse-normal table, 64 B each, behind the per-lane planes
    static constexpr uint32_t oNorm53 = oLast + (uint32_t)N * B * 8;
    static constexpr int kNorm53Rows = WIDE ? wide_lds_rows(N, WAVES) : 0;
    static constexpr int kNorm53First = kNormal53Rows - kNorm53Rows;
    static constexpr uint32_t kBytes = oNorm53 + (uint32_t)kNorm53Rows * (uint32_t)kNorm53LdsStride;
    // WIDE: [lap - 2][driver, padded to a multiple of 4] u64 survival thresholds of the retirement draw, if they fit
    static constexpr uint32_t oChain = kBytes;
    static constexpr uint32_t kChainStride = (uint32_t)((N + 3) & ~3) * 8u;
    __host__ __device__ static constexpr bool chain_fits(int total_laps)
    {
        return WIDE && total_laps >= 2 && kBytes + wide_chain_bytes(N, total_laps) + kLdsReserve <= kLdsPerCu;
    }
    static_assert(oW == shared_lds_bytes_reg(N) - (WIDE ? (size_t)kNormalRows * 16 : 0) && oLast % 8 == 0, "LDS map");
    static_assert(oW % 16 == 0 && oNorm53 % 16 == 0, "the event handler parks fields with 16-byte LDS accesses (ds_write_b128 / ds_read_b128); table rows are read 16 bytes at a time");
    static_assert(kBytes == (WIDE ? wide_lds_bytes(N, WAVES) : per_thread_lds_bytes_reg(N) * B + shared_lds_bytes_reg(N)), "LDS map");
    static_assert(kBytes + kLdsReserve <= kLdsPerCu, "block does not fit LDS");
    static_assert(oLast < 65536, "row bases must fit the DS immediate offset");
};
// Waves per block of the reference-width build (one block per CU).  The lean ordering of its lap step (MCGP_WIDE_LEAN) fits
// the 168 registers of 3 waves per SIMD, so the block is as large as the LDS allows while still holding kWideMinLdsRows
// rows of the table -- below that a batch of deviates leaves the LDS rows too often: 12 or 11 waves up to 20 cars, 10 at 21
// and 22 --; without room for more than 8 waves (23 cars and more), or without the lean ordering, 8 waves at 2 per SIMD
// (256 registers), fewer for the largest fields (31 cars and more).
constexpr int kWideMinLdsRows = 14 * 16;
__host__ __device__ constexpr int wide_block_waves(int n)
{
    if (kWideBlockWaves == 0 && MCGP_WIDE_LEAN && n <= 22) {        // (23 cars and more: the default build itself runs at 2 waves per SIMD)
        for (int w = 12; w > 8; --w)
            if (wide_fixed_lds_bytes(n, w) + kLdsReserve <= kLdsPerCu && wide_lds_rows(n, w) >= kWideMinLdsRows) return w;
    }
    int w = kWideBlockWaves ? kWideBlockWaves : 8;
    while (w > 1 && wide_fixed_lds_bytes(n, w) + kLdsReserve > kLdsPerCu) --w;
    return w;
}
// waves per SIMD it is compiled for (register budget 512 / this)
__host__ __device__ constexpr int wide_min_waves(int n)
{
#ifdef MCGP_WIDE_MIN_WAVES
    (void)n;
    return MCGP_WIDE_MIN_WAVES;
#else
    return wide_block_waves(n) > 8 ? 3 : 2;
#endif
}
template <int N>
using WideGeo = RegGeo<N, wide_block_waves(N), true>;

// The sorting network of the full sort: Knuth, TAOCP 5.2.2 Algorithm M (merge exchange) for any N, or, for the sizes
// listed in sort_networks.h, two size-optimal half sorters and Batcher's odd-even merge (20 cars: 93 comparators instead
// of 97; a few of them put their minimum on the HIGHER slot, which cmpx_time does not mind).  Comparators of one step /
// layer touch disjoint slots; they are grouped two by two (ga / gn) so that two go into one instruction block.
template <int N>
struct SortNetwork {
    int a[N * 8];
    int b[N * 8];
    int step[N * 8];
    int n;
    int ga[N * 8];          // group -> index of its first comparator
    int gn[N * 8];          // group -> 1 or 2 comparators
    int n_groups;
    constexpr SortNetwork() : a{}, b{}, step{}, n(0), ga{}, gn{}, n_groups(0)
    {
        const SortNetworkTable *table = nullptr;
        for (const SortNetworkTable &t : kSortNetworks)
            if (t.n == N) table = &t;
        if (table) {
            // two size-optimal half sorters + Batcher's odd-even merge (sort_networks.h, tools/gen_sort_networks.py)
            for (int i = 0; i < table->size; ++i) {
                a[n] = table->pairs[i][0];
                b[n] = table->pairs[i][1];
                step[n] = table->pairs[i][2];
                ++n;
            }
        } else {
            // merge exchange (Knuth, TAOCP 5.3.4, Algorithm M)
            int t = 0, s = 0;
            while ((1 << t) < N) ++t;
            for (int p = t > 0 ? 1 << (t - 1) : 0; p > 0; p >>= 1) {
                int q = 1 << (t - 1), r = 0, d = p;
                while (true) {
                    for (int i = 0; i < N - d; ++i)
                        if ((i & p) == r) { a[n] = i; b[n] = i + d; step[n] = s; ++n; }
                    ++s;
                    if (q == p) break;
                    d = q - p;
                    q >>= 1;
                    r = p;
                }
            }
        }
        for (int i = 0; i < n;) {
            const int len = (i + 1 < n && step[i + 1] == step[i]) ? 2 : 1;
            ga[n_groups] = i;
            gn[n_groups] = len;
            ++n_groups;
            i += len;
        }
    }
};

// compare-exchange on (cum, pk): after it, slot A sorts before slot B.
__device__ __forceinline__ bool cmpx(double &ca, uint32_t &pa, double &cb, uint32_t &pb)
{
    const bool sw = (ca > cb) || (ca == cb && pa > pb);
    double c0, c1;
    minmax_f64(ca, cb, c0, c1);
    const uint32_t p0 = sw ? pb : pa, p1 = sw ? pa : pb;
    ca = c0; cb = c1; pa = p0; pb = p1;
    return sw;
}

// cmpx_time(ca, pa, cb, pb): compare-exchange on the time alone (ties left as they are), race_isa.hip.h.

template <int N, int G>
__device__ __forceinline__ void network_group(double (&cum)[N], uint32_t (&pk)[N])
{
    constexpr SortNetwork<N> net{};
    constexpr int i = net.ga[G];
    if constexpr (net.gn[G] == 2)
        cmpx_time2(cum[net.a[i]], pk[net.a[i]], cum[net.b[i]], pk[net.b[i]],
                   cum[net.a[i + 1]], pk[net.a[i + 1]], cum[net.b[i + 1]], pk[net.b[i + 1]]);
    else
        cmpx_time(cum[net.a[i]], pk[net.a[i]], cum[net.b[i]], pk[net.b[i]]);
}

template <int N, size_t... G>
__device__ __forceinline__ void network_sort_impl(double (&cum)[N], uint32_t (&pk)[N], std::index_sequence<G...>)
{
    (network_group<N, (int)G>(cum, pk), ...);
}

// The same network on plain integer keys (classification): a comparator is one v_min_u32 and one v_max_u32.
template <int N, size_t... C>
__device__ __forceinline__ void network_sort_keys_impl(uint32_t (&key)[N], std::index_sequence<C...>)
{
    constexpr SortNetwork<N> net{};
    auto cx = [&](int a, int b) {
        const uint32_t lo = key[a] < key[b] ? key[a] : key[b], hi = key[a] < key[b] ? key[b] : key[a];
        key[a] = lo;
        key[b] = hi;
    };
    (cx(net.a[C], net.b[C]), ...);
}
template <int N>
__device__ __forceinline__ void network_sort_keys(uint32_t (&key)[N])
{
    constexpr SortNetwork<N> net{};
    network_sort_keys_impl<N>(key, std::make_index_sequence<(size_t)net.n>{});
}

// Forward bubble pass from slot I: comparators (I, I+1), (I+1, I+2), .. (N-2, N-1), four to an instruction block.
template <int N, int I>
__device__ __forceinline__ void bubble_forward(double (&c)[N], uint32_t (&p)[N])
{
    constexpr int left = N - 1 - I;
    if constexpr (left >= 4) {
        bubble_fwd4(c[I], p[I], c[I + 1], p[I + 1], c[I + 2], p[I + 2], c[I + 3], p[I + 3], c[I + 4], p[I + 4]);
        bubble_forward<N, I + 4>(c, p);
    } else if constexpr (left == 3) {
        bubble_fwd3(c[I], p[I], c[I + 1], p[I + 1], c[I + 2], p[I + 2], c[I + 3], p[I + 3]);
    } else if constexpr (left == 2) {
        bubble_fwd2(c[I], p[I], c[I + 1], p[I + 1], c[I + 2], p[I + 2]);
    } else if constexpr (left == 1) {
        cmpx_time(c[I], p[I], c[I + 1], p[I + 1]);
    }
}

// Backward bubble pass: comparators (T-1, T), (T-2, T-1), .. (0, 1).
template <int N, int T>
__device__ __forceinline__ void bubble_backward(double (&c)[N], uint32_t (&p)[N])
{
    if constexpr (T >= 4) {
        bubble_bwd4(c[T - 4], p[T - 4], c[T - 3], p[T - 3], c[T - 2], p[T - 2], c[T - 1], p[T - 1], c[T], p[T]);
        bubble_backward<N, T - 4>(c, p);
    } else if constexpr (T == 3) {
        bubble_bwd3(c[0], p[0], c[1], p[1], c[2], p[2], c[3], p[3]);
    } else if constexpr (T == 2) {
        bubble_bwd2(c[0], p[0], c[1], p[1], c[2], p[2]);
    } else if constexpr (T == 1) {
        cmpx_time(c[0], p[0], c[1], p[1]);
    }
}

// After a round that ends with the odd pairs (1,2), (3,4), .. those pairs are in order by
// construction; the field is in order iff the even pairs (0,1), (2,3), .. still are.
template <int N>
__device__ __forceinline__ bool even_pairs_in_order(const double (&cum)[N], const uint32_t (&pk)[N])
{
    bool bad = false;
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2)
        bad |= (cum[i] > cum[i + 1]) || (cum[i] == cum[i + 1] && pk[i] > pk[i + 1]);
    return !bad;
}

// Odd-even transposition rounds with the full (time, grid slot) comparator until the field is in
// order: the general re-sort, correct for any input.  Rarely reached (equal times, or a disturbance
// the cheaper passes below did not undo).
template <int N>
__device__ __forceinline__ void transposition_sort(double (&cum)[N], uint32_t (&pk)[N])
{
    do {
#pragma unroll
        for (int i = 0; i + 1 < N; i += 2) (void)cmpx(cum[i], pk[i], cum[i + 1], pk[i + 1]);
#pragma unroll
        for (int i = 1; i + 1 < N; i += 2) (void)cmpx(cum[i], pk[i], cum[i + 1], pk[i + 1]);
    } while (!even_pairs_in_order<N>(cum, pk));
}

// Is the field in (time, grid slot) order?  One pass over the adjacent pairs.
template <int N>
__device__ __forceinline__ bool in_order(const double (&cum)[N], const uint32_t (&pk)[N])
{
    bool bad = false;
#pragma unroll
    for (int i = 0; i + 1 < N; ++i)
        bad |= (cum[i] > cum[i + 1]) || (cum[i] == cum[i + 1] && pk[i] > pk[i + 1]);
    return !bad;
}

// Strictly increasing times: the field is in order whatever the tie-break says (one compare per pair).
// The cheap test; when it fails (an inversion, or two equal times) the exact, tie-aware code takes over.
template <int N>
__device__ __forceinline__ bool strictly_increasing(const double (&cum)[N])
{
    bool bad = false;
#pragma unroll
    for (int i = 0; i + 1 < N; ++i) bad |= !(cum[i] < cum[i + 1]);
    return !bad;
}

// After the time-only network the field is ordered by time; only equal times can still be in the
// wrong (grid) order.
template <int N>
__device__ __forceinline__ bool ties_in_order(const double (&cum)[N], const uint32_t (&pk)[N])
{
    bool bad = false;
#pragma unroll
    for (int i = 0; i + 1 < N; ++i) bad |= (cum[i] == cum[i + 1]) && pk[i] > pk[i + 1];
    return !bad;
}

// Full sort by (cumulative_time, grid slot): Python's stable sorted() of the reference (:506 etc.).
// The network orders by time; equal times are then put in grid order by the transposition rounds, which
// almost never have anything to do (the structural ties of the reference, lap-1 retirements at 0.0, are
// kept apart by the representation chosen in lap 1, see there).
template <int N>
__device__ __forceinline__ bool network_sort(double (&cum)[N], uint32_t (&pk)[N])
{
    constexpr SortNetwork<N> net{};
    network_sort_impl<N>(cum, pk, std::make_index_sequence<(size_t)net.n_groups>{});
    const bool strict = strictly_increasing<N>(cum);
    MCGP_STAT(13, !strict);
    if (__builtin_expect(!strict, 0)) {
        if (!ties_in_order<N>(cum, pk)) transposition_sort<N>(cum, pk);
    }
    return strict;                                 // no two equal times in the field (update_positions_reg)
}

// Re-sort after an overtake pass.  The pass moved a few cars: the overtaken one back by 0.2 s, the
// overtaking one to 0.1 s ahead of where its rival was.  One forward bubble pass carries every car
// that fell back to its place, one backward pass every car that moved up; time-only comparators
// (5 instructions instead of 9) never swap equal times, so ties keep their (correct) relative order.
// The result is then CHECKED with the full (time, grid slot) order and anything left -- cars crossing
// each other, a new exact tie in the wrong grid order -- goes to the general re-sort.
template <int N>
__device__ __forceinline__ bool resort_after_overtakes(double (&cum)[N], uint32_t (&pk)[N])
{
    bubble_forward<N, 0>(cum, pk);                 // (0,1), (1,2), .. (N-2,N-1)
    bubble_backward<N, N - 2>(cum, pk);            // (N-3,N-2), .. (0,1): the last slot already holds the maximum
    const bool strict = strictly_increasing<N>(cum);
    MCGP_STAT(14, !strict);
    if (__builtin_expect(!strict, 0)) {
        if (!in_order<N>(cum, pk)) transposition_sort<N>(cum, pk);
    }
    return strict;
}

// _update_positions, reference :538-560: gap to the leader (kept as the dirty-air flag, :209-212) and the DRS flag
// of every running car, in rank order; retired cars are skipped.
// DISTINCT: the caller knows that no two times of the field are equal (the sorts report it; all but a few wave-laps
// in a hundred).  Every car behind the leader then has a gap > 0 and the first half of :552's  0 < gap < threshold
// is the "not the first running car" mask the DRS flag needs anyway: one binary64 compare less per slot.
template <int N, bool DISTINCT = false>
__device__ __forceinline__ void update_positions_reg(const double (&cum)[N], uint32_t (&pk)[N],
                                                     bool drs_allowed, double dirty_thr)
{
    bool first = true;
    double leader = 0.0, prev = 0.0;
    // one straight pass, merged by selects (a few lanes hold retired cars; the wave would pay for the branches).
    // A RETIRED car's two flags are written like everybody's, with whatever the arithmetic gives: nothing reads them
    // again (the lap step and the event handler test the dnf bit first, a retired car's overtake pace is NaN whatever
    // its DRS bit adds, the classification does not look at flags), and not protecting them saves a select per slot.
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t p = pk[i];
        const bool act = !(p & k3Dnf);
        const double t = cum[i];
        leader = (act && first) ? t : leader;
        const double tbl = t - leader;                                               // :551
        const bool dirty = (DISTINCT ? !first : tbl > 0) && tbl < dirty_thr;
        const bool drs = !first && drs_allowed && (t - prev) < 1.0;                  // :553-558
        pk[i] = (p & ~(k3Drs | k3Dirty)) | (dirty ? k3Dirty : 0u) | (drs ? k3Drs : 0u);
        prev = act ? t : prev;
        first = first && !act;
    }
}

// Compound fitted at a pit stop (reference :469-490) as a function of the dry compounds already used (3 bits).
// Everything else it depends on is the same for the whole wave in a given lap, and the remaining laps matter only
// through four regimes (> 30, 21..30, 16..20, <= 15: the thresholds of :471-477 and :485), so the rule is tabulated
// once per block: 4 regimes x 8 used-sets, each entry the new compound and used-set bits where pk keeps them.
__host__ __device__ constexpr int pit_regime(int remaining_laps)
{
    return remaining_laps > 30 ? 0 : remaining_laps > 20 ? 1 : remaining_laps > 15 ? 2 : 3;
}
__device__ __forceinline__ uint32_t pit_rule_word(int track, int regime, uint32_t used, uint32_t pop_sh, uint32_t pop_mh)
{
    const int remaining_laps = regime == 0 ? 31 : regime == 1 ? 21 : regime == 2 ? 16 : 0;     // one value per regime
    uint32_t newc = stint_compound(track, remaining_laps);
    // two-compound rule, dry races only (:481-490): exactly one dry compound used so far and the stint rule would
    // fit it again -> take another one
    if (track == 0 && __popc(used) == 1 && ((used >> newc) & 1u)) {
        const uint32_t avail = 7u & ~used;
        const uint32_t popped = avail == 5u ? pop_sh : avail == 6u ? pop_mh : 0u;
        if (remaining_laps > 20) newc = (avail & 2u) ? 1u : popped;
        else newc = (avail & 1u) ? 0u : popped;
    }
    return (newc << k3CompShift) | ((used | (1u << newc)) & k3UsedMask);
}

// The instruction scheduler may not move anything across this point (no instruction is emitted).
#ifndef MCGP_SCHED_FENCE
#define MCGP_SCHED_FENCE() sched_fence()
#endif
#ifndef MCGP_STEP_BATCH
#define MCGP_STEP_BATCH 4          // slots whose LDS gathers (and Philox blocks) are in flight together in the lap step
#endif
#ifndef MCGP_DISTINCT_PATH
#define MCGP_DISTINCT_PATH 1       // update_positions_reg<N, true> for the wave-laps whose fields have no equal times
#endif
// Diagnostic (timing only, wrong results in the rare cases): 1 = the two rare paths of the reference-width build are never taken
#ifndef MCGP_NOCALLS
#define MCGP_NOCALLS 0
#endif
#ifndef MCGP_WIDE_STEP_BATCH
#define MCGP_WIDE_STEP_BATCH 4
#endif
#ifndef MCGP_WIDE_PACE_BATCH
#define MCGP_WIDE_PACE_BATCH 5
#endif
#ifndef MCGP_PACE_BATCH
#define MCGP_PACE_BATCH 5          // slots whose pace gathers are in flight together in an overtake pass (10: same speed, 52 B of spills against 20)
#endif

// A lower bound of every lap time of the problem, lap 1 included (reference :317-332, :301-306), from the inputs alone:
// slowest possible fuel effect, DRS gain, the largest negative noise a deviate can give (|z| < 6.5, 8.5 at reference width), the most
// negative compound delta and degradation x age, the dirty-air penalty if it is negative, the largest start gain.
// The register kernel relies on times that stay clear of zero: with every lap adding at least kRegTimeFloor seconds
// and an overtake pass taking at most 0.1 s per pair off a car's time (19 pairs x 3 passes < 6 s per lap), a running
// car's cumulative time never comes near 0.2 s, so the reference's  max(0.1, ahead - 0.1)  (:528) is  ahead - 0.1.
// A problem that cannot promise this -- lap times of a few seconds, NaN or infinite inputs -- runs on the generic kernel.
constexpr double kRegTimeFloor = 8.0;
__host__ __device__ inline double reg_time_floor(const KParams &kp)
{
    // largest |deviate| a draw can give: Phi^-1(2^-33) = -6.4 from the 32-bit table, Phi^-1(2^-54) = -8.3 at reference width
    const double z_max = kp.wide ? 8.5 : 6.5;
    double floor_ = __builtin_inf();
    double cdelta_min = __builtin_inf();
    for (int c = 0; c < 5; ++c) cdelta_min = kp.comp_delta[c] < cdelta_min ? kp.comp_delta[c] : cdelta_min;
    for (int d = 0; d < kp.n; ++d) {
        double eff_min = 0.0;
        for (int c = 0; c < 5; ++c) {
            const double e = kp.comp_deg[c] * kp.factor[d];
            eff_min = e < eff_min ? e : eff_min;
        }
        const double var = kp.variance[d] < 0 ? -kp.variance[d] : kp.variance[d];
        const double f = kp.base_pace[d] - var * z_max + eff_min * 2047.0;
        floor_ = f < floor_ ? f : floor_;
        if (!(f == f)) return f;                                   // NaN
    }
    floor_ = floor_ + cdelta_min - 110.0 * 0.03 - (kp.drs_delta > 0 ? kp.drs_delta : 0.0) +
             (kp.dirty_pen < 0 ? kp.dirty_pen : 0.0) - 0.5 * 1.5 * z_max;
    return floor_;
}

// Problems the register kernel takes (everything else runs on the generic kernel, with the same results).  Retirement
// probabilities need no condition: the retirement lap of every driver is drawn once per race (draw_retirement_lap,
// race_common.hip.h), where a probability >= 1 is just a survival threshold of 0.
__host__ __device__ inline bool reg_kernel_serves(const KParams &kp)
{
    // an overtake attempt is told from its threshold being non-zero when a lane has more than eight of them in a pass,
    // which wants overtake_delta >= 0 (dl > delta >= 0 makes ceil(dl 2^31) >= 1)
    if (!(kp.overtake_delta >= 0.0)) return false;
    // The kernel's tables carry powers of two (pace x 2^31, degradation x 2^-16 or x 2^15): exact, a power of two
    // commutes with every rounding, unless the scaled value leaves the normal range.  Magnitudes no race has, but the
    // kernel is bit-exact on what it accepts: anything near the ends of binary64 goes to the generic kernel.
    const double big = 0x1p900, tiny = 0x1p-900;
    auto scalable = [&](double x) { return x == 0.0 || (x > -big && x < big && (x >= tiny || x <= -tiny)); };
    for (int d = 0; d < kp.n; ++d) {
        if (!scalable(kp.base_pace[d]) || !scalable(kp.tire_deg[d])) return false;
        for (int c = 0; c < 5; ++c)
            if (!scalable(kp.comp_deg[c] * kp.factor[d])) return false;
    }
    return reg_time_floor(kp) >= kRegTimeFloor;
}

// Phase 1 of a block: fill the block-shared LDS tables (all threads, strided).
template <int N, class G = RegGeo<N>>
__device__ __forceinline__ void reg_load_tables(const KParams *__restrict__ P, unsigned char *smem, uint32_t tid,
                                                const double *__restrict__ norm53 = nullptr)
{
    constexpr int B = G::B;
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(smem + G::oHist);
    double *t_grid = reinterpret_cast<double *>(smem + G::oGrid);          // [slot][driver]
    if constexpr (G::kWide) {
        // the rows of the binary64 inverse-normal table this block keeps in LDS (RegGeo: the table's last kNorm53Rows)
        double *t53 = reinterpret_cast<double *>(smem + G::oNorm53);
        for (uint32_t i = tid; i < (uint32_t)G::kNorm53Rows * 8u; i += B)
            t53[(i >> 3) * (kNorm53LdsStride / 8) + (i & 7u)] = norm53[(uint32_t)G::kNorm53First * 8u + i];
        // the retirement chain of every driver (reference :190-197 drawn once per race, 64-bit form: S_2 = q,
        // S_{k+1} = floor(S_k q / 2^64), q = 2^64 - ceil(p 2^64)), one thread per driver; padding drivers get zeros
        if (G::chain_fits(P->total_laps)) {
            for (uint32_t d = tid; d < G::kChainStride / 8u; d += B) {
                const uint64_t q = d < (uint32_t)N ? P->q64_dnf[d] : 0ull;
                uint64_t S = q;
                for (int k = 2; k <= P->total_laps; ++k) {
                    *reinterpret_cast<uint64_t *>(smem + G::oChain + (uint32_t)(k - 2) * G::kChainStride + d * 8u) = S;
                    S = __umul64hi(S, q);
                }
            }
        }
    } else {
        float4 *t_norm = reinterpret_cast<float4 *>(smem + G::oNorm);
        for (uint32_t i = tid; i < (uint32_t)kNormalRows * 4; i += B)
            reinterpret_cast<uint32_t *>(t_norm)[i] = P->normal_bits[i];
    }
    for (uint32_t d = tid; d < (uint32_t)N; d += B) {
        double *a = reinterpret_cast<double *>(smem + G::oDrvA + d * 16);
        double *b = reinterpret_cast<double *>(smem + G::oDrvB + d * 16);
        const double qnan = __builtin_nan("");
        a[0] = P->variance[d];
        a[1] = P->base_pace[d];
        // overtake pace, scaled by 2^31 (exact: a power of two commutes with every rounding of base + age x deg):
        // the pace delta then comes out as delta x 2^31, whose ceiling IS the integer threshold of the draw word
        b[0] = P->base_pace[d] * 2147483648.0;
        b[1] = P->tire_deg[d] * 32768.0;                         // x 2^31 / 2^16: multiplies the age FIELD of pk (age << 16)
        b[2 * kMaxCars] = qnan;                                  // the retired copy: pace = NaN
        b[2 * kMaxCars + 1] = qnan;
    }
    if (tid < 2) {
        // {0.0, drs_delta}: a car's row is at byte offset (pk & k3Drs)
        *reinterpret_cast<double *>(smem + G::oDrs + tid * k3Drs) = tid ? P->drs_delta : 0.0;
        *reinterpret_cast<double *>(smem + G::oDrsB + tid * k3Drs) = tid ? P->drs_delta * 2147483648.0 : 0.0;
        *reinterpret_cast<double *>(smem + G::oPit + tid * 8u) = tid ? P->pit_loss : 0.0;
    }
    for (uint32_t i = tid; i < (uint32_t)kNumCompounds * N; i += B) {
        const uint32_t c = i / N, d = i % N;
        unsigned char *r = smem + G::oIc + (c * kMaxCars + d) * 16;
        // degradation per lap of tyre age: compound rate x driver factor, reference :319-322; stored x 2^-16 (exact) so
        // that it multiplies the age field of pk as it stands, (age << 16): one mask instead of a bit-field extract
        *reinterpret_cast<double *>(r) = (P->comp_deg[c] * P->factor[d]) * kAgeFieldUnit;
        // pit word: threshold << 16, to be compared with tyre age << 16 taken straight from pk (:454-465: age + 1 > threshold)
        *reinterpret_cast<uint32_t *>(r + 8) = (uint32_t)P->opt_laps[d * kCompStride + c] << 16;
        *reinterpret_cast<uint32_t *>(r + 12) = 0u;
    }
    for (uint32_t i = tid; i < 32u; i += B)
        *reinterpret_cast<uint32_t *>(smem + G::oLut + i * 4) =
            pit_rule_word(P->track, (int)(i >> 3), i & 7u, (uint32_t)P->pop_sh, (uint32_t)P->pop_mh);
    for (uint32_t c = tid; c < (uint32_t)kCompStride; c += B) {
        double *r = reinterpret_cast<double *>(smem + G::oComp + c * 16);
        r[0] = P->comp_delta[c];
        r[1] = 0.0;
    }
    for (uint32_t i = tid; i < (uint32_t)(N * N); i += B) {
        s_hist[i] = 0u;
        t_grid[(i % N) * N + (i / N)] = P->grid_probs[i];       // transposed: lanes gather by driver, conflict-free
    }
}

// Phase 3 of a block: add the block's histogram to the global one (reference :93-94 summed over the block).
template <int N, class G = RegGeo<N>>
__device__ __forceinline__ void reg_flush_hist(unsigned char *smem, uint32_t tid, unsigned long long *__restrict__ hist)
{
    const uint32_t *s_hist = reinterpret_cast<const uint32_t *>(smem + G::oHist);
    for (uint32_t i = tid; i < (uint32_t)(N * N); i += G::B) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&hist[i], (unsigned long long)c);
    }
}

// (a function's arguments arrive in vector registers; these are the same in every lane, which readfirstlane says)
__device__ __forceinline__ uint32_t uniform_u32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
#else
    return x;                                   // (the host debugging build: one thread at a time)
#endif
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t x) { return (uint64_t)uniform_u32((uint32_t)x) | ((uint64_t)uniform_u32((uint32_t)(x >> 32)) << 32); }
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p) { return reinterpret_cast<T *>(uniform_u64(reinterpret_cast<uint64_t>(p))); }
// The exact 53-bit decision of one overtake pass of the reference-width build (reg_simulate, overtakes): the path of a draw
// word that EQUALS the leading word of its threshold, or of a lane with more than eight attempts.  Plain code, four attempts
// at a time: their words in rows 0..3 of the W plane, the companion words in rows 4..7.  Returns bit i = the attempt at pair i
// succeeds.  Everything it reads besides the field's pk words is in the block's LDS tables.
template <int N, class G>
__device__ __forceinline__ uint32_t wide_pass_exact_body(const uint32_t (&pk)[N], double od31, uint32_t lap, uint32_t pass, uint32_t c0l,
                                                         uint32_t c1l, uint32_t k0l, uint32_t k1l, uint32_t tid)
{
    constexpr int B = G::B;
    const uint32_t tid4 = tid * 4u;
    auto w_row_ = [&](int r) -> uint32_t { return G::oW + (uint32_t)r * (B * 4) + tid4; };
    uint32_t hits53 = 0u;
    uint64_t thr64[N];
    uint32_t cand = 0u;
    {
        double pace_prev = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const f64x2 bd = lds_ld_f64x2(G::oDrvB + pk_slot16(pk[i]));
            const double pace = bd.x + (double)(pk[i] & k3AgeMask) * bd.y;
            thr64[i] = 0ull;
            if (i > 0) {
                const double dl = (pace_prev - pace) + lds_ld<double>(G::oDrsB + (pk[i] & k3Drs));
                const bool c = dl > od31;                                           // :522 (NaN: retired)
                double x = ceil_f64(dl * 0x1p21);
                x = x < 0x1p52 ? x : 0x1p52;
                const uint32_t xh = c ? cvt_u32_f64_sat(x * 0x1p-32) : 0u;
                const uint32_t xl = c ? cvt_u32_f64_sat(x - (double)xh * 0x1p32) : 0u;
                thr64[i] = ((uint64_t)xh << 32) | (uint64_t)xl;
                cand |= c ? (1u << i) : 0u;
            }
            pace_prev = pace;
        }
    }
    uint32_t rest = cand;
#pragma unroll 1
    for (int chunk = 0; MCGP_ANY(rest != 0u); ++chunk) {
        uint32_t m = rest;                                  // `rest` without its 4 lowest set bits
#pragma unroll 1
        for (int k = 0; k < 4 && m != 0u; ++k) m &= m - 1u;
        const uint32_t cur = rest & ~m;                     // attempts 4 chunk .. 4 chunk + 3
        uint32_t o0, o1, o2, o3;
        philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeOvt | (uint32_t)(8 * pass + chunk), k0l, k1l, o0, o1, o2, o3);
        lds_st<uint32_t>(w_row_(0), o0);
        lds_st<uint32_t>(w_row_(1), o1);
        lds_st<uint32_t>(w_row_(2), o2);
        lds_st<uint32_t>(w_row_(3), o3);
        philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeOvt | kCompanion | (uint32_t)(8 * pass + chunk), k0l, k1l, o0, o1, o2, o3);
        lds_st<uint32_t>(w_row_(4), o0);
        lds_st<uint32_t>(w_row_(5), o1);
        lds_st<uint32_t>(w_row_(6), o2);
        lds_st<uint32_t>(w_row_(7), o3);
        uint32_t h = 0u;
#pragma unroll
        for (int i = 1; i < N; ++i) {
            // (a pair outside `cur` reads some other words, or just past the plane: masked out below)
            const uint32_t r = (uint32_t)__popc(cur & ((1u << i) - 1u)) * (uint32_t)(B * 4) + tid4;
            const uint64_t u = uniform53(lds_ld<uint32_t>(G::oW + r), lds_ld<uint32_t>(G::oW + (uint32_t)(4 * B * 4) + r));
            h |= u < thr64[i] ? (1u << i) : 0u;
        }
        hits53 |= h & cur;
        rest = m;
    }
    return hits53;
}
// The same OUT OF LINE, for the builds at 3 waves per SIMD (up to 22 cars): inlined, its forty registers of 64-bit thresholds
// are the race loop's to pay for -- 880 B of scratch per lane at 168 registers; as a call it costs the loop about 20 scratch
// accesses per lap (what lives across a call is saved where it is defined: profiles/r5_ab.txt, MCGP_NOCALLS).  The field's
// pk words go in by value.  The builds at 2 waves per SIMD (23 cars and more, 256 registers) inline the body: a 32-car
// build with the call gave wrong results in most simulations in three of six otherwise equivalent variants of the
// surrounding code (a call inside divergent control flow of a kernel that spills 214 scalar registers to vector lanes;
// tools/dbg_wide_n.py, profiles/r5_ab.txt) -- the call stays where every build of it has been checked at scale
// (profiles/r5_deep_parity_wide.txt, also with every pass sent through it: MCGP_WIDE_EXACT=1).
template <int N>
struct PkWords {
    uint32_t v[N];
};
template <int N, class G>
__device__ __attribute__((noinline)) uint32_t wide_pass_exact_fn(PkWords<N> field, double od31, uint32_t lap, uint32_t pass, uint32_t c0l,
                                                                 uint32_t c1l, uint32_t k0l_, uint32_t k1l_, uint32_t tid)
{
    // (the key is the same in every lane; a function's arguments arrive in vector registers)
    return wide_pass_exact_body<N, G>(field.v, od31, lap, pass, c0l, c1l, uniform_u32(k0l_), uniform_u32(k1l_), tid);
}

// Phase 2: the simulations, one full race per lane.  The unit of work is a WAVE-CHUNK of 64 consecutive simulations,
// and every wave of the launch claims its next chunk from one ticket counter in device memory (zeroed by the host
// before the launch) until the chunks run out: a wave that runs faster than its neighbours takes more of them instead
// of idling at the end.  That matters at an odd number of waves per SIMD, where the VALU issue slots are not shared
// evenly (tools/valu_peak.hip: with a fixed share per wave, 3 waves per SIMD ran 14 % slower), and it evens out blocks,
// CUs and the tail of the launch as well.  Which wave runs a simulation changes nothing in its result: every draw is
// addressed by the simulation's global id.
// WIDE = the reference's deviate width (mcgp_config.deviates = MCGP_DEVIATES_53): every draw keeps the word the default
// mode reads as its leading 32 bits and takes 21 more from the same word position of a companion Philox block (counter
// word 3 | kCompanion); Bernoulli tests compare 53-bit numerators with 53-bit thresholds, normals are binary64
// (normal53, race_common.hip.h; `norm53` = its table in device memory).  The oracle's PHILOX53 back-end is the same
// arithmetic.  A priced option: everything WIDE sits behind `if constexpr`, the default kernel is untouched by it.
template <int N, bool WIDE = false, class G = RegGeo<N>>
__device__ __forceinline__ void reg_simulate(const KParams *__restrict__ P, unsigned char *smem, uint32_t tid,
                                             uint32_t *__restrict__ ticket, uint64_t n_sims,
                                             uint64_t sim_offset, uint32_t seed_lo, uint32_t seed_hi,
                                             uint8_t *__restrict__ orders, const uint8_t *__restrict__ fixed_grid,
                                             uint32_t n_chunks, uint32_t *__restrict__ retire_ws_base, uint32_t ws_stride,
                                             uint32_t ws_first_lane, const double *__restrict__ norm53 = nullptr)
{
    constexpr int B = G::B;
    // slots in flight together in the lap step / in an overtake pass (the reference-width build has 256 registers to spend)
    constexpr int kStepBatch = WIDE ? MCGP_WIDE_STEP_BATCH : MCGP_STEP_BATCH;
    // (WIDE, compiled for 3 waves per SIMD: deviates first, gathers last, one table row at a time -- see MCGP_WIDE_LEAN)
    constexpr bool kLean = WIDE && wide_min_waves(N) >= 3;
    constexpr int kPaceBatch = WIDE ? MCGP_WIDE_PACE_BATCH : MCGP_PACE_BATCH;
    static_assert(kStepBatch % 4 == 0, "a Philox block serves four consecutive places");
    const uint32_t tid4 = tid * 4u, tid8 = tid * 8u;
    const int L = P->total_laps;
    const int track = P->track;

    // ---- typed views of the LDS map (RegGeo) for the cold paths (grid sampling, histogram) ----
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(smem + G::oHist);

    // per-lane rows by ABSOLUTE LDS address (race_isa.hip.h): a compile-time row is an immediate offset ...
    auto w_row = [&](int r) -> uint32_t { return G::oW + (uint32_t)r * (B * 4) + tid4; };
    auto l_row = [&](int d) -> uint32_t { return G::oLast + (uint32_t)d * (B * 8) + tid8; };
    // ... the LAST row of the driver held in a pk word is one bit-field extract and one multiply-add (the base
    // G::oLast goes into the DS immediate) ...
    auto last_of = [&](uint32_t p) -> uint32_t { return ((p >> k3IdShift) & 31u) * (uint32_t)(B * 8) + tid8; };
    // ... and the byte of grid slot `pos` in the W plane (the sampled grid, before lap 1)
    auto grid_byte = [&](int pos) -> uint32_t { return G::oW + (uint32_t)(pos >> 2) * (B * 4) + tid4 + (uint32_t)(pos & 3); };
    auto norm_row = [&](uint32_t off) -> float4 { return lds_ld_float4(G::oNorm - kNormalRowBias * 16u + off); };
    // WIDE: the binary64 deviate of a draw from the table rows in LDS (normal53_prepare / normal53_evaluate, race_common.hip.h).
    // `hi` (the leading word of double(q)) names the row; a row the block does not hold -- the cells of the smallest tail
    // indices, RegGeo::kNorm53First -- is flagged by hi < kRareHi and read from the table in device memory (normal53()).
    constexpr uint32_t kRareHi = (kNormal53HiBias + (uint32_t)(G::kNorm53First > 16 ? G::kNorm53First : 16)) << 16;
    [[maybe_unused]] auto norm53_row = [&](uint32_t hi, double (&c)[8]) {
        const uint32_t a = (hi >> 16) * (uint32_t)kNorm53LdsStride + (G::oNorm53 - (kNormal53HiBias + (uint32_t)G::kNorm53First) * (uint32_t)kNorm53LdsStride);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f64x2 v = lds_ld_f64x2(a + 16u * k);
            c[2 * k] = v.x;
            c[2 * k + 1] = v.y;
        }
    };
    [[maybe_unused]] auto normal53_one = [&](uint32_t w, uint32_t x) -> double {
        uint32_t hi;
        double t;
        normal53_prepare(w, x, hi, t);
        if (__builtin_expect(MCGP_ANY(G::kNorm53Rows == 0 || MCGP_WIDE_EXACT || hi < kRareHi), 0)) return normal53(w, x, norm53);
        double c[8];
        norm53_row(hi, c);
        return normal53_evaluate(w, c, t);
    };

    const double dirty_thr = P->dirty_thr;                     // (lap 1; the lap loop reads its constants per lap)
    const float kNaN = __uint_as_float(0x7fc00000u);

    // Everything the lap step of one slot reads from LDS.
    struct SlotIn {
        uint32_t pitw;          // pit threshold << 16
        double last, var, base, eff, cdelta, drs;
        uint32_t la;            // byte offset of the driver's LAST row
        uint32_t fit;           // compound and used-set a pit stop on this lap would leave, positioned as in pk
    };
    auto load_slot = [&](uint32_t p, uint32_t lut_base) -> SlotIn {
        SlotIn r;
        r.fit = lds_ld<uint32_t>(((p & k3UsedMask) << 2) + lut_base);
        const uint32_t id16 = pk_driver16(p);                                     // 16 x driver
        r.la = id16 * (uint32_t)(B / 2) + tid8;                                   // = driver x (B x 8) + tid8
        r.last = lds_ld<double>(G::oLast + r.la);
        const uint32_t ic = id16 + pk_comp512(p);                                 // 16 x (32 compound + driver)
        const f64x2 vb = lds_ld_f64x2(G::oDrvA + id16);
        r.var = vb.x;
        r.base = vb.y;
        uint32_t pad;
        lds_ld_f64_u32x2(G::oIc + ic, r.eff, r.pitw, pad);
        r.cdelta = lds_ld<double>(G::oComp + pk_comp16(p));                        // 16 x compound
        r.drs = lds_ld<double>(G::oDrs + (p & k3Drs));                            // 0.0 or drs_delta
        return r;
    };

    for (uint32_t turn = 0u;; ++turn) {
        const uint32_t chunk = next_ticket(ticket, tid, turn, G::kWaves);
        if (chunk >= n_chunks) break;                            // (every wave ends on its first ticket past the end)
        const uint64_t local = (uint64_t)chunk * 64ull + (uint64_t)(tid & 63u);
        // A lane past the end of the run still runs a race while any lane of its wave has one to run -- the wave's
        // lanes work for each other in the event handler (lane = car) -- but records nothing.
        const bool live = local < n_sims;
        if (!MCGP_ANY(live)) continue;
        const uint64_t sim = sim_offset + local;
        const uint32_t c0 = (uint32_t)sim, c1 = (uint32_t)(sim >> 32);

        double cum[N];
        uint32_t pk[N];

        // ================= _sample_grid, reference :102-145 =================
        // One grid slot per iteration, all N drivers in straight-line code (like the reference's list comprehensions,
        // which also run over every driver with zeros for the placed ones: adding +0.0 changes no sum, and a placed
        // driver's cdf entry repeats its predecessor's, so it is never the first one above u).  The column of the
        // grid matrix is the same for every lane: N wave-uniform LDS reads; everything else stays in registers.
        {
            uint32_t remaining = (N >= 32) ? 0xffffffffu : ((1u << N) - 1u);
            int n_remaining = N;
            uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll 1
            for (int pos = 0; pos < N; ++pos) {
                uint32_t sel;
                if (fixed_grid) {
                    sel = fixed_grid[pos];
                } else if (MCGP_SKIP & 4) {
                    sel = (uint32_t)((pos * 7 + (int)(c0 & 3u)) % N);
                    while (!((remaining >> sel) & 1u)) sel = (sel + 1u) % (uint32_t)N;
                } else {
                    if ((pos & 3) == 0) {
                        philox4x32_10(c0, c1, 0u, kPurposeGrid | (uint32_t)(pos >> 2), seed_lo, seed_hi, g0, g1, g2, g3);
                        if constexpr (WIDE)
                            philox4x32_10(c0, c1, 0u, kPurposeGrid | kCompanion | (uint32_t)(pos >> 2), seed_lo, seed_hi, h0, h1, h2, h3);
                    }
                    const uint32_t gw = (pos & 3) == 0 ? g0 : (pos & 3) == 1 ? g1 : (pos & 3) == 2 ? g2 : g3;
                    const uint32_t hw = (pos & 3) == 0 ? h0 : (pos & 3) == 1 ? h1 : (pos & 3) == 2 ? h2 : h3;
                    const double u = WIDE ? (double)uniform53(gw, hw) * 0x1p-53 : u32_to_unit(gw);
                    const uint32_t gcol = G::oGrid + (uint32_t)(pos * N) * 8u;           // [slot][driver], wave-uniform
                    // Fast path, no division.  The reference normalises the column over the remaining drivers (:119-126),
                    // numpy's choice() normalises the cumulative sums once more and takes the first entry above u: the
                    // value compared with u is  R_d = fl(fl-sum_{j<=d} fl(p_j / total) / cdf_last),  within
                    // (2 n + 6) 2^-53 (relative) of  S_d / S_n,  the exact partial sums of the column -- every term is
                    // non-negative, so rounding errors stay relative and the common factor 1 / total cancels.  The partial
                    // sums A_d computed here and uc = fl(u A_n) carry errors of the same size, so outside a band of 2^-40
                    // around the threshold the comparison  A_d > u A_n  decides  R_d > u  whatever the roundings did; a draw
                    // inside the band (about one in 10^10), a column without mass or a sum near the ends of binary64 takes
                    // the exact path below, division by division.  (Placed drivers add +0.0 and repeat their predecessor's
                    // partial sum: never the first one above.  u = 0 makes both bounds 0: the first positive sum.)
                    double acc = 0.0;
                    double A[N];
#pragma unroll
                    for (int d = 0; d < N; ++d) {
                        const double pd = ((remaining >> d) & 1u) ? lds_ld<double>(gcol + 8u * d) : 0.0;
                        acc = acc + pd;                                                  // :119-123 (the reference's `total`)
                        A[d] = acc;
                    }
                    const double total = acc;
                    const double uc = u * total;
                    const double sure_gt = uc * (1.0 + 0x1p-40), sure_le = uc * (1.0 - 0x1p-40);
                    uint32_t n_le = 0u, n_le_lo = 0u;                 // partial sums not above the upper / the lower bound
#pragma unroll
                    for (int d = 0; d < N; ++d) {
                        n_le += A[d] > sure_gt ? 0u : 1u;
                        n_le_lo += A[d] > sure_le ? 0u : 1u;
                    }
                    // (the sums are non-decreasing: the first one above the upper bound is entry n_le, and some sum lies
                    //  inside the band iff the two counts differ)
                    const bool exact_path = MCGP_GRID_EXACT || !(total > 0x1p-900 && total < 0x1p900) || n_le != n_le_lo || n_le >= (uint32_t)N;
                    sel = n_le;
                    if (__builtin_expect(MCGP_ANY(exact_path), 0)) {
                        double p[N];
#pragma unroll
                        for (int d = 0; d < N; ++d) p[d] = ((remaining >> d) & 1u) ? lds_ld<double>(gcol + 8u * d) : 0.0;
                        const bool has_mass = total > 0;
                        const double uniform_p = 1.0 / (double)n_remaining;                  // :127-130
                        double prob_sum = 0.0;                                               // :125-133
#pragma unroll
                        for (int d = 0; d < N; ++d) {
                            const double q = has_mass ? p[d] / total : uniform_p;
                            p[d] = ((remaining >> d) & 1u) ? q : 0.0;
                            prob_sum = prob_sum + p[d];
                        }
                        if (prob_sum > 0 && fabs(prob_sum - 1.0) > 1e-9) {                   // :134-135 (practically never)
#pragma unroll
                            for (int d = 0; d < N; ++d) p[d] = p[d] / prob_sum;
                        }
                        // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, 'right'): the first driver
                        // with cdf[d] / cdf_last > u.  The comparison is strict: a cdf entry of 0 -- the drivers placed
                        // before the first remaining one -- is not above u = 0.
                        double cdf = 0.0;
#pragma unroll
                        for (int d = 0; d < N; ++d) {
                            cdf = cdf + p[d];
                            p[d] = cdf;
                        }
                        const double cdf_last = cdf;
                        uint32_t above = 0u;
#pragma unroll
                        for (int d = 0; d < N; ++d) above |= !(p[d] / cdf_last <= u) ? (1u << d) : 0u;
                        const uint32_t sel_exact = above ? (uint32_t)__ffs((int)above) - 1u : 31u - (uint32_t)__clz((int)remaining);   // (never empty: cdf[-1] == 1 > u)
                        sel = exact_path ? sel_exact : sel;
                    }
                }
                if ((remaining >> sel) & 1u) { remaining &= ~(1u << sel); --n_remaining; }
                lds_st<uint8_t>(grid_byte(pos), (uint8_t)sel);
            }
        }

        // ================= _initialize_cars, reference :244-273 (slot i = grid position i) =================
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t id = lds_ld<uint8_t>(grid_byte(i));
            uint32_t comp, age;
            if (track == 2) { comp = 4u; age = 0u; }
            else if (track == 1) { comp = 3u; age = 0u; }
            else { comp = i < 10 ? 0u : 1u; age = i < 10 ? 4u : 0u; }
            pk[i] = (age << k3AgeShift) | (comp << k3CompShift) | ((1u << comp) & k3UsedMask) | (id << k3IdShift) |
                    ((uint32_t)i << k3GposShift);
            cum[i] = 0.0;
        }

        // ================= _simulate_lap_1, reference :275-311 =================
        if constexpr (WIDE) {
            // reference-width deviates: two passes over the drivers, each staging one binary64 deviate per driver in the
            // LAST rows -- first the lap noise (NaN = retired on lap 1), then the start delta
#pragma unroll 1
            for (int d = 0; d < N; ++d) {
                uint32_t w0, w1, w2, w3, x0, x1, x2, x3;
                philox4x32_10(c0, c1, 1u, kPurposeCar | (uint32_t)d, seed_lo, seed_hi, w0, w1, w2, w3);
                philox4x32_10(c0, c1, 1u, kPurposeCar | kCompanion | (uint32_t)d, seed_lo, seed_hi, x0, x1, x2, x3);
                const bool out = uniform53(w0, x0) < P->t53_dnf1[d];
                lds_st<double>(l_row(d), out ? __builtin_nan("") : normal53_one(w1, x1));
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t p = pk[i];
                const double z = lds_ld<double>(G::oLast + last_of(p));
                const uint32_t id16 = pk_driver16(p);
                const f64x2 vb = lds_ld_f64x2(G::oDrvA + id16);                        // {variance, base pace}
                const double eff = lds_ld<double>(G::oIc + id16 + pk_comp512(p));
                const double cdelta = lds_ld<double>(G::oComp + pk_comp16(p));
                if (z != z) {
                    pk[i] = (p & ~k3AgeMask) | k3Dnf | (1u << k3AgeShift);
                    cum[i] = -(double)(i + 1) * 0x1p-1000;                              // (see the default path below)
                } else {
                    const double tire = (double)(p & k3AgeMask) * eff;
                    const double fuel_effect = (110.0 - 110.0) * 0.03;
                    const double noise = 0.0 + vb.x * z;
                    cum[i] = vb.y + tire - fuel_effect + cdelta - 0.0 + noise;          // base_lap, finished below
                }
            }
#pragma unroll 1
            for (int d = 0; d < N; ++d) {
                uint32_t w0, w1, w2, w3, x0, x1, x2, x3;
                philox4x32_10(c0, c1, 1u, kPurposeCar | (uint32_t)d, seed_lo, seed_hi, w0, w1, w2, w3);
                philox4x32_10(c0, c1, 1u, kPurposeCar | kCompanion | (uint32_t)d, seed_lo, seed_hi, x0, x1, x2, x3);
                lds_st<double>(l_row(d), normal53_one(w2, x2));
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t p = pk[i];
                const double zs = lds_ld<double>(G::oLast + last_of(p));
                if (!(p & k3Dnf)) {
                    double pf = 0.5 + (double)(i + 1) * 0.1;
                    if (!(pf < 1.5)) pf = 1.5;
                    double sd = 0.0 + pf * zs;
                    if (i + 1 <= 3 && 1.0 < sd) sd = 1.0;
                    cum[i] = 0.0 + (cum[i] - sd * 0.5);
                    pk[i] = p + (1u << k3AgeShift);
                }
            }
        } else {
            // draws by DRIVER (wave-uniform thresholds), staged in the LAST rows, which lap 1 does not use otherwise (Q3):
            // low word = lap-noise deviate or NaN (retired), high word = start deviate
#pragma unroll 1
            for (int d = 0; d < N; ++d) {
                uint32_t w0, w1, w2, w3;
                philox4x32_10(c0, c1, 1u, kPurposeCar | (uint32_t)d, seed_lo, seed_hi, w0, w1, w2, w3);
                const bool out = (uint64_t)w0 < P->t_dnf1[d];
                lds_st<float>(l_row(d), out ? kNaN : normal_from_u32_rows(w1, norm_row));
                lds_st<float>(l_row(d) + 4u, normal_from_u32_rows(w2, norm_row));
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t p = pk[i];
                const uint32_t la = last_of(p);
                const float z = lds_ld<float>(G::oLast + la);
                const double zs = (double)lds_ld<float>(G::oLast + la + 4u);
                const uint32_t id16 = pk_driver16(p);
                const f64x2 vb = lds_ld_f64x2(G::oDrvA + id16);                        // {variance, base pace}
                const double eff = lds_ld<double>(G::oIc + id16 + pk_comp512(p));
                const double cdelta = lds_ld<double>(G::oComp + pk_comp16(p));
                if (z != z) {
                    pk[i] = (p & ~k3AgeMask) | k3Dnf | (1u << k3AgeShift);
                    // The reference leaves a lap-1 retirement at cumulative_time 0.0, so several of them tie; ties sort
                    // in grid order.  Here the car on grid slot i is given -(i + 1) * 2^-1000 instead: still below every
                    // running car's time, never equal to anything, and among themselves in the order the only consumer
                    // of their relative order wants -- the classification sorts retirements of one lap by time
                    // DESCENDING, i.e. grid slot ascending, exactly the reference's stable tie.  (Overtakes, events and
                    // _update_positions skip retired cars; nothing else reads these values.)  Keeping them apart lets
                    // "strictly increasing times" be the sortedness test of the hot loop.
                    cum[i] = -(double)(i + 1) * 0x1p-1000;
                } else {
                    const double tire = (double)(p & k3AgeMask) * eff;                          // (eff x 2^-16)
                    const double fuel_effect = (110.0 - 110.0) * 0.03;
                    const double noise = 0.0 + vb.x * (double)z;
                    const double base_lap = vb.y + tire - fuel_effect + cdelta - 0.0 + noise;
                    double pf = 0.5 + (double)(i + 1) * 0.1;
                    if (!(pf < 1.5)) pf = 1.5;
                    double sd = 0.0 + pf * zs;
                    if (i + 1 <= 3 && 1.0 < sd) sd = 1.0;
                    cum[i] = 0.0 + (base_lap - sd * 0.5);
                    pk[i] = p + (1u << k3AgeShift);
                }
            }
        }
#pragma unroll 1
        for (int d = 0; d < N; ++d) lds_st<double>(l_row(d), 0.0);          // last_lap_time is not set on lap 1 (Q3)
        network_sort<N>(cum, pk);
        update_positions_reg<N>(cum, pk, false, dirty_thr);

        // ================= retirements of laps 2..L, reference :190-197, drawn once per race =================
        // The reference draws one uniform per running car per lap and retires the car if it is below the driver's per-lap
        // probability p: the lap of the first success is geometric, independent of everything else in the race.  It is
        // drawn HERE, from one word per driver (draw_retirement_lap, race_common.hip.h), which halves the Philox blocks
        // of the lap step (a block then serves four cars' lap noise instead of two cars' noise + retirement draws).
        // The survival thresholds S_k are wave-uniform (scalar unit: one s_mul_hi per lap); a lane counts the laps its
        // word survives.  A lane's retirements go to its column of `retire_ws` (device memory: about one entry per
        // race, read back only on the lap of a retirement) as keys lap << 5 | driver; the smallest is kept in `next_out`.
        // `next_out` holds the lane's next two retirements, 15-bit keys (lap << 5 | driver, 0x7FFF = none) in bits 0..14
        // and 16..30, bit 31 = "the list in device memory has more": the race loop tests one register per lap and goes to
        // memory only for a lane's third, fifth .. retirement.
        constexpr uint32_t kNoKey = 0x7FFFu;
        // the two smallest keys above `after` of this lane's list in device memory, packed as above
        auto retirements_after = [&](const uint32_t *ws, uint32_t after) -> uint32_t {
            const uint32_t n_out = ws[(size_t)N * ws_stride];
            uint32_t k1 = kNoKey, k2 = kNoKey, left = 0u;
#pragma unroll 1
            for (uint32_t k = 0; k < n_out; ++k) {
                const uint32_t e = ws[(size_t)k * ws_stride];
                if (e > after) {
                    ++left;
                    const uint32_t hi = e > k1 ? e : k1;
                    k1 = e < k1 ? e : k1;
                    k2 = hi < k2 ? hi : k2;
                }
            }
            return k1 | (k2 << 16) | (left > 2u ? 0x80000000u : 0u);
        };
        uint32_t next_out;
        {
            // (this lane's column of the workspace: formed where it is used, from a thread index the optimiser cannot trace
            //  back, so that no 64-bit address sits in registers for the whole race)
            uint32_t tid_w = tid;
            pin(tid_w);
            uint32_t *retire_ws = retire_ws_base + ws_first_lane + tid_w;
            uint32_t n_out = 0u, k1 = kNoKey, k2 = kNoKey;
            // four drivers at a time (the four words of one Philox block): their survival thresholds advance together on
            // the scalar unit, one loop iteration per lap for the four of them
#pragma unroll 1
            for (int d0 = 0; d0 < N; d0 += 4) {
                uint32_t rw[4];
                philox4x32_10(c0, c1, 0u, kPurposeRetire | (uint32_t)(d0 >> 2), seed_lo, seed_hi, rw[0], rw[1], rw[2], rw[3]);
                uint32_t survived[4];
                if constexpr (WIDE) {
                    // the word refined to 53 bits (left-aligned in 64) against 64-bit thresholds S_2 = q,
                    // S_{k+1} = floor(S_k q / 2^64), q = 2^64 - ceil(p 2^64) (P->q64_dnf; the oracle's philox_retirement_lap)
                    uint32_t rx[4];
                    philox4x32_10(c0, c1, 0u, kPurposeRetire | kCompanion | (uint32_t)(d0 >> 2), seed_lo, seed_hi, rx[0], rx[1], rx[2], rx[3]);
                    uint64_t Q[4], q64[4], S64[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Q[j] = ((uint64_t)rw[j] << 32) | ((uint64_t)(rx[j] >> 11) << 11);
                        q64[j] = d0 + j < N ? P->q64_dnf[d0 + j] : 0ull;
                        S64[j] = q64[j];
                        survived[j] = 0u;
                    }
                    if (G::chain_fits(L)) {
                        // the thresholds of these four drivers, lap by lap, from the block's table (wave-uniform reads)
                        uint32_t a = G::oChain + (uint32_t)d0 * 8u;
#pragma unroll 1
                        for (int k = 2; k <= L; ++k) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) survived[j] += Q[j] < lds_ld<uint64_t>(a + 8u * j) ? 1u : 0u;
                            a += G::kChainStride;
                        }
                    } else {
#pragma unroll 1
                    for (int k = 2; k <= L; ++k) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            survived[j] += Q[j] < S64[j] ? 1u : 0u;
                            S64[j] = __umul64hi(S64[j], q64[j]);
                        }
                    }
                    }
                } else {
                uint32_t q[4], S[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint64_t t = d0 + j < N ? P->t_dnf[d0 + j] : 0ull;         // ceil(p 2^32), 0 .. 2^32
                    // p = 0 (t = 0) never retires: a threshold no word reaches ... q = 2^32 does not fit, so such a
                    // driver is skipped below; p >= 1 (t = 2^32) has q = 0 and retires on lap 2
                    q[j] = (uint32_t)(4294967296ull - t);
                    S[j] = q[j];
                    survived[j] = 0u;
                }
#pragma unroll 1
                for (int k = (MCGP_SKIP & 256) ? L + 1 : 2; k <= L; ++k) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        survived[j] += rw[j] < S[j] ? 1u : 0u;                       // survives lap k
                        S[j] = retire_next_threshold(S[j], q[j]);
                    }
                }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = d0 + j;
                    if (d >= N) continue;
                    if (P->t_dnf[d] == 0ull) continue;                               // (wave-uniform)
                    const uint32_t out_lap = 2u + survived[j];
                    if (out_lap <= (uint32_t)L && !(MCGP_SKIP & 256)) {
                        const uint32_t key = (out_lap << 5) | (uint32_t)d;
                        // the list in device memory is only ever read by a lane with MORE than two retirements: the
                        // first two are written when a third turns up (about one race in thirteen), not before
                        if (n_out == 2u) {
                            retire_ws[0] = k1;
                            retire_ws[(size_t)ws_stride] = k2;
                        }
                        if (n_out >= 2u) retire_ws[(size_t)n_out * ws_stride] = key;
                        ++n_out;
                        const uint32_t hi = key > k1 ? key : k1;
                        k1 = key < k1 ? key : k1;
                        k2 = hi < k2 ? hi : k2;
                    }
                }
            }
            if (n_out > 2u) retire_ws[(size_t)N * ws_stride] = n_out;
            next_out = k1 | (k2 << 16) | (n_out > 2u ? 0x80000000u : 0u);
        }

        // ================= laps 2..L, reference :166-228 =================
        int drs_disabled_until = 0;
#pragma unroll 1
        for (int lap = 2; lap <= ((MCGP_SKIP & 8) ? 1 : L); ++lap) {
            const int remaining_laps = L - lap;
            // The simulation id as this lap sees it: opaque to the optimiser, so that it does not lift the lap-
            // invariant part of every Philox block of the loop body (round 1 depends on the id and the block's
            // constant address only) out of the lap loop -- a dozen blocks x 3 registers held across the whole race.
            // Within the lap the shared part is still computed once.
            // (re-derived from the wave's chunk number -- scalar -- and the lane number rather than copied from c0 / c1:
            //  the simulation id then needs no register of its own across the lap loop)
            uint32_t lane_l = tid;
            pin(lane_l);
            const uint64_t sim_l = sim_offset + ((uint64_t)chunk * 64ull + (uint64_t)(lane_l & 63u));
            uint32_t c0l = (uint32_t)sim_l, c1l = (uint32_t)(sim_l >> 32);
            pin(c0l);
            pin(c1l);
            // Likewise the key: its ten round keys (seed + r x constant, twenty SGPRs) are worked out per lap from an opaque
            // copy, on the scalar unit, instead of living across the whole kernel -- where the register allocator parks
            // them in VGPR lanes and fetches each one back with a v_readlane (a VALU instruction) at every block.
            uint32_t k0l = seed_lo, k1l = seed_hi;
            pin_scalar(k0l);
            pin_scalar(k1l);
            // The per-race constants of the lap body are read from the parameter block EVERY lap (scalar loads through
            // a pointer the optimiser cannot see through) instead of being held in registers across the whole race:
            // eight 64-bit values that would otherwise push as many lane masks and addresses out to scratch.
            const KParams *Pl = P;
            pin_ptr(Pl);
            const double od31 = Pl->overtake_delta_31, dirty_thr = Pl->dirty_thr;
            const double dirty_pen_lap = lap == 2 ? 0.0 : Pl->dirty_pen;         // (no dirty-air constraint in lap 2: see the lap step)
            const uint64_t t_red = WIDE ? Pl->t53_red : Pl->t_red, t_sc = WIDE ? Pl->t53_sc : Pl->t_sc,
                           t_vsc = WIDE ? Pl->t53_vsc : Pl->t_vsc, t_vsc_tire = WIDE ? Pl->t53_vsc_tire : Pl->t_vsc_tire;
            // ---- race-interrupting events, :168-176 ----
            {
                uint32_t e0, e1, e2, e3;
                philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeEvent, k0l, k1l, e0, e1, e2, e3);
                if (MCGP_DUP & 16) {
                    uint32_t f0, f1, f2, f3;
                    philox4x32_10(c0l ^ e0, c1l, (uint32_t)lap, kPurposeEvent, k0l, k1l, f0, f1, f2, f3);
                    if ((f0 | f1 | f2 | f3) == 0u) e0 = f0;     // never true in practice; keeps the block alive
                }
                // the lap's four Bernoulli draws (:168, :171, :174, :392)
                bool d_red, d_sc, d_vsc, d_tire;
                if constexpr (WIDE) {
                    // 53-bit numerator q = w 2^21 + (21 bits of the companion word) against T = ceil(p 2^53): the word alone
                    // decides -- q < T iff w < T >> 21 -- unless w == T >> 21, one draw in 2^32.  The companion block is
                    // computed only for a wave with such a draw.  (T = 2^53, p >= 1: every word is below.)
                    const uint32_t h_red = (uint32_t)(t_red >> 21), h_sc = (uint32_t)(t_sc >> 21), h_vsc = (uint32_t)(t_vsc >> 21),
                                   h_tire = (uint32_t)(t_vsc_tire >> 21);
                    const bool a_red = (t_red >> 53) != 0, a_sc = (t_sc >> 53) != 0, a_vsc = (t_vsc >> 53) != 0,
                               a_tire = (t_vsc_tire >> 53) != 0;
                    auto near = [](uint32_t w, uint32_t h) { return ((w ^ h) >> MCGP_WIDE_TIE_SHIFT) == 0u; };
                    const bool tie = (!a_red && near(e0, h_red)) || (!a_sc && near(e1, h_sc)) || (!a_vsc && near(e2, h_vsc)) ||
                                     (!a_tire && near(e3, h_tire));
                    if (__builtin_expect(!MCGP_ANY(tie || MCGP_WIDE_EXACT), 1)) {
                        d_red = a_red || e0 < h_red;
                        d_sc = a_sc || e1 < h_sc;
                        d_vsc = a_vsc || e2 < h_vsc;
                        d_tire = a_tire || e3 < h_tire;
                    } else {
                        uint32_t f0, f1, f2, f3;
                        philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeEvent | kCompanion, k0l, k1l, f0, f1, f2, f3);
                        d_red = uniform53(e0, f0) < t_red;
                        d_sc = uniform53(e1, f1) < t_sc;
                        d_vsc = uniform53(e2, f2) < t_vsc;
                        d_tire = uniform53(e3, f3) < t_vsc_tire;
                    }
                } else {
                    d_red = (uint64_t)e0 < t_red;
                    d_sc = (uint64_t)e1 < t_sc;
                    d_vsc = (uint64_t)e2 < t_vsc;
                    d_tire = (uint64_t)e3 < t_vsc_tire;
                }
                const bool red = d_red;
                const bool sc = !red && d_sc;
                const bool vsc = !red && !sc && d_vsc;
                MCGP_STAT(12, red || sc || vsc);
#if MCGP_COOPERATIVE_EVENTS
                // An event is rare per simulation (2.7 % of laps on the benchmark fields) but not per wave: 83 % of
                // wave-laps have one in SOME lane, and a handler run by every lane for the sake of ~2 of them was a
                // tenth of the kernel.  So the wave turns the problem round: a lane with an event parks its field in
                // the wave's window of the W plane (free outside the overtake passes), the wave handles one such field
                // (two, in fact: one per half of the wave) at a time with LANE = CAR -- the leader is a ballot + readlane,
                // a car's place among the running cars a popcount of the ballot below it, the neighbour's new time a
                // bpermute -- and the lane takes its field back.
                const bool evt = !(MCGP_SKIP & 2) && (red || sc || vsc);
                const unsigned long long emask = __ballot(evt);
                if (emask != 0ull) {
                    constexpr int kRowsPerField = (align16(8 * N) + 4 * N <= 256) ? 1 : 2;   // cum (8 N bytes) and pk (4 N) of one field
                    constexpr int kFields = kWordRows / kRowsPerField;          // fields parked at a time
                    constexpr uint32_t oPk = kRowsPerField == 1 ? (uint32_t)align16(8 * N) : (uint32_t)(B * 4);   // 16-byte aligned
                    // (lane number, lane masks and window address are worked out HERE, from a thread index the optimiser
                    //  cannot trace back: hoisted out of the race they would sit in registers -- in scratch, as it turned
                    //  out -- for all of it)
                    uint32_t tid_e = tid;
                    pin(tid_e);
                    const uint32_t lane = tid_e & 63u;
                    const uint32_t wbase = G::oW + ((tid_e * 4u) & ~255u);      // this wave's 256-byte window of row 0
                    const bool dec_age = sc || (vsc && d_tire);
                    const uint32_t flags = (red ? 1u : 0u) | (vsc ? 4u : 0u) | (dec_age ? 8u : 0u);
                    const uint32_t newc = stint_compound(track, remaining_laps);
                    const uint32_t red_bits = (newc << k3CompShift) | ((1u << newc) & k3UsedMask);
                    unsigned long long todo = emask, tiemask = 0ull;
                    do {
                        unsigned long long rest = todo;                         // `todo` without its kFields lowest set bits
#pragma unroll 1
                        for (int k = 0; k < kFields && rest != 0ull; ++k) rest &= rest - 1ull;
                        const unsigned long long cur = todo & ~rest;
                        const bool mine = evt && ((cur >> lane) & 1ull);
                        const uint32_t my_base = wbase + (uint32_t)__popcll(cur & ((1ull << lane) - 1ull)) * (uint32_t)(kRowsPerField * B * 4);
                        if (mine) {                                         // park: two times / four pk words per store
#pragma unroll
                            for (int i = 0; i + 1 < N; i += 2) lds_st_f64x2(my_base + 8u * i, cum[i], cum[i + 1]);
                            if (N & 1) lds_st<double>(my_base + 8u * (N - 1), cum[N - 1]);
#pragma unroll
                            for (int i = 0; i + 3 < N; i += 4) lds_st_u32x4(my_base + oPk + 4u * i, pk[i], pk[i + 1], pk[i + 2], pk[i + 3]);
#pragma unroll
                            for (int i = N & ~3; i < N; ++i) lds_st<uint32_t>(my_base + oPk + 4u * i, pk[i]);
                        }
                        wave_sync();
                        // two parked fields at a time: lanes 0..31 are the cars of one, lanes 32..63 of the next
                        uint32_t field = 0u;
                        for (unsigned long long mm = cur; mm != 0ull; field += 2u) {
                            const int LA = __ffsll((long long)mm) - 1;                              // the lanes whose fields these are
                            mm &= mm - 1ull;
                            const bool two = mm != 0ull;
                            const int LB = two ? __ffsll((long long)mm) - 1 : LA;
                            mm &= mm - 1ull;
                            const uint32_t half = lane >> 5, car = lane & 31u;
                            const uint32_t flA = (uint32_t)__builtin_amdgcn_readlane((int)flags, LA);
                            const uint32_t flB = (uint32_t)__builtin_amdgcn_readlane((int)flags, LB);
                            const uint32_t fl = half ? flB : flA;
                            const bool red_L = fl & 1u, vsc_L = fl & 4u;
                            const uint32_t dec_unit = (fl & 8u) ? (1u << k3AgeShift) : 0u;
                            const double step = red_L ? 0.1 : 0.5;
                            const uint32_t base = wbase + (field + half) * (uint32_t)(kRowsPerField * B * 4);
                            bool tie_here = false;
                            if (car < (uint32_t)N && (half == 0u || two)) {
                                const double t = lds_ld<double>(base + 8u * car);
                                const uint32_t p = lds_ld<uint32_t>(base + oPk + 4u * car);
                                const bool act = !(p & k3Dnf);
                                const unsigned long long am2 = __ballot(act);                        // the running cars of both fields
                                const uint32_t amA = (uint32_t)am2, amB = (uint32_t)(am2 >> 32);
                                // (a field without a running car: nothing below is committed for it)
                                const double leadA = readlane_f64(t, amA ? __ffs((int)amA) - 1 : 0);
                                const double leadB = readlane_f64(t, amB ? 32 + __ffs((int)amB) - 1 : 32);
                                const double leader = half ? leadB : leadA;
                                const uint32_t below = (half ? amB : amA) & ((1u << car) - 1u);     // running cars ahead of this one
                                const double kd = (double)__popc(below);
                                const double nt_fixed = leader + kd * step;                          // :363 / :412
                                const double gap = t - leader;
                                const double nt_vsc = leader + gap * 0.8;                            // :386-387
                                const double nt = vsc_L ? nt_vsc : nt_fixed;
                                // x0.8 may round two running cars onto one time: compare with the running car ahead
                                const double pn = bpermute_f64(nt, (int)(half * 32u) + (below ? 31 - __clz((int)below) : 0));
                                tie_here = act && below != 0u && pn == nt;
                                const double tbl = nt - leader;                                      // :371 / :388 / :413
                                uint32_t q = (p & ~k3Dirty) | ((tbl > 0 && tbl < dirty_thr) ? k3Dirty : 0u);
                                const uint32_t agef = q & k3AgeMask;
                                q -= agef < dec_unit ? agef : dec_unit;                              // max(0, tire_age - 1), :375 / :393-395
                                const uint32_t q_red = (q & ~(k3CompMask | k3AgeMask)) | red_bits;   // :414-429
                                q = red_L ? q_red : q;
                                lds_st<double>(base + 8u * car, act ? nt : t);
                                lds_st<uint32_t>(base + oPk + 4u * car, act ? q : p);
                            }
                            // (every lane of the wave records the verdicts: the lanes whose fields these are may not be car lanes)
                            const unsigned long long ties = __ballot(tie_here);
                            if ((uint32_t)ties != 0u) tiemask |= 1ull << LA;
                            if ((uint32_t)(ties >> 32) != 0u) tiemask |= 1ull << LB;
                        }
                        wave_sync();
                        if (mine) {                                         // take the field back
#pragma unroll
                            for (int i = 0; i + 1 < N; i += 2) {
                                const f64x2 v = lds_ld_f64x2(my_base + 8u * i);
                                cum[i] = v.x;
                                cum[i + 1] = v.y;
                            }
                            if (N & 1) cum[N - 1] = lds_ld<double>(my_base + 8u * (N - 1));
#pragma unroll
                            for (int i = 0; i + 3 < N; i += 4) lds_ld_u32x4(my_base + oPk + 4u * i, pk[i], pk[i + 1], pk[i + 2], pk[i + 3]);
#pragma unroll
                            for (int i = N & ~3; i < N; ++i) pk[i] = lds_ld<uint32_t>(my_base + oPk + 4u * i);
                        }
                        wave_sync();                                        // before the next batch of fields reuses the window
                        todo = rest;
                    } while (todo != 0ull);
                    if (evt) {
                        drs_disabled_until = lap + (vsc ? 1 : 2);
                        // Re-spacing keeps the running cars' relative order (the only order the lap loop below needs;
                        // the full order is rebuilt after it), and the FIELD ORDER that addresses this lap's draws stays
                        // what the last lap left.  Equal times must fall back to grid order: re-sort only then (the
                        // oracle does the same).
                        if ((tiemask >> lane) & 1ull) transposition_sort<N>(cum, pk);
                    }
                }
#else
                if (!(MCGP_SKIP & 2) && (red || sc || vsc)) {
                    // Only a few lanes of a wave are in here, but the wave pays for every instruction: the handlers
                    // (:334-431) are one pass over the ranks without branches.  Running cars are re-spaced behind the
                    // leader (red flag 0.1 s apart, safety car 0.5 s apart, VSC gaps x 0.8), their time_behind_leader
                    // -- kept as the dirty-air flag -- follows, tyres age one lap less (SC; VSC with probability 0.3)
                    // or are changed (red flag).  Retired cars are left alone.
                    const bool dec_age = sc || (vsc && d_tire);
                    const uint32_t newc = stint_compound(track, remaining_laps);
                    const uint32_t red_bits = (newc << k3CompShift) | ((1u << newc) & k3UsedMask);
                    const uint32_t dec_unit = dec_age ? (1u << k3AgeShift) : 0u;
                    const double step = red ? 0.1 : 0.5;
                    double leader = 0.0, kd = 0.0, prev_nt = -1.0;
                    bool first = true, tie = false;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const uint32_t p = pk[i];
                        const bool act = !(p & k3Dnf);
                        const double t = cum[i];
                        leader = (act && first) ? t : leader;
                        const double nt_fixed = leader + kd * step;                  // :363 / :412
                        const double gap = t - leader;
                        const double nt_vsc = leader + gap * 0.8;                    // :386-387
                        const double nt = vsc ? nt_vsc : nt_fixed;
                        tie |= act && nt == prev_nt;
                        prev_nt = act ? nt : prev_nt;
                        const double tbl = nt - leader;                             // :371 / :388 / :413
                        uint32_t q = (p & ~k3Dirty) | ((tbl > 0 && tbl < dirty_thr) ? k3Dirty : 0u);
                        const uint32_t agef = q & k3AgeMask;
                        q -= agef < dec_unit ? agef : dec_unit;                     // max(0, tire_age - 1), :375 / :393-395
                        const uint32_t q_red = (q & ~(k3CompMask | k3AgeMask)) | red_bits;   // :414-429
                        q = red ? q_red : q;
                        pk[i] = act ? q : p;
                        cum[i] = act ? nt : t;
                        kd = kd + (act ? 1.0 : 0.0);
                        first = first && !act;
                    }
                    drs_disabled_until = lap + (vsc ? 1 : 2);
                    // Re-spacing keeps the running cars' relative order (the only order the lap loop below
                    // needs; the full order is rebuilt after it), and the FIELD ORDER that addresses this lap's
                    // draws stays what the last lap left.  x0.8 is monotone but may round two gaps together, and
                    // equal times must fall back to grid order: re-sort only then (the oracle does the same).
                    if (tie) transposition_sort<N>(cum, pk);
                }
#endif
            }

            // ---- this lap's retirements (:194-197): the driver whose drawn lap this is gets the SIGN of his LAST row set --
            // the lap step reads that row anyway (the car ahead's last lap, :179-183) and takes the sign for "retires now";
            // lap times are positive (reg_time_floor), a row is rewritten by every lap its driver runs and never read once
            // he is out.  Rare per lane (a retirement per race), so the wave branches; the next key comes from the lane's
            // list in device memory.
            while (!(MCGP_SKIP & 512) && MCGP_ANY(((next_out >> 5) & 0x3FFu) == (uint32_t)lap)) {
                if (((next_out >> 5) & 0x3FFu) == (uint32_t)lap) {
                    const uint32_t key = next_out & kNoKey;
                    lds_or_u32(G::oLast + (key & 31u) * (uint32_t)(B * 8) + tid8 + 4u, 0x80000000u);   // (no value comes back: nothing to wait for)
                    next_out = (next_out & 0x80000000u) | (kNoKey << 16) | ((next_out >> 16) & kNoKey);
                    if (next_out == (0x80000000u | (kNoKey << 16) | kNoKey)) {      // both used, more in memory
                        uint32_t tid_w = tid;
                        pin(tid_w);
                        next_out = retirements_after(retire_ws_base + ws_first_lane + tid_w, key);
                    }
                }
            }

            // ---- every running car's lap (:179-223) with its pit stop (:433-494), in time-rank order ----
            // Written without branches: every slot computes its lap and the results are merged by selects, so the
            // LDS gathers of kStepBatch slots stay in flight together (nothing can be sunk into a branch) and
            // the wave does not pay exec-mask bookkeeping per car.  What a retired car "computes" is discarded:
            // +0.0 on its time, its pk kept, and a LAST value nobody reads (only running cars feed `carry`).
            // The lap-noise draws of places 4j .. 4j+3 are the four words of ONE Philox block, computed here between the
            // gathers and their use (it covers their latency) and consumed from registers.
            {
                double fuel = 110.0 - 1.5 * (double)(lap - 1);
                if (!(fuel > 0)) fuel = 0.0;
                const double fuel_effect = (110.0 - fuel) * 0.03;
                const bool pit_window = remaining_laps > 5;                                     // :451
                const uint32_t lut_base = G::oLut + 32u * (uint32_t)pit_regime(remaining_laps);  // this lap's row of the pit rule
                // in a vector register, so that (p & ~age) | retire_word is one v_and_or_b32 (an instruction takes one scalar operand)
                uint32_t retire_word = k3Dnf | ((uint32_t)lap << k3AgeShift);
                if constexpr (reg_min_waves(N) <= 3) pin(retire_word);      // (at 4+ waves per SIMD the register it takes is spilled)
                double carry = 0.0;
#pragma unroll
                for (int i0 = 0; i0 < ((MCGP_SKIP & 64) ? 0 : N); i0 += kStepBatch) {
                    MCGP_SCHED_FENCE();          // one batch at a time: gathers hoisted from later batches cost registers
                    SlotIn in[kStepBatch];
                    if constexpr (!kLean) {
#pragma unroll
                        for (int j = 0; j < kStepBatch; ++j)
                            if (i0 + j < N) in[j] = load_slot(pk[i0 + j], lut_base);
                    }
                    uint32_t w[kStepBatch / 4][4], x[kStepBatch / 4][4];
#pragma unroll
                    for (int b = 0; b < kStepBatch / 4; ++b) {
                        w[b][0] = w[b][1] = w[b][2] = w[b][3] = 0u;
                        x[b][0] = x[b][1] = x[b][2] = x[b][3] = 0u;
                        if (i0 + 4 * b < N) {
                            philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeCar | (uint32_t)((i0 >> 2) + b), k0l, k1l,
                                          w[b][0], w[b][1], w[b][2], w[b][3]);
                            if constexpr (WIDE)
                                philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeCar | kCompanion | (uint32_t)((i0 >> 2) + b), k0l, k1l,
                                              x[b][0], x[b][1], x[b][2], x[b][3]);
                        }
                    }
                    // the lap-noise deviate: binary32 from the cubic table, or (WIDE) binary64 from the degree-7 table
                    using Deviate = std::conditional_t<WIDE, double, float>;
                    Deviate z[kStepBatch];
                    if constexpr (WIDE) {
                        // rows named for the whole batch, then fetched (4 x ds_read_b128 each), then evaluated; a batch with
                        // a deviate whose row is not in LDS (kRareHi) reads all its rows from device memory instead
                        uint32_t zhi[kStepBatch];
                        double zt[kStepBatch];
                        bool rare = G::kNorm53Rows == 0 || MCGP_WIDE_EXACT;
#pragma unroll
                        for (int j = 0; j < kStepBatch; ++j) {
                            normal53_prepare(w[j >> 2][j & 3], x[j >> 2][j & 3], zhi[j], zt[j]);
                            rare |= (i0 + j < N) && zhi[j] < kRareHi;
                        }
                        MCGP_STAT(15, rare);
                        if (MCGP_NOCALLS || __builtin_expect(!MCGP_ANY(rare), 1)) {
                            constexpr int R = kLean ? 1 : 2;                  // table rows fetched together (16 registers each)
#pragma unroll
                            for (int j0 = 0; j0 < kStepBatch; j0 += R) {
                                double zc[R][8];
#pragma unroll
                                for (int j = 0; j < R; ++j) norm53_row(zhi[j0 + j], zc[j]);
#pragma unroll
                                for (int j = 0; j < R; ++j)
                                    z[j0 + j] = (i0 + j0 + j < N) ? normal53_evaluate(w[(j0 + j) >> 2][(j0 + j) & 3], zc[j], zt[j0 + j]) : 0.0;
                            }
                        } else {
                            // One deviate at a time (the eight coefficients of a row are sixteen registers), in line: as an
                            // out-of-line call this path cost the race loop 23 scratch accesses per lap -- the register
                            // allocator saves what lives across a call where it is defined (87.1 -> 85.0 ms).
#pragma unroll
                            for (int j = 0; j < kStepBatch; ++j) {
                                z[j] = (i0 + j < N) ? normal53(w[j >> 2][j & 3], x[j >> 2][j & 3], norm53) : 0.0;
                                pin(z[j]);
                                MCGP_SCHED_FENCE();
                            }
                        }
                        if constexpr (kLean) {
                            // the deviates are done: only now the slots' gathers
#pragma unroll
                            for (int j = 0; j < kStepBatch; ++j) pin(z[j]);
                            MCGP_SCHED_FENCE();
#pragma unroll
                            for (int j = 0; j < kStepBatch; ++j)
                                if (i0 + j < N) in[j] = load_slot(pk[i0 + j], lut_base);
                        }
                    } else {
                        // two table rows are fetched at a time, then evaluated (half the LDS round trips in a row; four at a
                        // time cost sixteen registers the loop does not have)
#pragma unroll
                        for (int j0 = 0; j0 < kStepBatch; j0 += 2) {
                            uint32_t zrow[2];
                            float zt[2];
                            float4 zc[2];
#pragma unroll
                            for (int j = 0; j < 2; ++j) normal_prepare(w[(j0 + j) >> 2][(j0 + j) & 3], zrow[j], zt[j]);
#pragma unroll
                            for (int j = 0; j < 2; ++j) zc[j] = norm_row(zrow[j]);
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                z[j0 + j] = (i0 + j0 + j < N) ? normal_evaluate(w[(j0 + j) >> 2][(j0 + j) & 3], zc[j], zt[j]) : 0.0f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < kStepBatch; ++j) {
                        const int i = i0 + j;
                        if (i >= N) continue;
                        const SlotIn &s = in[j];
                        const uint32_t p = pk[i];
                        const bool active = !(p & k3Dnf);                   // running at the start of the lap
                        // retires on this lap (:194-197): the sign of the driver's LAST row, set above
                        const bool dnf_hit = __double2hiint(s.last) < 0;
                        const bool run = active && !dnf_hit;
                        const double ahead_last = carry;                    // last lap of the running car ahead  :179-183 (sign: see above)
                        carry = active ? s.last : carry;
                        const uint32_t agef = p & k3AgeMask;
                        const double tire = (double)agef * s.eff;                                   // :319-322 (eff x 2^-16)
                        const double drs_gain = s.drs;                                              // :327
                        // :330.  np.random.normal(0, sigma) is 0.0 + sigma z: the addition only turns a -0.0 product into +0.0,
                        // which the sum below cannot tell apart unless its other terms cancel exactly -- a lap time of pure
                        // noise, which reg_time_floor() keeps out of this kernel
                        const double noise = s.var * (double)z[j];
                        const double clean = s.base + tire - fuel_effect + s.cdelta - drs_gain + noise;   // :332
                        // :207-215, without a select.  A car flagged dirty is not the first running car (its gap to the
                        // leader is > 0: _update_positions and every event handler keep flag and order consistent), and from lap
                        // 3 on the running car ahead of it has driven the lap before, so its last lap time is a lap time: > 0
                        // (reg_time_floor), and :211's  car_ahead_lap > 0  is the flag itself.  In lap 2 no car has a last lap
                        // time yet (lap 1 does not record one, :219 is the only writer) and the constraint is off for everybody:
                        // that lap runs with a penalty of +0.0 and, ahead = 0, a maximum that returns the clean time.
                        // Masked by the flag, both terms of :213-215 vanish for a clean-air car:  max(|0|, clean + 0.0) = clean
                        // (clean > 0).
                        const uint32_t dirty_m = (uint32_t)((int32_t)(p << (31 - 13)) >> 31);                  // all ones iff k3Dirty
                        const double pen_if_dirty = __hiloint2double((int)((uint32_t)__double2hiint(dirty_pen_lap) & dirty_m),
                                                                     (int)((uint32_t)__double2loint(dirty_pen_lap) & dirty_m));
                        const double ahead_if_dirty = __hiloint2double((int)((uint32_t)__double2hiint(ahead_last) & dirty_m),
                                                                       (int)((uint32_t)__double2loint(ahead_last) & dirty_m));
                        const double lap_time = max_abs_f64(ahead_if_dirty, clean + pen_if_dirty);          // :213, :215 (neither is NaN)
                        // pit stop (:450-492): (tyre age + 1) << 16 against the pit word; compound and used-set from
                        // the rule table (pit_rule_word)
                        const bool pit = run && pit_window && agef >= s.pitw;       // age + 1 > threshold (both fields << 16)
                        const uint32_t p_pit = (p & ~(k3CompMask | k3UsedMask | k3AgeMask)) | s.fit;
                        uint32_t p_run = pit ? p_pit : p + (1u << k3AgeShift);
                        uint32_t p_ret = (p & ~k3AgeMask) | retire_word;
                        pin(p_run);              // both computed for every lane: the merge below stays a pair of selects
                        pin(p_ret);
                        const uint32_t p_act = dnf_hit ? p_ret : p_run;
                        pk[i] = active ? p_act : p;
                        // :218, :464 as two additions for every car: the lap time or +0.0 (not running), then the stop's time or
                        // +0.0 from a two-entry LDS table (an offset select and a read instead of two 64-bit-encoded selects).
                        // x + 0.0 is x for every x but -0.0, which no cumulative time is.
                        const double lap_add = run ? lap_time : 0.0;
                        const double pit_add = lds_ld<double>(G::oPit + (pit ? 8u : 0u));           // (pit implies run)
                        cum[i] = (cum[i] + lap_add) + pit_add;
                        pin(cum[i]);             // done HERE: not sunk, with its masks and lap time, to where it is used
                        lds_st<double>(G::oLast + s.la, lap_time);                                  // :219
                    }
                }
            }

            // ---- _simulate_overtakes, :496-536 ----
            bool distinct_times = true;
            if (!(MCGP_SKIP & 16)) distinct_times = network_sort<N>(cum, pk);
            if (MCGP_DUP & 1) network_sort<N>(cum, pk);
            // The three passes are three copies of the code: a rolled loop makes the compiler shuffle the whole field
            // (60 registers, renamed by every compare-exchange) back into place at its back edge; unrolled it is 3-4 %
            // faster at every field size although the lap body no longer fits a 64 KB instruction cache.
#pragma unroll
            for (int pass = 0; pass < ((MCGP_SKIP & 1) ? 0 : 3); ++pass) {
                // ---- overtakes: pace deltas and candidates ----
                // pace of every slot (:514-515), the pace delta of every adjacent pair, and what an attempt needs to
                // succeed.  Everything is scaled by 2^31 (tables, reg_load_tables): u < min(0.5, delta / 2) for the
                // uniform u = w 2^-32 is  w < 2^31  and  w < delta 2^31,  i.e.  w < thr = min(ceil(delta 2^31), 2^31)
                // -- one integer per pair instead of a binary64 delta kept across the draw-word generation.
                uint32_t thr[N];
                uint32_t ow[N];                  // ow[i] = the draw word of the attempt at pair i
                uint32_t n_attempts = 0u;        // (statistics of the host build)
                // WIDE (reference-width draws): u < min(0.5, delta / 2) for u = q / 2^53, q = w 2^21 + (21 bits of the companion
                // word), is  q < T = min(ceil(dl 2^21), 2^52)  (the pace deltas carry 2^31).  T lies in [floor(dl) 2^21,
                // (floor(dl) + 1) 2^21], so the draw's word alone decides -- a success iff w < floor(dl), none iff w > floor(dl)
                // -- unless w == floor(dl), one attempt in 2^32: the pass runs the default code with thr = min(floor(dl), 2^31)
                // (ovt_threshold without the ceiling), and a wave with such an attempt (or with more than eight attempts in
                // a lane) decides the whole pass with the exact 53-bit code below, companion blocks and all.
                [[maybe_unused]] auto wide_pass_exact = [&]() -> uint32_t {
                    if constexpr (kLean) {
                        PkWords<N> f;
#pragma unroll
                        for (int i = 0; i < N; ++i) f.v[i] = pk[i];
                        return wide_pass_exact_fn<N, G>(f, od31, (uint32_t)lap, (uint32_t)pass, c0l, c1l, k0l, k1l, tid);
                    } else {
                        return wide_pass_exact_body<N, G>(pk, od31, (uint32_t)lap, (uint32_t)pass, c0l, c1l, k0l, k1l, tid);
                    }
                };
                {
                    // ---- overtakes: draw words ----
                    // the k-th attempt of this pass reads word k & 3 of block 8 * pass + k / 4.  The first eight -- all that
                    // all but a few wave-passes in a thousand need -- are drawn FIRST, into the eight rows of the W plane: the
                    // stage below then fetches a pair's word the moment its row is known, and no array of addresses or of
                    // candidate bits has to live across the Philox blocks (which cost a dozen registers themselves).
                    {
                        uint32_t o0, o1, o2, o3;
                        philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeOvt | (uint32_t)(8 * pass), k0l, k1l, o0, o1, o2, o3);
                        lds_st<uint32_t>(w_row(0), o0);
                        lds_st<uint32_t>(w_row(1), o1);
                        lds_st<uint32_t>(w_row(2), o2);
                        lds_st<uint32_t>(w_row(3), o3);
                        philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeOvt | (uint32_t)(8 * pass + 1), k0l, k1l, o0, o1, o2, o3);
                        lds_st<uint32_t>(w_row(4), o0);
                        lds_st<uint32_t>(w_row(5), o1);
                        lds_st<uint32_t>(w_row(6), o2);
                        lds_st<uint32_t>(w_row(7), o3);
                    }
                    // W-plane address of the NEXT attempt's word: a running sum over the candidates so far (row k of the plane
                    // holds the word of the lane's k-th attempt), advanced inside ovt_threshold by the compare that makes
                    // the pair a candidate
                    uint32_t row = tid4;
                    {
                        constexpr int H = kPaceBatch;
                            double pace_prev = 0.0;
#pragma unroll
                        for (int h = 0; h < N; h += H) {
                            MCGP_SCHED_FENCE();
                            double pb[H], pd[H], pa[H];
#pragma unroll
                            for (int j = 0; j < H; ++j) {
                                if (h + j < N) {
                                    const uint32_t id16 = pk_slot16(pk[h + j]);           // 16 x (32 dnf + driver)
                                    const f64x2 bd = lds_ld_f64x2(G::oDrvB + id16);
                                    pb[j] = bd.x;
                                    pd[j] = bd.y;
                                    pa[j] = lds_ld<double>(G::oDrsB + (pk[h + j] & k3Drs));
                                }
                            }
#pragma unroll
                            for (int j = 0; j < H; ++j) {
                                const int i = h + j;
                                if (i < N) {
                                    // a retired car's pace is NaN (table): its two pairs compare false below (:511)
                                    const double pace = pb[j] + (double)(pk[i] & k3AgeMask) * pd[j];
                                    if (i > 0) {
                                        const double dl = (pace_prev - pace) + pa[j];                   // :516, :519-520 (x 2^31)
                                        // candidate iff dl > overtake_delta (:522); threshold min(ceil(dl), 2^31) (:523-524)
                                        uint32_t next;
                                        ovt_threshold<(uint32_t)(B * 4), !WIDE>(dl, od31, row, thr[i], next);
                                        // (a pair that is no candidate fetches the word of the next attempt, or one just past
                                        //  the plane: its threshold is 0)
                                        ow[i] = lds_ld<uint32_t>(G::oW + row);
                                        row = next;
                                    }
                                    pace_prev = pace;
                                }
                            }
                        }
                        thr[0] = 0u;
                        ow[0] = 0u;
                    }
                    const uint32_t words_end = row - tid4;          // = attempts of this lane x the row stride
                    MCGP_STAT(0 + pass, words_end != 0u);
                    if (words_end == 0u) break;
                    n_attempts = words_end / (uint32_t)(B * 4);
                    // A wave with a lane that has more than eight attempts takes the general path, 8 attempts at a time, and
                    // leaves its verdicts in the same two arrays (ow = 0, thr = 1 for a success).
                    if constexpr (WIDE) {
                        bool tie = false;
#pragma unroll
                        for (int i = 1; i < N; ++i) tie |= ((ow[i] ^ thr[i]) >> MCGP_WIDE_TIE_SHIFT) == 0u;
                        MCGP_STAT(11, tie);
                        if (!MCGP_NOCALLS && __builtin_expect(MCGP_ANY(tie || MCGP_WIDE_EXACT || words_end > (uint32_t)(kWordRows * B * 4)), 0)) {
                            const uint32_t hits = wide_pass_exact();
#pragma unroll
                            for (int i = 1; i < N; ++i) {
                                thr[i] = (hits >> i) & 1u;
                                ow[i] = 0u;
                            }
                        }
                    } else if (__builtin_expect(MCGP_ANY(words_end > (uint32_t)(kWordRows * B * 4)), 0)) {
                        // the candidate mask: a candidate's threshold is at least 1 (dl > overtake_delta >= 0: reg_kernel_serves
                        // sends a negative delta -- attempts at a pace DEFICIT -- to the generic kernel)
                        uint32_t cand = 0u;
#pragma unroll
                        for (int i = 1; i < N; ++i) cand |= thr[i] != 0u ? (1u << i) : 0u;
                        uint32_t hits = 0u, rest = cand;
#pragma unroll 1
                        for (int chunk = 0; rest != 0u; ++chunk) {
                            uint32_t m = rest;                                  // `rest` without its 8 lowest set bits
#pragma unroll 1
                            for (int k = 0; k < kWordRows && m != 0u; ++k) m &= m - 1u;
                            const uint32_t cur = rest & ~m;                     // attempts 8 chunk .. 8 chunk + 7
#pragma unroll 1
                            for (int b = 0; b < 2; ++b) {
                                uint32_t o0, o1, o2, o3;
                                philox4x32_10(c0l, c1l, (uint32_t)lap, kPurposeOvt | (uint32_t)(8 * pass + 2 * chunk + b), k0l,
                                              k1l, o0, o1, o2, o3);
                                lds_st<uint32_t>(w_row(4 * b + 0), o0);
                                lds_st<uint32_t>(w_row(4 * b + 1), o1);
                                lds_st<uint32_t>(w_row(4 * b + 2), o2);
                                lds_st<uint32_t>(w_row(4 * b + 3), o3);
                            }
                            uint32_t h = 0u;
#pragma unroll
                            for (int i = 1; i < N; ++i) {
                                // (a pair outside `cur` reads some other word, or just past the plane: masked out below)
                                const uint32_t word = lds_ld<uint32_t>(G::oW + (uint32_t)__popc(cur & ((1u << i) - 1u)) * (uint32_t)(B * 4) + tid4);
                                h |= word < thr[i] ? (1u << i) : 0u;
                            }
                            hits |= h & cur;
                            rest = m;
                        }
#pragma unroll
                        for (int i = 1; i < N; ++i) {
                            thr[i] = (hits >> i) & 1u;
                            ow[i] = 0u;
                        }
                    }
                }
                // ---- overtakes: success test and write-back chain ----
                // :523-531 in sorted order, each pair seeing the previous pair's mutation (Q15).  Branch-free: the
                // new times of a pair are computed for every slot and committed by selects under the success mask.
                bool any_succ = false;
#pragma unroll
                for (int i = 1; i < N; ++i) {
                    const bool hit = ow[i] < thr[i];
                    const double nb = cum[i - 1] - 0.1;                        // max(0.1, ahead - 0.1), :528: reg_time_floor()
                    const double na = nb + 0.3;                                // :530
                    cum[i] = hit ? nb : cum[i];
                    cum[i - 1] = hit ? na : cum[i - 1];
                    any_succ |= hit;
                }
                MCGP_STAT(4 + pass, any_succ);
                MCGP_STAT(8, n_attempts);
                MCGP_TRACE_PASS(local, lap, pass, (int)n_attempts);
                (void)n_attempts;
                if (!any_succ) break;
                // ---- overtakes: re-sort ----
                distinct_times = resort_after_overtakes<N>(cum, pk);     // sorted again for the next pass / _update_positions
                if (MCGP_DUP & 8) resort_after_overtakes<N>(cum, pk);
            }
            // ---- _update_positions, :227-228 ----
            if (!(MCGP_SKIP & 32)) {
                if (MCGP_DISTINCT_PATH && !MCGP_ANY(!distinct_times))
                    update_positions_reg<N, true>(cum, pk, lap > 2 && lap > drs_disabled_until, dirty_thr);
                else
                    update_positions_reg<N>(cum, pk, lap > 2 && lap > drs_disabled_until, dirty_thr);
            }
            if (MCGP_DUP & 4) update_positions_reg<N>(cum, pk, lap > 2 && lap > drs_disabled_until, dirty_thr);
        }

        // ================= classification, reference :230-242 =================
        // Running cars by time, then retired cars by (lap, time) descending, stable.  The field is in (time, grid
        // slot) order, so all of that is one integer per car: a running car's key is its rank; a retired car's is
        //     1 << 26 | (2047 - lap) << 15 | (31 - g) << 10 | grid slot << 5,     g = index of its time among the
        // distinct times in ascending order (equal times share g and fall back to grid order, the reference's
        // stable tie; lap-1 retirements carry distinct negative times, see lap 1).  The driver rides in the low 5 bits.
        // The keys are sorted in registers by the lap loop's network (a comparator on integers is a v_min_u32 and a
        // v_max_u32); the insertion sort in LDS that used to stand here ran a data-dependent loop per car, as long as the
        // wave's slowest lane needed -- every retirement of a wave's 64 races stretched it for all of them.
        uint32_t key[N];
        {
            uint32_t g = 0u;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (i > 0) g += cum[i] != cum[i - 1] ? 1u : 0u;
                const uint32_t p = pk[i];
                const uint32_t id = (p >> k3IdShift) & 31u;
                const uint32_t lapf = (p >> k3AgeShift) & 0x7FFu;
                const uint32_t key_dnf = (1u << 26) | ((2047u - lapf) << 15) | ((31u - g) << 10) | ((p >> k3GposShift) << 5) | id;
                const uint32_t key_run = ((uint32_t)i << 5) | id;
                key[i] = (p & k3Dnf) ? key_dnf : key_run;
            }
        }
        network_sort_keys<N>(key);
        // finishing order: N bytes per simulation, contiguous per lane.  Four positions are packed into one dword
        // store when the lane's N-byte record is dword aligned (N % 4 == 0 and an aligned buffer): N / 4 stores per
        // lane that the L2 merges into full lines, instead of N single-byte stores.
        if (live) {
#pragma unroll
            for (int p = 0; p < N; ++p) atomicAdd(&s_hist[(key[p] & 31u) * N + p], 1u);              // reference :93-94
            if (orders) {
                const bool dword_orders = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(orders) & 3u) == 0);
                if (dword_orders) {
#pragma unroll
                    for (int p = 0; p + 3 < N; p += 4)
                        *reinterpret_cast<uint32_t *>(orders + local * (uint64_t)N + (uint64_t)p) =
                            (key[p] & 31u) | ((key[p + 1] & 31u) << 8) | ((key[p + 2] & 31u) << 16) | ((key[p + 3] & 31u) << 24);
                } else {
#pragma unroll
                    for (int p = 0; p < N; ++p) orders[local * (uint64_t)N + (uint64_t)p] = (uint8_t)(key[p] & 31u);
                }
            }
        }
    }
}

// WAVES: the block shape.  The default is the one measured best for the field size (reg_block_waves); every field size is
// also compiled in blocks of kSmallBlockWaves waves, whose LDS footprint is less than half a CU's: the shape the library
// falls back to when the default block does not fit what the device (or a runtime that keeps some LDS to itself) offers.
constexpr int kSmallBlockWaves = 4;
template <int N, int WAVES = reg_block_waves(N)>
__global__ void __launch_bounds__((RegGeo<N, WAVES>::B), reg_min_waves(N))
race_kernel_reg(const KParams *__restrict__ P, uint64_t n_sims, uint64_t sim_offset,
                uint32_t seed_lo, uint32_t seed_hi, unsigned long long *__restrict__ hist,
                uint8_t *__restrict__ orders, const uint8_t *__restrict__ fixed_grid, uint32_t n_chunks,
                uint32_t *__restrict__ ticket, uint32_t *__restrict__ retire_ws)
{
    using G = RegGeo<N, WAVES>;
    extern __shared__ __align__(16) unsigned char smem[];
    // host and kernel must agree on the geometry, and the rows are addressed by absolute LDS address
    // (race_isa.hip.h): a violation aborts the launch (a HIP error at the next synchronisation), never a silent result
    if ((int)blockDim.x != G::B || lds_base_of(smem) != 0u) __builtin_trap();
    reg_load_tables<N, G>(P, smem, threadIdx.x);
    __syncthreads();
    // retirement lists: one column per lane of the launch, N + 1 rows (reg_retire_ws_bytes)
    reg_simulate<N, false, G>(P, smem, threadIdx.x, ticket, n_sims, sim_offset, seed_lo, seed_hi, orders, fixed_grid, n_chunks,
                              retire_ws, (uint32_t)(gridDim.x * G::B), (uint32_t)(blockIdx.x * G::B));
    __syncthreads();
    reg_flush_hist<N, G>(smem, threadIdx.x, hist);
}

// The reference-width build (mcgp_config.deviates = MCGP_DEVIATES_53): same phases, blocks of 8 waves at 2 waves per SIMD.
template <int N>
__global__ void __launch_bounds__((WideGeo<N>::B), wide_min_waves(N))
race_kernel_reg_wide(const KParams *__restrict__ P, uint64_t n_sims, uint64_t sim_offset,
                     uint32_t seed_lo, uint32_t seed_hi, unsigned long long *__restrict__ hist,
                     uint8_t *__restrict__ orders, const uint8_t *__restrict__ fixed_grid, uint32_t n_chunks,
                     uint32_t *__restrict__ ticket, uint32_t *__restrict__ retire_ws, const double *__restrict__ norm53)
{
    using G = WideGeo<N>;
    extern __shared__ __align__(16) unsigned char smem[];
    if ((int)blockDim.x != G::B || lds_base_of(smem) != 0u) __builtin_trap();
    reg_load_tables<N, G>(P, smem, threadIdx.x, norm53);
    __syncthreads();
    reg_simulate<N, true, G>(P, smem, threadIdx.x, ticket, n_sims, sim_offset, seed_lo, seed_hi, orders, fixed_grid, n_chunks,
                             retire_ws, (uint32_t)(gridDim.x * G::B), (uint32_t)(blockIdx.x * G::B), norm53);
    __syncthreads();
    reg_flush_hist<N, G>(smem, threadIdx.x, hist);
}

// ---- several problems in one launch (mcgp_run_batch) ----
// The reference predicts a race from 10 000 simulations (reference src/predictor.py:284) and a backtest runs two dozen
// races one after the other (src/validation.py:179-185): at that size a launch is as long as ONE race of one lane and the
// device is mostly idle, so the races of a sweep go into one launch.  The block-shared tables belong to one problem, so a
// BLOCK works on one problem at a time; inside it every WAVE claims chunks from that problem's own ticket counter, as in
// a single-problem launch, until the problem has none left.  The block then moves on to the next problem (in cyclic order
// from where it started) that still has chunks to hand out, reloads the tables and goes on; blocks start spread evenly
// over the problems.  Nothing waits for a whole group of chunks: the second round of a problem whose chunks do not fill
// its blocks' waves evenly runs on the few waves that come free first, with the SIMDs nearly to themselves.
// Per-problem inputs that are not in the parameter block:
struct BatchItem {
    uint64_t sim_offset, seed;
};
// One problem of the batch on this block: tables, the block's share of the problem's chunks, histogram flush.  A function
// of its own (MCGP_BATCH_CALL=1) so that the register allocator sees the race loop as it sees it in the single-problem kernel.
#ifndef MCGP_BATCH_CALL
#define MCGP_BATCH_CALL 0
#endif
template <int N>
#if MCGP_BATCH_CALL
__device__ __attribute__((noinline))
#else
__device__ __forceinline__
#endif
void batch_problem(const KParams *__restrict__ P_, unsigned char *smem_, uint32_t *__restrict__ ticket_, uint64_t n_sims_,
                   uint64_t sim_offset_, uint64_t seed_, uint32_t n_chunks_, uint32_t *__restrict__ retire_ws_,
                   unsigned long long *__restrict__ hist_)
{
    using G = RegGeo<N>;
#if MCGP_BATCH_CALL
    const KParams *__restrict__ P = uniform_ptr(P_);
    unsigned char *smem = uniform_ptr(smem_);
    uint32_t *__restrict__ ticket = uniform_ptr(ticket_);
    uint32_t *__restrict__ retire_ws = uniform_ptr(retire_ws_);
    unsigned long long *__restrict__ hist = uniform_ptr(hist_);
    const uint64_t n_sims = uniform_u64(n_sims_), sim_offset = uniform_u64(sim_offset_), seed = uniform_u64(seed_);
    const uint32_t n_chunks = uniform_u32(n_chunks_);
#else
    const KParams *__restrict__ P = P_;
    unsigned char *smem = smem_;
    uint32_t *__restrict__ ticket = ticket_;
    uint32_t *__restrict__ retire_ws = retire_ws_;
    unsigned long long *__restrict__ hist = hist_;
    const uint64_t n_sims = n_sims_, sim_offset = sim_offset_, seed = seed_;
    const uint32_t n_chunks = n_chunks_;
#endif
    reg_load_tables<N>(P, smem, threadIdx.x);                    // (zeroes the LDS histogram)
    __syncthreads();
    reg_simulate<N>(P, smem, threadIdx.x, ticket, n_sims, sim_offset, (uint32_t)seed, (uint32_t)(seed >> 32),
                    nullptr, nullptr, n_chunks, retire_ws, (uint32_t)(gridDim.x * G::B), (uint32_t)(blockIdx.x * G::B));
    __syncthreads();                                             // every wave's counts are in
    reg_flush_hist<N>(smem, threadIdx.x, hist);
    __syncthreads();                                             // (the flush reads what the next table load zeroes)
}
template <int N>
__global__ void __launch_bounds__(RegGeo<N>::B, reg_min_waves(N))
race_kernel_reg_batch(const KParams *__restrict__ P, const BatchItem *__restrict__ items, uint32_t n_problems,
                      uint64_t n_sims, unsigned long long *__restrict__ hist, uint32_t n_chunks,
                      uint32_t *__restrict__ tickets, uint32_t *__restrict__ retire_ws)
{
    using G = RegGeo<N>;
    extern __shared__ __align__(16) unsigned char smem[];
    if ((int)blockDim.x != G::B || lds_base_of(smem) != 0u) __builtin_trap();
    // first problem of this block; `visited` problems have been looked at (each at most once: the loop ends)
    uint32_t p = (uint32_t)(((uint64_t)blockIdx.x * n_problems) / gridDim.x);
    uint32_t visited = 0u;
    for (;;) {
        // wave 0 looks for the next problem with chunks left, 64 counters at a time, and leaves it (or "none") in the word
        // just past the kernel's LDS map
        if (threadIdx.x < 64u) {
            uint32_t found = 0xFFFFFFFFu, seen = visited;
            while (seen < n_problems) {
                const uint32_t k = seen + threadIdx.x;                         // k-th problem after the start, cyclically
                uint32_t q = p + (k - visited);
                if (q >= n_problems) q -= n_problems;
                const uint32_t lane = first_lane_with(k < n_problems && peek_ticket(&tickets[q]) < n_chunks);
                if (lane < 64u) {
                    found = seen + lane;
                    break;
                }
                seen += 64u;
            }
            if (threadIdx.x == 0u) lds_st<uint32_t>(G::kBytes, found);
        }
        __syncthreads();
        const uint32_t found = lds_ld<uint32_t>(G::kBytes);
        if (found == 0xFFFFFFFFu) break;
        p += found - visited;
        if (p >= n_problems) p -= n_problems;
        visited = found + 1u;
        const BatchItem it = items[p];
        batch_problem<N>(P + p, smem, tickets + p, n_sims, it.sim_offset, it.seed, n_chunks, retire_ws, hist + (size_t)p * N * N);
        p += 1u;
        if (p >= n_problems) p -= n_problems;
    }
}
constexpr size_t kBatchLdsExtra = 16;                            // the word that names the block's next problem

}  // namespace mcgp
