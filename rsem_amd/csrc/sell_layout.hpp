// sell_layout.hpp -- the "sliced" device layout shared by the EM and Gibbs kernels.
//
// Reads (CSR rows) are radix-sorted on the device by (shape, min sid, hash of the sid tuple), so
// reads hitting the same transcript set become neighbours, and packed into 64-lane slices:
// G = 2^lg lanes per read (G = 1 for reads with <= 4 alignments, up to 64 for <= 256), K <= 4 planes
// of 64 entries per slice; lane l of plane k holds alignment k*G + (l % G) of the read in row slot
// l / G of the slice.  Slices are grouped in BLOCKS of T consecutive slices = one wave's work; inside
// a block the sorted reads are laid out LANE-MAJOR: row slot r walks T consecutive sorted reads over
// the block's T slices.  A lane therefore sees long runs of reads with the identical sid tuple and
// can keep their partial counts in registers, while every global load stays a fully coalesced
// 256 B (sid) / 512 B (conprb) wave access.  A 64-bit mask per slice tells which lanes start a new
// tuple there.  Reads with > 256 alignments stay in the caller's CSR ("long rows").
//
// Value planes come in two formats, chosen per read and therefore per shape (a shape = (format, G, K)):
//   F64  the caller's doubles, 512 B per plane;
//   Q32  block floating point: a 32-bit unsigned mantissa per alignment (256 B per plane) and one int16 exponent e per
//        read, value = m * 2^e with 2^e chosen so that the read's largest value lands in [2^31, 2^32), m rounded to
//        nearest.  Only reads whose non-zero values all lie within a factor 2^range_bits of their largest one take
//        this format (every value keeps >= 32 - range_bits significant bits); the rest stay F64 in shapes of their
//        own.  m * 2^e is exact in a double, so a kernel reading Q32 planes computes bit for bit what the F64 kernel
//        computes on the rounded values (tests pin that; quantize_q32 in tests/ is the numpy statement of the rule).
// Included by em.hip and gibbs.hip (each TU gets its own copy of the kernels).
#pragma once
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "common.hpp"

namespace {

#include "sell_shape.hpp"

constexpr int kShapesPerFmt = 28;   // (lg 0..6) x (K 1..4)
constexpr int kShapeBits = 7;
constexpr int kMaxShapes = 3 * kShapesPerFmt;  // F64 shapes, then Q32 shapes, then F64X shapes (split rows)
constexpr int kShapeIds = 1 << kShapeBits;
constexpr int kLongShape = kShapeIds - 1;      // rows with more than 256 alignments: CSR kernel
constexpr uint32_t kKeyMinSidCap = (1u << (31 - kShapeBits)) - 1;  // sort key: shape | apart bit | anchor sid (capped) | hash of the tuple
constexpr int kKeyApartBit = 63 - kShapeBits;
constexpr int kMaxK = 4;
constexpr int kBlock = 256;         // 4 waves

// ---- Q32 quantisation rule (one read) ----------------------------------------------------------
// mx = the read's largest value.  Returns false when the read must stay F64: no positive value, a non-zero value
// below mx * 2^-range_bits, or an exponent for which 2^e / value * 2^-e would leave the normal doubles.
struct Q32Scale { int e; };
__host__ __device__ inline bool q32_scale_of(double mx, double mn_nonzero, int range_bits, Q32Scale& q) {
    if (!(mx > 0.0) || !(mx < 1e300)) return false;
    int ex;
    (void)frexp(mx, &ex);              // mx = f * 2^ex, f in [0.5, 1)
    q.e = ex - 32;                     // mx / 2^e in [2^31, 2^32)
    if (q.e < -1000 || q.e > 900) return false;
    return mn_nonzero >= ldexp(mx, -range_bits);
}
__host__ __device__ inline uint32_t q32_mantissa(double v, int e) {
    const double m = rint(ldexp(v, -e));      // exact scaling, round to nearest even
    return m >= 4294967295.0 ? 0xffffffffu : (uint32_t)m;
}

__host__ __device__ inline int shape_id_of(uint64_t L) {
    if (L <= 4) return (int)(L == 0 ? 0 : L - 1);  // lg = 0, K = L
    int lg = 1;
    uint64_t cap = 8;
    while (L > cap) { cap <<= 1; ++lg; }
    if (lg > 6) return kLongShape;
    int K = (int)((L + (1u << lg) - 1) >> lg);  // 3 or 4
    return lg * 4 + (K - 1);
}

// ... and back: (slice within the shape, row slot within the slice) -> sorted read q of the shape.  false: the slot is empty
// (the last block of a shape is shorter).
__host__ __device__ inline bool slot_to_row(const Shape& S, uint32_t T, uint32_t slice_local, uint32_t r, uint32_t& q) {
    const uint32_t R = shape_R(S), rpb = R * T;
    const uint32_t b = slice_local / T, t = slice_local % T;
    if ((uint64_t)b * rpb >= S.n_rows) return false;
    const uint32_t left = S.n_rows - b * rpb;
    const uint32_t nb = left < rpb ? left : rpb;
    const uint32_t Tb = (nb + R - 1) / R;
    if (t >= Tb || r * Tb + t >= nb) return false;
    q = b * rpb + r * Tb + t;
    return true;
}

__host__ __device__ inline uint32_t mix32(uint32_t h, uint32_t v) {
    h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2);
    return h;
}

// ---- layout construction ---------------------------------------------------------------------

// The per-thread bodies of the construction kernels are host-callable functions: tests/simt_emu.hpp builds the layout on
// the CPU with these very functions.
//
// sort key of read i = shape | anchor sid (capped) | hash of the sid tuple.  The anchor says where the read's LDS window
// should start: the smallest sid of the read that lies within kAnchorReach below its MEDIAN sid.  For a read whose
// alignments all sit in one gene that is its smallest sid (the key of rounds 1-2); for a read that ALSO hits a few
// transcripts of a far-away gene (paralogs, cross-gene multi-mappers) it is still its own gene's first isoform, wherever
// the foreign ids lie -- with the plain minimum half of those reads sorted into the FOREIGN gene's neighbourhood and sent
// all their own gene's counts through global atomics (configs[2] with 10 % such reads: 2.05 ms per E step instead of 0.97,
// profiles/r03d_bench_default.json).  A layout hint only: ids outside a unit's window take the global path, whatever the key.
// `apart` (0: off; else the reach, kLayoutWindow in the product): reads with an id outside [anchor, anchor + apart) sort BEHIND all the others of their shape (the apart
// bit).  Such a read has a tuple of its own, so wherever it sits the lane's run of equal tuples ends, twice (at it and after
// it), and its slice -- all 64 lanes -- loads the sid planes; mixed in at 10 % they leave hardly a slice without (16 reads
// per slice: 0.9^16), which is what configs[2] with such reads paid for (profiles/r03m).  Apart, the compact reads keep
// their runs and only the slices of the others pay.
// cp != nullptr: reads that qualify (q32_scale_of) go to the Q32 twin of their shape.  *err: 1 row_ptr not monotone,
// 2 sid outside 1..M.
constexpr int kLayoutWindow = 2048;               // ids per LDS window of the kernels that walk this layout (em.hip kWindow, gibbs.hip kGWindow)
constexpr int kAnchorReach = kLayoutWindow / 2;
constexpr int kMedianExactMax = 64; // reads up to this length: exact median (rank selection); longer: the middle position
// `split` (with apart): a read with an id outside [anchor, anchor + apart) is SPLIT -- its ids inside that range form a row of
// an F64X shape (length, sort key and tuple hash are those of the in-window part alone, so it sits among reads of its own
// gene that share that part), the others go to the far-entry side arrays (sell_build_far).  No apart bit then: the row no
// longer holds a foreign id.
__host__ __device__ inline uint64_t row_key_of(uint64_t i, int32_t M, const uint64_t* __restrict__ row_ptr,
                                               const int32_t* __restrict__ sid, const double* __restrict__ cp, int range_bits,
                                               int apart, int* err, int split = 0, uint32_t* split_far = nullptr) {
    uint64_t fr = row_ptr[i], to = row_ptr[i + 1];
    if (to < fr) { *err = 1; return 0; }
    uint32_t h = 0x811c9dc5u, mn = 0xffffffffu;
    double vmx = 0.0, vmn = 1.79e308;
    for (uint64_t j = fr; j < to; j++) {
        int32_t s = sid[j];
        if (s < 1 || s > M) { *err = 2; s = 1; }
        h = mix32(h, (uint32_t)s);
        mn = (uint32_t)s < mn ? (uint32_t)s : mn;
        if (cp) {
            const double v = cp[j];
            if (!(v >= 0.0)) vmx = 1e308;  // negative / NaN: never compressed
            vmx = fmax(vmx, v);
            if (v > 0.0) vmn = fmin(vmn, v);
        }
    }
    const uint64_t L = to - fr;
    if (L > 1 && !*err) {
        uint32_t med = (uint32_t)sid[fr + L / 2];
        if (L <= (uint64_t)kMedianExactMax) {
            for (uint64_t j = fr; j < to; j++) {  // the sid with <= L/2 smaller ones and > L/2 smaller-or-equal ones
                const uint32_t v = (uint32_t)sid[j];
                uint32_t less = 0, leq = 0;
                for (uint64_t k = fr; k < to; k++) {
                    const uint32_t u = (uint32_t)sid[k];
                    less += u < v ? 1u : 0u;
                    leq += u <= v ? 1u : 0u;
                }
                if (less <= L / 2 && L / 2 < leq) { med = v; break; }
            }
        }
        const uint32_t floor_sid = med > (uint32_t)kAnchorReach ? med - (uint32_t)kAnchorReach : 0u;
        uint32_t lo = med;
        for (uint64_t j = fr; j < to; j++) {
            const uint32_t v = (uint32_t)sid[j];
            if (v >= floor_sid && v < lo) lo = v;
        }
        mn = lo;
    }
    uint64_t far = 0, L_in = 0;
    uint32_t h_in = 0x811c9dc5u;
    if (apart && L > 1 && !*err)
        for (uint64_t j = fr; j < to; j++) {
            const uint32_t v = (uint32_t)sid[j];
            if (v < mn || v >= mn + (uint32_t)apart) far = 1;
            else { ++L_in; h_in = mix32(h_in, v); }
        }
    int shape = shape_id_of(to - fr);
    // Split only where the far alignments are at least half of the read (a read without a gene to speak of: each of its
    // alignments would otherwise be a global atomic per round).  A read of a gene that ALSO hits a few transcripts elsewhere
    // stays whole: as a row of its own shape its tuple runs are too short for the lane kernel to skip anything (a wave
    // re-reads its ids whenever ANY of its 64 lanes starts a new tuple), so the split only added the two side passes
    // (configs[2] with 10 % such reads: 1.48 ms split against 1.17-1.25 whole, profiles/r04d_call.log).
    // split == 2: every read with an id outside splits (the X units then run beside the compact ones: em.hip launch_estep)
    if (far && split && (split == 2 || 2 * L_in <= L) && shape != kLongShape && (uint64_t)mn <= kKeyMinSidCap) {  // split row: keyed by its in-window part
        shape = shape_id_of(L_in) + 2 * kShapesPerFmt;
        if (split_far) *split_far = (uint32_t)(L - L_in);
        return ((uint64_t)shape << (64 - kShapeBits)) | ((uint64_t)mn << 32) | h_in;
    }
    if (mn > kKeyMinSidCap) mn = kKeyMinSidCap;
    Q32Scale q;
    if (cp && shape != kLongShape && q32_scale_of(vmx, vmn, range_bits, q)) shape += kShapesPerFmt;
    return ((uint64_t)shape << (64 - kShapeBits)) | (far << kKeyApartBit) | ((uint64_t)mn << 32) | h;
}

__global__ void k_row_keys(uint64_t N1, int32_t M, const uint64_t* __restrict__ row_ptr,
                           const int32_t* __restrict__ sid, const double* __restrict__ cp, int range_bits, int apart, int split,
                           const unsigned char* __restrict__ also_apart, uint64_t* keys, uint32_t* vals, int* err,
                           unsigned long long* n_split = nullptr /* [0] rows that split, [1] their far alignments */) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N1) return;
    int e = 0;
    uint32_t nf = 0;
    uint64_t key = row_key_of(i, M, row_ptr, sid, cp, range_bits, apart, &e, split, &nf);
    if (n_split && nf) { atomicAdd(&n_split[0], 1ull); atomicAdd(&n_split[1], (unsigned long long)nf); }
    // (a read the first layout found outside its unit's window; split rows have no such bit: their key's hash is all 32 bits)
    if (also_apart && also_apart[i] && (int)(key >> (64 - kShapeBits)) < 2 * kShapesPerFmt) key |= 1ull << kKeyApartBit;
    if (e) *err = e;
    if (e == 1) return;
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

__global__ void k_shape_bounds(uint64_t N1, const uint64_t* __restrict__ keys, uint32_t* first) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N1) return;
    int sh = (int)(keys[p] >> (64 - kShapeBits));
    if (p == 0 || sh != (int)(keys[p - 1] >> (64 - kShapeBits))) first[sh] = (uint32_t)p;
}

__device__ inline int find_shape_by_row(const Shape* shapes, int n, uint32_t p) {
    int sh = 0;
    while (sh + 1 < n && p >= shapes[sh + 1].row_base) ++sh;
    return sh;
}

// sorted row p of shape S: scatter its alignments into the planes (values optional; sval = the value planes of all shapes,
// F64 or Q32 per shape; sexp = per-slot exponents of the Q32 reads)
// A split row (F64X shape) holds only the alignments whose id lies in [anchor, anchor + kLayoutWindow), in file order; the
// anchor is the one its sort key was made of.
// (reach = the `apart` the keys were made with: kLayoutWindow in the product, smaller in the CPU emulator's tests)
__host__ __device__ inline bool in_split_window(int32_t id, uint32_t anchor, int reach) { return (uint32_t)id - anchor < (uint32_t)reach; }

template <bool kIds>
__host__ __device__ inline void sell_fill_row(const Shape& S, uint32_t T, uint32_t p, const uint32_t* __restrict__ order,
                                              const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid,
                                              const double* __restrict__ cp, const double* __restrict__ ncp, int32_t* ssid,
                                              unsigned char* sval, double* sncp, int16_t* sexp, int* err, uint32_t anchor = 0,
                                              int reach = kLayoutWindow) {
    const int G = shape_G(S);
    uint32_t slice_local, r;
    row_to_slot(S, T, p - S.row_base, slice_local, r);
    uint32_t orig = order[p];
    uint64_t fr = row_ptr[orig];
    int L = (int)(row_ptr[orig + 1] - fr);
    const uint64_t pl_local = (uint64_t)slice_local * S.K * 64;   // first entry of the slice, within the shape
    const uint64_t pl0 = S.plane_base * 64 + pl_local;
    const uint32_t slot = S.slot_base + slice_local * shape_R(S) + r;
    Q32Scale q{0};
    if (cp && S.fmt == kFmtQ32) {
        double vmx = 0.0;
        for (int c = 0; c < L; c++) vmx = fmax(vmx, cp[fr + c]);
        // the format was decided from these very values (k_row_keys); anything else means the caller changed them
        if (!q32_scale_of(vmx, vmx, 0, q)) { if (err) *err = 3; q.e = 0; }
        if (sexp) sexp[slot] = (int16_t)q.e;
    }
    int ci = 0;  // position within the row: = c, except in a split row (its in-window alignments only)
    for (int c = 0; c < L; c++) {
        if (S.fmt == kFmtF64X && !in_split_window(sid[fr + c], anchor, reach)) continue;
        const uint64_t off = (uint64_t)(ci >> S.lg) * 64 + r * G + (ci & (G - 1));
        ++ci;
        if (kIds) ssid[pl0 + off] = sid[fr + c];
        if (cp) {
            if (S.fmt == kFmtQ32) ((uint32_t*)(sval + S.val_base))[pl_local + off] = q32_mantissa(cp[fr + c], q.e);
            else ((double*)(sval + S.val_base))[pl_local + off] = cp[fr + c];
        }
    }
    if (ncp) sncp[slot] = ncp[orig];
}

// The inverse of sell_fill_row for F64 rows: the caller-order CSR (ids and values) of sorted row p, read back from the planes --
// the same doubles, so a CSR that was released to save memory (rsem_em_set_option "release_csr") is restored bit for bit.
__global__ void k_unfill_sell(const Shape* __restrict__ shapes, int n_shapes, uint32_t T, uint32_t n_sell_rows,
                              const uint32_t* __restrict__ order, const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ ssid,
                              const unsigned char* __restrict__ sval, int32_t* sid, double* cp) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_sell_rows) return;
    const Shape S = shapes[find_shape_by_row(shapes, n_shapes, p)];
    const int G = shape_G(S);
    uint32_t slice_local, r;
    row_to_slot(S, T, p - S.row_base, slice_local, r);
    const uint32_t orig = order[p];
    const uint64_t fr = row_ptr[orig];
    const int L = (int)(row_ptr[orig + 1] - fr);
    const uint64_t pl_local = (uint64_t)slice_local * S.K * 64;
    const uint64_t pl0 = S.plane_base * 64 + pl_local;
    for (int c = 0; c < L; c++) {
        const uint64_t off = (uint64_t)(c >> S.lg) * 64 + r * G + (c & (G - 1));
        sid[fr + c] = ssid[pl0 + off];
        cp[fr + c] = ((const double*)(sval + S.val_base))[pl_local + off];
    }
}

// d_xanchor: anchors of the split rows, indexed by sorted row - x_row_base (nullptr: no split rows)
template <bool kIds>
__global__ void k_fill_sell(const Shape* __restrict__ shapes, int n_shapes, uint32_t T, uint32_t n_sell_rows,
                            const uint32_t* __restrict__ order, const uint64_t* __restrict__ row_ptr,
                            const int32_t* __restrict__ sid, const double* __restrict__ cp,
                            const double* __restrict__ ncp, int32_t* ssid, unsigned char* sval, double* sncp,
                            int16_t* sexp, int* err, const uint32_t* __restrict__ xanchor, uint32_t x_row_base, int reach,
                            const uint32_t* __restrict__ xreach = nullptr /* [n_x_rows] a window of its own per split row (sell_refine_split_windows) */) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_sell_rows) return;
    const Shape S = shapes[find_shape_by_row(shapes, n_shapes, p)];
    const bool x = S.fmt == kFmtF64X;
    sell_fill_row<kIds>(S, T, p, order, row_ptr, sid, cp, ncp, ssid, sval, sncp, sexp, err, x ? xanchor[p - x_row_base] : 0u,
                        (x && xreach) ? (int)xreach[p - x_row_base] : reach);
}

// per slice: bit l set when lane l's read has a different sid tuple than the same lane's read in
// the previous slice of its block (= the previous read in sorted order), or starts a block
// lane of slice sl (within shape S): does its read's sid tuple differ from the same lane's in the previous slice of the block?
__host__ __device__ inline bool slice_lane_changed(const Shape& S, uint32_t T, uint32_t sl, int lane, const int32_t* __restrict__ ssid) {
    if (sl % T == 0) return true;
    const uint64_t pl0 = (S.plane_base + (uint64_t)sl * S.K) * 64;
    bool changed = false;
    for (int k = 0; k < S.K; k++) {
        int v = ssid[pl0 + (uint64_t)k * 64 + lane];
        int pv = ssid[pl0 - (uint64_t)S.K * 64 + (uint64_t)k * 64 + lane];
        changed = changed || (pv != v);
    }
    return changed;
}
// the lanes of a read restart together: the bits of `lane`'s read in a 64-lane mask
__host__ __device__ inline unsigned long long read_lanes_of(const Shape& S, int lane) {
    const int G = shape_G(S);
    const int gb = lane & ~(G - 1);
    return (G == 64) ? ~0ull : (((1ull << G) - 1) << gb);
}

__global__ void k_slice_masks(const Shape* __restrict__ shapes, int n_shapes, uint32_t T, uint32_t n_slices,
                              const int32_t* __restrict__ ssid, unsigned long long* masks) {
    __shared__ Shape sh_shapes[kMaxShapes];
    for (int i = threadIdx.x; i < n_shapes; i += blockDim.x) sh_shapes[i] = shapes[i];
    __syncthreads();
    int lane = threadIdx.x & 63;
    uint32_t s = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    int sh = 0;
    while (sh + 1 < n_shapes && s >= sh_shapes[sh + 1].slice_base) ++sh;
    const Shape S = sh_shapes[sh];
    const bool changed = slice_lane_changed(S, T, s - S.slice_base, lane, ssid);
    unsigned long long m = __ballot(changed);
    m = __ballot((m & read_lanes_of(S, lane)) != 0);
    if (lane == 0) masks[s] = m;
}

// How many sid planes does a pass over the layout fetch?  A slice's sid planes are loaded from HBM only when some lane
// starts a new tuple there (mask != 0); the other slices re-read the shape's first sid slice (cache hits).  Planes of the
// slices with a set bit, summed over the layout: the "physical bytes" of the roofline (bench.py) come from this.
__global__ void k_count_sid_planes(const Shape* __restrict__ shapes, int n_shapes, uint32_t n_slices,
                                   const unsigned long long* __restrict__ masks, unsigned long long* out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long k = 0;
    if (s < n_slices && masks[s] != 0ull) {
        int sh = 0;
        while (sh + 1 < n_shapes && s >= shapes[sh + 1].slice_base) ++sh;
        k = (unsigned long long)shapes[sh].K;
    }
    for (int d = 32; d >= 1; d >>= 1) k += __shfl_xor(k, d);
    if ((threadIdx.x & 63) == 0 && k) atomicAdd(out, k);
}

// anchor sid (row_key_of) of the read in row slot 0 of every slice (non-decreasing along the blocks of a shape, except
// where its far-reaching reads begin)
__global__ void k_slice_minsid(const Shape* __restrict__ shapes, int n_shapes, uint32_t T, uint32_t n_slices,
                               const uint64_t* __restrict__ keys_sorted, uint32_t* slice_minsid) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slices) return;
    int sh = 0;
    while (sh + 1 < n_shapes && s >= shapes[sh + 1].slice_base) ++sh;
    const Shape S = shapes[sh];
    const uint32_t sl = s - S.slice_base, R = shape_R(S);
    uint32_t q = (sl / T) * R * T + sl % T;  // row slot 0 of this slice
    slice_minsid[s] = (uint32_t)((keys_sorted[S.row_base + q] >> 32) & kKeyMinSidCap);
}

// ... and the anchor of the read in the slice's LAST occupied row slot: the largest anchor of the slice (the rows of a block are sorted
// by anchor and laid out lane-major, so row slot r holds later rows than row slot r - 1).  sell_build_units needs it to see when the
// anchors of a would-be unit alone outrun one LDS window.
__global__ void k_slice_maxanchor(const Shape* __restrict__ shapes, int n_shapes, uint32_t T, uint32_t n_slices,
                                  const uint64_t* __restrict__ keys_sorted, uint32_t* slice_maxanchor) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slices) return;
    int sh = 0;
    while (sh + 1 < n_shapes && s >= shapes[sh + 1].slice_base) ++sh;
    const Shape S = shapes[sh];
    const uint32_t sl = s - S.slice_base;
    uint32_t q = 0, best = 0;
    bool any = false;
    for (int r = (int)shape_R(S) - 1; r >= 0 && !any; r--)
        if (slot_to_row(S, T, sl, (uint32_t)r, q)) { best = q; any = true; }
    slice_maxanchor[s] = any ? (uint32_t)((keys_sorted[S.row_base + best] >> 32) & kKeyMinSidCap) : 0u;
}

// largest sid of every slice that a window starting at the slice's own anchor could still hold (for the extent of a unit's
// LDS windows: a unit starts at or below the anchors of its slices, so ids beyond anchor + window_cap are outside anyway,
// and a far-away foreign id must not stretch the window of a small gene to the full capacity)
__global__ void k_slice_maxsid(const Shape* __restrict__ shapes, int n_shapes, uint32_t n_slices,
                               const int32_t* __restrict__ ssid, const uint32_t* __restrict__ slice_minsid, int window_cap,
                               uint32_t* slice_maxsid) {
    __shared__ Shape sh_shapes[kMaxShapes];
    for (int i = threadIdx.x; i < n_shapes; i += blockDim.x) sh_shapes[i] = shapes[i];
    __syncthreads();
    int lane = threadIdx.x & 63;
    uint32_t s = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    int sh = 0;
    while (sh + 1 < n_shapes && s >= sh_shapes[sh + 1].slice_base) ++sh;
    const Shape S = sh_shapes[sh];
    uint64_t pl0 = (S.plane_base + (uint64_t)(s - S.slice_base) * S.K) * 64;
    int mx = 0;
    const long long lim = (long long)slice_minsid[s] + window_cap;
    for (int k = 0; k < S.K; k++) {
        const int v = ssid[pl0 + (uint64_t)k * 64 + lane];
        if ((long long)v < lim) mx = max(mx, v);
    }
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
    if (lane == 0) slice_maxsid[s] = (uint32_t)mx;
}

template <typename T>
hipError_t dmalloc(T** p, size_t n) { return hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)); }

// ---- split rows: the far entries -------------------------------------------------------------------------------------------
// Per round the E step handles them in two more passes (em.hip): BEFORE the lane kernel, split row x gets
// extra[x] = sum over its far entries of theta[sid] * conprb (row order: far_ptr / far_sid / far_cp), which the lane kernel
// adds to the row's normaliser; AFTER it, every far entry adds theta[sid] * conprb / normaliser to counts[sid] in COLUMN order
// (csc_*: sorted by transcript id, so the sums per id are segmented reductions and the counts see a handful of atomics) --
// the "transposed CSC pass" instead of one global atomic per entry.
__global__ void k_x_anchors(uint32_t n_x, uint32_t x_row_base, const uint64_t* __restrict__ keys_sorted, uint32_t* xanchor) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n_x) xanchor[x] = (uint32_t)((keys_sorted[x_row_base + x] >> 32) & kKeyMinSidCap);
}
// The far entries are kept in ROW SLOT order (index xs = slot - x_slot_base, empty slots included): the pass that sums them
// per read then reads and writes contiguous memory, as the lane kernel reads extra[] / writes inv[] by slot.  (In sorted-row
// order -- lane-major, a slot apart is 512 B apart -- that pass made ten million scattered 8-byte stores per round at
// configs[1]'s size without gene structure: 0.41 ms, profiles/r04e_call.log.)
// far entries of split row x: count into nfar[xs] (far_ptr == nullptr) or write at far_ptr[xs]
__global__ void k_x_far(const Shape* __restrict__ shapes, int n_shapes, uint32_t T, uint32_t n_x, uint32_t x_row_base, uint32_t x_slot_base,
                        const uint32_t* __restrict__ order, const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid,
                        const uint32_t* __restrict__ xanchor, uint64_t* nfar, const uint64_t* __restrict__ far_ptr, int32_t* far_sid, uint64_t* far_src,
                        uint32_t* far_eslot, int reach, const uint32_t* __restrict__ xreach = nullptr) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_x) return;
    if (xreach) reach = (int)xreach[x];
    const uint32_t p = x_row_base + x, orig = order[p], anchor = xanchor[x];
    const uint64_t fr = row_ptr[orig], to = row_ptr[orig + 1];
    const Shape S = shapes[find_shape_by_row(shapes, n_shapes, p)];
    uint32_t slice_local, r;
    row_to_slot(S, T, p - S.row_base, slice_local, r);
    const uint32_t slot = S.slot_base + slice_local * shape_R(S) + r, xs = slot - x_slot_base;
    if (!far_ptr) {
        uint64_t n = 0;
        for (uint64_t j = fr; j < to; j++) n += in_split_window(sid[j], anchor, reach) ? 0 : 1;
        nfar[xs] = n;
        return;
    }
    uint64_t e = far_ptr[xs];
    for (uint64_t j = fr; j < to; j++)
        if (!in_split_window(sid[j], anchor, reach)) { far_sid[e] = sid[j]; far_src[e] = j; far_eslot[e] = slot; ++e; }
}
// Column order = (block of row slots, transcript id): within a block of 2^kCscSlotBlockLg slots the reciprocals the column
// pass gathers are 512 KB of memory -- they stay in the L2 of whichever XCD asks -- where a gather over all the slots of
// configs[1]'s size (80 MB) went to the Infinity Cache for every entry (0.68 ms per round).  Smaller blocks mean more
// (block, id) runs, i.e. more atomics: 2^12 ... 2^31 slots measured, 2^15-2^16 is the minimum (0.335 ms; profiles/r04j_call.log,
// r04j2_call.log; with the pass's workgroups dealt to the XCDs in contiguous eighths 2^16 / 2^17 / 2^18 are the same to 0.4 %,
// 2^20 is 20 % slower: profiles/r04p_call.log).  Measured and dropped: workgroup tasks per range of 2048 ids with an LDS window (0.70 ms whether the
// windows leave by atomics or as rows summed afterwards: a workgroup walks its piece one dependent step at a time,
// profiles/r04k_call.log, r04k2_call.log).
constexpr int kCscSlotBlockLg = 16;
__global__ void k_x_csc_keys(uint64_t n_far, uint32_t x_slot_base, const int32_t* __restrict__ far_sid, const uint32_t* __restrict__ far_eslot,
                             uint64_t* keys, uint64_t* vals, int block_lg) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_far) return;
    keys[e] = ((uint64_t)((far_eslot[e] - x_slot_base) >> block_lg) << 32) | (uint32_t)far_sid[e];
    vals[e] = e;
}
__global__ void k_x_csc(uint64_t n_far, const uint64_t* __restrict__ perm, const int32_t* __restrict__ far_sid, const uint64_t* __restrict__ far_src,
                        const uint32_t* __restrict__ far_eslot, int32_t* csc_sid, uint64_t* csc_src, uint32_t* csc_slot) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_far) return;
    const uint64_t e = perm[i];
    csc_sid[i] = far_sid[e];
    csc_src[i] = far_src[e];
    csc_slot[i] = far_eslot[e];
}
__global__ void k_x_values(uint64_t n_far, const double* __restrict__ cp, const uint64_t* __restrict__ far_src, const uint64_t* __restrict__ csc_src,
                           double* far_cp, double* csc_cp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_far) return;
    far_cp[i] = cp[far_src[i]];
    csc_cp[i] = cp[csc_src[i]];
}

struct SellLayout {
    uint64_t N1 = 0;
    uint32_t T = 16;              // slices per block (one wave's unit of work)
    uint32_t* d_order = nullptr;  // sorted row -> caller row
    uint32_t n_sell_rows = 0;     // sorted rows that live in the sliced layout
    uint32_t n_long_rows = 0;     // rows > 256 alignments (tail of d_order)
    Shape h_shapes[kMaxShapes];
    int n_shapes = 0;
    Shape* d_shapes = nullptr;
    uint32_t n_slices = 0;
    uint32_t n_slots = 0;
    uint64_t n_planes = 0;
    uint64_t val_bytes = 0;       // value planes of all shapes (Shape::val_base points into them)
    uint32_t n_q32_rows = 0;      // sorted rows held in Q32 shapes
    uint64_t n_q32_planes = 0;
    uint64_t n_sid_planes_loaded = 0;  // sid planes of the slices in which some lane starts a new tuple (k_count_sid_planes)
    // split rows (F64X shapes: the last shapes of the table) and their far entries (sell_build_far)
    uint32_t x_row_base = 0, n_x_rows = 0;   // sorted rows [x_row_base, x_row_base + n_x_rows)
    uint32_t x_slot_base = 0;                // their row slots start here (extra / inv arrays are indexed by slot - x_slot_base)
    uint64_t n_far = 0;                      // alignments of split rows outside their window
    uint32_t* d_xanchor = nullptr;           // [n_x_rows]
    uint32_t* d_xreach = nullptr;            // [n_x_rows] ids [anchor, anchor + reach) stay in the row (nullptr: kLayoutWindow for every row)
    uint32_t n_x_slots = 0;                  // n_slots - x_slot_base
    uint64_t* d_far_ptr = nullptr;           // [n_x_slots + 1] far entries of the split row in slot x_slot_base + xs, in file order
    int32_t* d_far_sid = nullptr;            // [n_far]
    uint64_t* d_far_src = nullptr;           // [n_far] index into the caller's CSR
    double* d_far_cp = nullptr;              // [n_far] values (sell_fill_values)
    int32_t* d_csc_sid = nullptr;            // the same entries sorted by transcript id ...
    uint64_t* d_csc_src = nullptr;
    uint32_t* d_csc_slot = nullptr;          // ... each with the row slot of its read
    double* d_csc_cp = nullptr;
    int32_t* d_ssid = nullptr;
    unsigned long long* d_masks = nullptr;
    uint32_t* d_slice_minsid = nullptr;
    uint32_t* d_slice_maxsid = nullptr;
    uint32_t* d_slice_maxanchor = nullptr;
};

inline void sell_free(SellLayout& L) {
    hipFree(L.d_order); hipFree(L.d_shapes); hipFree(L.d_ssid); hipFree(L.d_masks); hipFree(L.d_slice_minsid); hipFree(L.d_slice_maxsid); hipFree(L.d_slice_maxanchor);
    hipFree(L.d_xanchor); hipFree(L.d_xreach); hipFree(L.d_far_ptr); hipFree(L.d_far_sid); hipFree(L.d_far_src); hipFree(L.d_far_cp);
    hipFree(L.d_csc_sid); hipFree(L.d_csc_src); hipFree(L.d_csc_slot); hipFree(L.d_csc_cp);
    L = SellLayout();
}

// (re)write the value planes / per-slot noise values (and exponents of the Q32 reads) from the caller-order arrays.
// d_sval: L.val_bytes bytes (= n_planes * 512 for a layout without Q32 shapes); d_sexp / d_err only with Q32 shapes.
inline int sell_fill_values(const SellLayout& L, hipStream_t st, const uint64_t* d_row_ptr, const double* d_cp,
                            const double* d_ncp, void* d_sval, double* d_sncp, int16_t* d_sexp = nullptr,
                            int* d_err = nullptr, const int32_t* d_sid = nullptr) {
    RSEM_HIP_TRY(hipMemsetAsync(d_sval, 0, L.val_bytes, st));
    RSEM_HIP_TRY(hipMemsetAsync(d_sncp, 0, sizeof(double) * L.n_slots, st));
    if (d_sexp) RSEM_HIP_TRY(hipMemsetAsync(d_sexp, 0, sizeof(int16_t) * L.n_slots, st));
    if (L.n_x_rows && !d_sid) { rsem::set_last_error("a layout with split rows needs the transcript ids to place their values"); return RSEM_ERR_STATE; }
    if (L.n_sell_rows) {
        hipLaunchKernelGGL(k_fill_sell<false>, dim3(rsem::ceil_div(L.n_sell_rows, kBlock)), dim3(kBlock), 0, st,
                           L.d_shapes, L.n_shapes, L.T, L.n_sell_rows, L.d_order, d_row_ptr, d_sid,
                           d_cp, d_ncp, (int32_t*)nullptr, (unsigned char*)d_sval, d_sncp, d_sexp, d_err, (const uint32_t*)L.d_xanchor, L.x_row_base, kLayoutWindow,
                           (const uint32_t*)L.d_xreach);
        RSEM_HIP_TRY(hipGetLastError());
    }
    if (L.n_far) {
        hipLaunchKernelGGL(k_x_values, dim3(rsem::ceil_div(L.n_far, kBlock)), dim3(kBlock), 0, st, L.n_far, d_cp, (const uint64_t*)L.d_far_src,
                           (const uint64_t*)L.d_csc_src, L.d_far_cp, L.d_csc_cp);
        RSEM_HIP_TRY(hipGetLastError());
    }
    return RSEM_OK;
}

// The far entries of the split rows, in row order and in column (transcript id) order.  Called by sell_build once the shape
// table and the sorted order stand.
// d_keys_sorted == nullptr: once more for a layout whose anchors stand (L.d_xanchor) -- the rows' windows have changed (L.d_xreach).
inline int sell_build_far(SellLayout& L, hipStream_t st, const uint64_t* d_row_ptr, const int32_t* d_sid, const uint64_t* d_keys_sorted) {
    const uint32_t nx = L.n_x_rows;
    if (!nx) return RSEM_OK;
    const uint32_t nxs = L.n_slots - L.x_slot_base;
    L.n_x_slots = nxs;
    if (d_keys_sorted) {
        RSEM_HIP_TRY(dmalloc(&L.d_xanchor, nx));
        hipLaunchKernelGGL(k_x_anchors, dim3(rsem::ceil_div(nx, kBlock)), dim3(kBlock), 0, st, nx, L.x_row_base, d_keys_sorted, L.d_xanchor);
    } else {
        hipFree(L.d_far_ptr); hipFree(L.d_far_sid); hipFree(L.d_far_src); hipFree(L.d_far_cp);
        hipFree(L.d_csc_sid); hipFree(L.d_csc_src); hipFree(L.d_csc_slot); hipFree(L.d_csc_cp);
        L.d_far_ptr = nullptr; L.d_far_sid = nullptr; L.d_far_src = nullptr; L.d_far_cp = nullptr;
        L.d_csc_sid = nullptr; L.d_csc_src = nullptr; L.d_csc_slot = nullptr; L.d_csc_cp = nullptr;
        L.n_far = 0;
    }
    RSEM_HIP_TRY(dmalloc(&L.d_far_ptr, (size_t)nxs + 1));
    uint64_t* d_n = nullptr;
    void* d_tmp = nullptr;
    uint64_t *d_perm_in = nullptr, *d_perm = nullptr, *d_k_in = nullptr, *d_k_out = nullptr;
    uint32_t* d_eslot = nullptr;
    auto cleanup = [&]() { hipFree(d_n); hipFree(d_tmp); hipFree(d_perm_in); hipFree(d_perm); hipFree(d_k_in); hipFree(d_k_out); hipFree(d_eslot); };
    hipError_t e = dmalloc(&d_n, (size_t)nxs + 1);
    if (e == hipSuccess) e = hipMemsetAsync(d_n, 0, sizeof(uint64_t) * ((size_t)nxs + 1), st);
    if (e != hipSuccess) { cleanup(); RSEM_HIP_TRY(e); }
    hipLaunchKernelGGL(k_x_far, dim3(rsem::ceil_div(nx, kBlock)), dim3(kBlock), 0, st, (const Shape*)L.d_shapes, L.n_shapes, L.T, nx, L.x_row_base, L.x_slot_base,
                       (const uint32_t*)L.d_order, d_row_ptr, d_sid, (const uint32_t*)L.d_xanchor, d_n, (const uint64_t*)nullptr, (int32_t*)nullptr,
                       (uint64_t*)nullptr, (uint32_t*)nullptr, kLayoutWindow, (const uint32_t*)L.d_xreach);
    size_t tb = 0;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_n, L.d_far_ptr, (size_t)nxs + 1, st);
    if (e == hipSuccess) e = hipMalloc(&d_tmp, tb ? tb : 1);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, d_n, L.d_far_ptr, (size_t)nxs + 1, st);
    uint64_t nf = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&nf, L.d_far_ptr + nxs, sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { cleanup(); RSEM_HIP_TRY(e); }
    L.n_far = nf;
    e = dmalloc(&L.d_far_sid, nf);
    if (e == hipSuccess) e = dmalloc(&L.d_far_src, nf);
    if (e == hipSuccess) e = dmalloc(&L.d_far_cp, nf);
    if (e == hipSuccess) e = dmalloc(&L.d_csc_sid, nf);
    if (e == hipSuccess) e = dmalloc(&L.d_csc_src, nf);
    if (e == hipSuccess) e = dmalloc(&L.d_csc_slot, nf);
    if (e == hipSuccess) e = dmalloc(&L.d_csc_cp, nf);
    if (e == hipSuccess) e = dmalloc(&d_eslot, nf);
    if (e == hipSuccess) e = hipMemsetAsync(L.d_far_cp, 0, sizeof(double) * std::max<uint64_t>(nf, 1), st);
    if (e == hipSuccess) e = hipMemsetAsync(L.d_csc_cp, 0, sizeof(double) * std::max<uint64_t>(nf, 1), st);
    if (e != hipSuccess) { cleanup(); RSEM_HIP_TRY(e); }
    hipLaunchKernelGGL(k_x_far, dim3(rsem::ceil_div(nx, kBlock)), dim3(kBlock), 0, st, (const Shape*)L.d_shapes, L.n_shapes, L.T, nx, L.x_row_base, L.x_slot_base,
                       (const uint32_t*)L.d_order, d_row_ptr, d_sid, (const uint32_t*)L.d_xanchor, (uint64_t*)nullptr, (const uint64_t*)L.d_far_ptr,
                       L.d_far_sid, L.d_far_src, d_eslot, kLayoutWindow, (const uint32_t*)L.d_xreach);
    if (nf) {  // column order: a stable sort of the entries by (block of row slots, transcript id)
        hipFree(d_tmp); d_tmp = nullptr;
        e = dmalloc(&d_perm_in, nf);
        if (e == hipSuccess) e = dmalloc(&d_perm, nf);
        if (e == hipSuccess) e = dmalloc(&d_k_in, nf);
        if (e == hipSuccess) e = dmalloc(&d_k_out, nf);
        if (e != hipSuccess) { cleanup(); RSEM_HIP_TRY(e); }
        int block_lg = kCscSlotBlockLg;
        if (const char* eb = getenv("RSEM_HIP_CSC_BLOCK_LG")) block_lg = std::min(31, std::max(8, atoi(eb)));  // measurement knob
        hipLaunchKernelGGL(k_x_csc_keys, dim3(rsem::ceil_div(nf, kBlock)), dim3(kBlock), 0, st, nf, L.x_slot_base, (const int32_t*)L.d_far_sid,
                           (const uint32_t*)d_eslot, d_k_in, d_perm_in, block_lg);
        tb = 0;
        e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint64_t*)d_k_in, d_k_out, (const uint64_t*)d_perm_in, d_perm, nf, 0, 64, st);
        if (e == hipSuccess) e = hipMalloc(&d_tmp, tb ? tb : 1);
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, (const uint64_t*)d_k_in, d_k_out, (const uint64_t*)d_perm_in, d_perm, nf, 0, 64, st);
        if (e != hipSuccess) { cleanup(); RSEM_HIP_TRY(e); }
        hipLaunchKernelGGL(k_x_csc, dim3(rsem::ceil_div(nf, kBlock)), dim3(kBlock), 0, st, nf, (const uint64_t*)d_perm, (const int32_t*)L.d_far_sid,
                           (const uint64_t*)L.d_far_src, (const uint32_t*)d_eslot, L.d_csc_sid, L.d_csc_src, L.d_csc_slot);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    RSEM_HIP_TRY(e);
    return RSEM_OK;
}

// sort the rows, derive the shape table, scatter the sid planes and compute the per-slice masks.
// target_waves: how many wave-sized blocks the caller wants (sets T).  d_cp_for_q32 != nullptr: reads whose values
// qualify (q32_scale_of with range_bits) are placed in Q32 shapes.
inline int sell_build(SellLayout& L, hipStream_t st, uint64_t N1, int32_t M, const uint64_t* d_row_ptr,
                      const int32_t* d_sid, uint32_t target_waves, uint32_t forced_T = 0,
                      const double* d_cp_for_q32 = nullptr, int range_bits = 0, const unsigned char* d_also_apart = nullptr, int split = 0) {
    L.N1 = N1;
    int apart = kLayoutWindow;
    if (const char* e = getenv("RSEM_HIP_APART")) apart = atoi(e) ? kLayoutWindow : 0;  // measurement knob: 0 = one sorted sequence per shape
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_vals = nullptr, *d_first = nullptr;
    int* d_err = nullptr;
    void* d_tmp = nullptr;
    unsigned long long* d_ns = nullptr;
    auto cleanup = [&]() {
        hipFree(d_keys); hipFree(d_keys2); hipFree(d_vals); hipFree(d_first); hipFree(d_err); hipFree(d_tmp); hipFree(d_ns);
    };
    RSEM_HIP_TRY(dmalloc(&d_keys, N1));
    RSEM_HIP_TRY(dmalloc(&d_keys2, N1));
    RSEM_HIP_TRY(dmalloc(&d_vals, N1));
    RSEM_HIP_TRY(dmalloc(&L.d_order, N1));
    RSEM_HIP_TRY(dmalloc(&d_first, kShapeIds));
    RSEM_HIP_TRY(dmalloc(&d_err, 1));
    RSEM_HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int), st));
    RSEM_HIP_TRY(hipMemsetAsync(d_first, 0xff, kShapeIds * sizeof(uint32_t), st));
    if (N1) {
        // Split rows pay for two more passes and the kernel-sequence loop: taken only where they carry weight -- at least one
        // read in twenty would split (a read splits where most of its alignments lie outside its window: row_key_of).
        // Otherwise (a transcriptome with genes, a few stray multi-mappers) every read stays whole.
        int do_split = apart ? split : 0;
        if (do_split) {
            unsigned long long h_ns[2] = {0, 0};
            RSEM_HIP_TRY(dmalloc(&d_ns, 2));
            RSEM_HIP_TRY(hipMemsetAsync(d_ns, 0, 2 * sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_row_keys, dim3(rsem::ceil_div(N1, kBlock)), dim3(kBlock), 0, st, N1, M, d_row_ptr, d_sid,
                               d_cp_for_q32, range_bits, apart, do_split, d_also_apart, d_keys, d_vals, d_err, d_ns);
            RSEM_HIP_TRY(hipGetLastError());
            RSEM_HIP_TRY(hipMemcpyAsync(h_ns, d_ns, sizeof(h_ns), hipMemcpyDeviceToHost, st));
            RSEM_HIP_TRY(hipStreamSynchronize(st));
            if (h_ns[0] * 20ull < N1 && !getenv("RSEM_HIP_SPLIT_ALWAYS")) do_split = 0;
        }
        if (!do_split) {
            hipLaunchKernelGGL(k_row_keys, dim3(rsem::ceil_div(N1, kBlock)), dim3(kBlock), 0, st, N1, M, d_row_ptr, d_sid,
                               d_cp_for_q32, range_bits, apart, 0, d_also_apart, d_keys, d_vals, d_err, (unsigned long long*)nullptr);
            RSEM_HIP_TRY(hipGetLastError());
        }
        size_t tb = 0;
        RSEM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_keys, d_keys2, d_vals, L.d_order, N1, 0, 64, st));
        RSEM_HIP_TRY(hipMalloc(&d_tmp, tb ? tb : 1));
        RSEM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_keys, d_keys2, d_vals, L.d_order, N1, 0, 64, st));
        hipLaunchKernelGGL(k_shape_bounds, dim3(rsem::ceil_div(N1, kBlock)), dim3(kBlock), 0, st, N1, d_keys2, d_first);
        RSEM_HIP_TRY(hipGetLastError());
    }
    uint32_t h_first[kShapeIds];
    int h_err = 0;
    RSEM_HIP_TRY(hipMemcpyAsync(h_first, d_first, sizeof(h_first), hipMemcpyDeviceToHost, st));
    RSEM_HIP_TRY(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    RSEM_HIP_TRY(hipStreamSynchronize(st));
    if (h_err) {
        cleanup();
        rsem::set_last_error(h_err == 1 ? "row_ptr is not monotone" : "sid outside 1..M");
        return RSEM_ERR_INVALID;
    }
    L.n_shapes = 0;
    L.n_slices = 0;
    L.n_planes = 0;
    L.n_slots = 0;
    L.val_bytes = 0;
    L.n_q32_rows = 0;
    L.n_q32_planes = 0;
    L.n_x_rows = 0;
    L.x_row_base = 0;
    L.x_slot_base = 0;
    L.n_far = 0;
    L.n_x_slots = 0;
    uint32_t long_first = (h_first[kLongShape] == 0xffffffffu) ? (uint32_t)N1 : h_first[kLongShape];
    L.n_sell_rows = long_first;
    L.n_long_rows = (uint32_t)N1 - long_first;
    for (int id = 0; id < kMaxShapes; id++) {
        if (h_first[id] == 0xffffffffu) continue;
        uint32_t next = long_first;
        for (int j = id + 1; j < kMaxShapes; j++)
            if (h_first[j] != 0xffffffffu) { next = h_first[j]; break; }
        Shape& S = L.h_shapes[L.n_shapes++];
        S.fmt = id / kShapesPerFmt;
        S.lg = (id % kShapesPerFmt) / 4;
        S.K = id % 4 + 1;
        S.row_base = h_first[id];
        S.n_rows = next - h_first[id];
        uint32_t rps = shape_R(S);
        S.n_slices = (S.n_rows + rps - 1) / rps;
        S.slice_base = L.n_slices;
        S.plane_base = L.n_planes;
        S.slot_base = L.n_slots;
        S.val_base = L.val_bytes;
        L.n_slices += S.n_slices;
        L.n_planes += (uint64_t)S.n_slices * S.K;
        L.n_slots += S.n_slices * rps;
        L.val_bytes += (uint64_t)S.n_slices * S.K * plane_bytes(S.fmt);
        if (S.fmt == kFmtQ32) { L.n_q32_rows += S.n_rows; L.n_q32_planes += (uint64_t)S.n_slices * S.K; }
        if (S.fmt == kFmtF64X) {  // (the split shapes are the last ones of the table: their rows and slots are contiguous)
            if (!L.n_x_rows) { L.x_row_base = S.row_base; L.x_slot_base = S.slot_base; }
            L.n_x_rows += S.n_rows;
        }
    }
    // slices per block: enough blocks to fill the chip a few times over, long enough lane runs
    uint32_t T = forced_T ? forced_T : L.n_slices / std::max(1u, target_waves);
    L.T = std::min<uint32_t>(256, std::max<uint32_t>(forced_T ? 1 : 8, T));
    RSEM_HIP_TRY(dmalloc(&L.d_shapes, kMaxShapes));
    RSEM_HIP_TRY(hipMemcpyAsync(L.d_shapes, L.h_shapes, sizeof(Shape) * kMaxShapes, hipMemcpyHostToDevice, st));
    RSEM_HIP_TRY(dmalloc(&L.d_ssid, L.n_planes * 64));
    RSEM_HIP_TRY(dmalloc(&L.d_masks, (size_t)L.n_slices));
    RSEM_HIP_TRY(dmalloc(&L.d_slice_minsid, (size_t)L.n_slices));
    RSEM_HIP_TRY(dmalloc(&L.d_slice_maxsid, (size_t)L.n_slices));
    RSEM_HIP_TRY(dmalloc(&L.d_slice_maxanchor, (size_t)L.n_slices));
    RSEM_HIP_TRY(hipMemsetAsync(L.d_ssid, 0, sizeof(int32_t) * L.n_planes * 64, st));
    if (L.n_x_rows) {
        const int frc = sell_build_far(L, st, d_row_ptr, d_sid, d_keys2);
        if (frc != RSEM_OK) { cleanup(); return frc; }
    }
    if (L.n_sell_rows) {
        hipLaunchKernelGGL(k_fill_sell<true>, dim3(rsem::ceil_div(L.n_sell_rows, kBlock)), dim3(kBlock), 0, st,
                           L.d_shapes, L.n_shapes, L.T, L.n_sell_rows, L.d_order, d_row_ptr, d_sid, (const double*)nullptr,
                           (const double*)nullptr, L.d_ssid, (unsigned char*)nullptr, (double*)nullptr, (int16_t*)nullptr,
                           (int*)nullptr, (const uint32_t*)L.d_xanchor, L.x_row_base, kLayoutWindow);
        RSEM_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_slice_minsid, dim3(rsem::ceil_div(L.n_slices, kBlock)), dim3(kBlock), 0, st, L.d_shapes,
                           L.n_shapes, L.T, L.n_slices, d_keys2, L.d_slice_minsid);
        RSEM_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_slice_maxanchor, dim3(rsem::ceil_div(L.n_slices, kBlock)), dim3(kBlock), 0, st, L.d_shapes,
                           L.n_shapes, L.T, L.n_slices, d_keys2, L.d_slice_maxanchor);
        RSEM_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_slice_maxsid, dim3(rsem::ceil_div(L.n_slices, kBlock / 64)), dim3(kBlock), 0, st, L.d_shapes,
                           L.n_shapes, L.n_slices, L.d_ssid, L.d_slice_minsid, kLayoutWindow, L.d_slice_maxsid);
        RSEM_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_slice_masks, dim3(rsem::ceil_div(L.n_slices, kBlock / 64)), dim3(kBlock), 0, st,
                           L.d_shapes, L.n_shapes, L.T, L.n_slices, L.d_ssid, L.d_masks);
        RSEM_HIP_TRY(hipGetLastError());
        unsigned long long* d_cnt = (unsigned long long*)d_keys;  // (the unsorted keys are spent)
        RSEM_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_count_sid_planes, dim3(rsem::ceil_div(L.n_slices, kBlock)), dim3(kBlock), 0, st, L.d_shapes, L.n_shapes,
                           L.n_slices, (const unsigned long long*)L.d_masks, d_cnt);
        RSEM_HIP_TRY(hipGetLastError());
        unsigned long long h_cnt = 0;
        RSEM_HIP_TRY(hipMemcpyAsync(&h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
        RSEM_HIP_TRY(hipStreamSynchronize(st));
        L.n_sid_planes_loaded = h_cnt;
    }
    RSEM_HIP_TRY(hipStreamSynchronize(st));
    cleanup();
    return RSEM_OK;
}

// One workgroup's work: a run of consecutive slices of one shape, `per_wave` of them for each of its 4 waves, plus
// the base of its LDS windows = the smallest sid any of its reads can touch.  Most units are 4 whole blocks (one per
// wave); the last part of every shape is cut into half- and quarter-size units so that the launch does not end with
// a few long-lived workgroups (the tail of a launch costs about half a workgroup lifetime).
struct Unit {
    int32_t shape;
    uint32_t slice_begin;  // first slice, relative to the shape
    uint32_t n_slices;
    uint32_t per_wave;     // wave w walks slices [slice_begin + w * per_wave, + per_wave)
    int32_t base;
    int32_t span;          // sids [base, base + span) are staged in LDS (<= the window capacity)
    int32_t pad[2];        // pad[0]: 1 = some id of the unit lies outside [base, base + span), pad[1]: how many entries (sell_flag_far_units)
    Shape S;               // copy of the shape: one dependent load less at the start of every workgroup
};

inline int sell_build_units(const SellLayout& L, std::vector<Unit>& units, int window_cap) {
    std::vector<uint32_t> ms(L.n_slices), mx(L.n_slices), ma(L.n_slices);
    if (L.n_slices) {
        RSEM_HIP_TRY(hipMemcpy(ms.data(), L.d_slice_minsid, sizeof(uint32_t) * L.n_slices, hipMemcpyDeviceToHost));
        RSEM_HIP_TRY(hipMemcpy(mx.data(), L.d_slice_maxsid, sizeof(uint32_t) * L.n_slices, hipMemcpyDeviceToHost));
        RSEM_HIP_TRY(hipMemcpy(ma.data(), L.d_slice_maxanchor, sizeof(uint32_t) * L.n_slices, hipMemcpyDeviceToHost));
    }
    // fractions of a shape's blocks that go into full-size and half-size units (the rest: quarter-size)
    double f_full = 1.0, f_half = 0.0;
    if (const char* e = getenv("RSEM_HIP_TAPER")) {  // tuning knob: "full,half"
        double a = 0, b = 0;
        if (sscanf(e, "%lf,%lf", &a, &b) == 2 && a >= 0 && b >= 0 && a + b <= 1.0) { f_full = a; f_half = b; }
    }
    constexpr uint32_t W = 4;  // waves per workgroup
    units.clear();
    auto add = [&](int sh, const Shape& S, uint32_t sl0, uint32_t n, uint32_t per_wave) {
        Unit U;
        U.shape = sh;
        U.slice_begin = sl0;
        U.n_slices = n;
        U.per_wave = std::max<uint32_t>(per_wave, 1);
        const uint32_t s0 = S.slice_base + sl0, s1 = s0 + n;
        uint32_t top = 0, low = ms[s0];
        for (uint32_t t = s0; t < s1; t++) { top = std::max(top, mx[t]); low = std::min(low, ms[t]); }
        U.base = (int32_t)low;  // (= ms[s0], but for the one unit of a shape in which the far-reaching reads begin)
        const long long span = (long long)top - U.base + 1;
        U.span = (int32_t)std::min<long long>(std::max<long long>(span, 1), window_cap);
        U.pad[0] = U.pad[1] = 0;
        U.S = S;
        units.push_back(U);
    };
    for (int sh = 0; sh < L.n_shapes; sh++) {
        const Shape& S = L.h_shapes[sh];
        const uint32_t T = L.T, nb = (S.n_slices + T - 1) / T;
        // full units: 4 blocks; half units: 2 blocks (T/2 per wave); quarter units: 1 block (T/4 per wave)
        uint32_t b_full = (uint32_t)(nb * f_full) / W * W;
        uint32_t b_half = (uint32_t)(nb * f_half) / 2 * 2;
        if (T < 4) { b_full = nb / W * W; b_half = 0; }
        if (b_full + b_half > nb) b_half = (nb - b_full) / 2 * 2;
        uint32_t b = 0;
        auto emit = [&](uint32_t nblocks, uint32_t per_wave) {
            const uint32_t sl0 = b * T, n = std::min(S.n_slices, (b + nblocks) * T) - sl0;
            add(sh, S, sl0, n, per_wave);
            b += nblocks;
        };
        // A unit's LDS window holds window_cap ids from its smallest anchor on.  Where the reads of a shape are thin on the ground --
        // the split rows of reads that reach beyond their gene, one read in ten: their four blocks span ten times the ids four blocks
        // of compact reads do -- four blocks' anchors alone can outrun the window, and then every read behind it is "outside" and the
        // whole unit runs the loop with the global gather and atomics (configs[2] with 10 % such reads: all 390 units of split rows,
        // profiles/r06c_xrows_probe.log).  So a unit is as many blocks (4, 2 or 1) as fit one window.
        auto fits = [&](uint32_t nblocks) -> bool {
            const uint32_t s0 = S.slice_base + b * T, s1 = S.slice_base + std::min(S.n_slices, (b + nblocks) * T);
            uint32_t top = 0, low = 0xffffffffu;
            for (uint32_t t = s0; t < s1; t++) { top = std::max(top, std::max(mx[t], ma[t])); low = std::min(low, ms[t]); }
            return s1 <= s0 || (long long)top - (long long)low + 1 <= (long long)window_cap;
        };
        auto emit_fitting = [&](uint32_t want) {
            if (want >= W && fits(W)) emit(W, T);
            else if (want >= 2 && T >= 2 && fits(2)) emit(2, (T + 1) / 2);
            else if (T >= 4) emit(1, (T + 3) / 4);
            else emit(std::min<uint32_t>(want, nb - b), T);
        };
        while (b < b_full) emit_fitting(std::min<uint32_t>(W, b_full - b));
        while (b < b_full + b_half) emit_fitting(2);
        if (T < 4) while (b < nb) emit(std::min<uint32_t>(W, nb - b), T);
        else while (b < nb) emit(1, (T + 3) / 4);
    }
    // longest-processing-time-first: the hardware hands workgroups out in order
    std::stable_sort(units.begin(), units.end(), [&](const Unit& a, const Unit& b) {
        return (uint64_t)L.h_shapes[a.shape].K * a.n_slices > (uint64_t)L.h_shapes[b.shape].K * b.n_slices;
    });
    return RSEM_OK;
}

// Which units have an id outside their window?  The kernels run the others through a loop that never leaves LDS for theta
// and counts (estep_block.hpp, gibbs_block.hpp: template parameter kFar).  Empty plane entries hold id 0 and value 0: not
// counted.  One workgroup per unit; the flag lands in Unit::pad[0] on the device and in `units`.
__host__ __device__ inline bool unit_entry_is_far(const Unit& U, int32_t v) { return v > 0 && (unsigned)(v - U.base) >= (unsigned)U.span; }

__global__ __launch_bounds__(256) void k_unit_far(Unit* units, const int32_t* __restrict__ ssid) {
    __shared__ int n_far;
    if (threadIdx.x == 0) n_far = 0;
    __syncthreads();
    const Unit U = units[blockIdx.x];
    const uint64_t p0 = (U.S.plane_base + (uint64_t)U.slice_begin * U.S.K) * 64, n = (uint64_t)U.n_slices * U.S.K * 64;
    int mine = 0;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) mine += unit_entry_is_far(U, ssid[p0 + i]) ? 1 : 0;
    if (mine) atomicAdd(&n_far, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        units[blockIdx.x].pad[0] = n_far != 0;
        units[blockIdx.x].pad[1] = n_far;  // how many: the far-queue launch takes the units with few of them per slice (em.hip partition_units)
    }
}

inline int sell_flag_far_units(const SellLayout& L, std::vector<Unit>& units, Unit* d_units, hipStream_t st) {
    if (units.empty()) return RSEM_OK;
    hipLaunchKernelGGL(k_unit_far, dim3((unsigned)units.size()), dim3(256), 0, st, d_units, (const int32_t*)L.d_ssid);
    RSEM_HIP_TRY(hipGetLastError());
    RSEM_HIP_TRY(hipMemcpyAsync(units.data(), d_units, sizeof(Unit) * units.size(), hipMemcpyDeviceToHost, st));
    RSEM_HIP_TRY(hipStreamSynchronize(st));
    return RSEM_OK;
}

// ---- split rows: every row's window = its UNIT's window ----------------------------------------------------------------------------------
// A split row keeps the ids inside [anchor, anchor + kLayoutWindow) -- but the unit it lands in stages [base, base + span) only, as wide
// as its reads' own genes need, and ONE id beyond that sends the whole unit through the loop with the global gather and atomics (kFar:
// every wait of that instantiation is a wait for everything, estep_block.hpp).  With a few reads in a hundred carrying a stray id
// somewhere in the 2048 above their anchor that is every unit of split rows (configs[2] with 10 % cross-gene reads: 390 of 390,
// profiles/r06b_xrows_probe.log).  So, once the units are cut: a split row's window ends where its unit's does -- reach[x] = base +
// span - anchor[x] -- the ids beyond it join the row's far entries, the planes of the split rows are filled again, masks, far arrays
// and flags follow.  One workgroup per unit.
__global__ __launch_bounds__(256) void k_x_reach(const Unit* __restrict__ units, uint32_t T, const uint32_t* __restrict__ xanchor, uint32_t x_row_base, uint32_t* xreach) {
    const Unit U = units[blockIdx.x];
    const Shape& S = U.S;
    if (S.fmt != kFmtF64X) return;
    const uint32_t R = shape_R(S);
    for (uint32_t i = threadIdx.x; i < U.n_slices * R; i += blockDim.x) {
        uint32_t q;
        if (!slot_to_row(S, T, U.slice_begin + i / R, i % R, q)) continue;
        const uint32_t x = S.row_base + q - x_row_base;
        const long long top = (long long)U.base + U.span;  // (anchor >= base: the unit's base is the smallest anchor of its slices)
        const long long r = top - (long long)xanchor[x];
        // (a row whose anchor itself lies beyond the window -- a block wider than a window, sell_build_units -- keeps what it has: its
        // unit runs the far loop whatever this row gives up)
        xreach[x] = (uint32_t)(r < 1 ? kLayoutWindow : (r > kLayoutWindow ? kLayoutWindow : r));
    }
}

inline int sell_masks(SellLayout& L, hipStream_t st) {
    if (!L.n_slices) return RSEM_OK;
    hipLaunchKernelGGL(k_slice_masks, dim3(rsem::ceil_div(L.n_slices, kBlock / 64)), dim3(kBlock), 0, st, L.d_shapes, L.n_shapes, L.T, L.n_slices, L.d_ssid, L.d_masks);
    RSEM_HIP_TRY(hipGetLastError());
    unsigned long long* d_cnt = nullptr;
    RSEM_HIP_TRY(dmalloc(&d_cnt, 1));
    hipError_t e = hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_count_sid_planes, dim3(rsem::ceil_div(L.n_slices, kBlock)), dim3(kBlock), 0, st, L.d_shapes, L.n_shapes, L.n_slices,
                           (const unsigned long long*)L.d_masks, d_cnt);
        e = hipGetLastError();
    }
    unsigned long long h_cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_cnt);
    RSEM_HIP_TRY(e);
    L.n_sid_planes_loaded = h_cnt;
    return RSEM_OK;
}

inline int sell_refine_split_windows(SellLayout& L, hipStream_t st, const uint64_t* d_row_ptr, const int32_t* d_sid, std::vector<Unit>& units, Unit* d_units) {
    if (!L.n_x_rows || units.empty()) return RSEM_OK;
    size_t n_far_x = 0;
    for (const Unit& u : units) n_far_x += (u.S.fmt == kFmtF64X && u.pad[0] != 0) ? 1 : 0;
    if (!n_far_x) return RSEM_OK;
    RSEM_HIP_TRY(dmalloc(&L.d_xreach, L.n_x_rows));
    hipLaunchKernelGGL(k_x_reach, dim3((unsigned)units.size()), dim3(256), 0, st, (const Unit*)d_units, L.T, (const uint32_t*)L.d_xanchor, L.x_row_base, L.d_xreach);
    RSEM_HIP_TRY(hipGetLastError());
    // the id planes once more (the split rows' entries move up where an id left; everything else is written as before)
    RSEM_HIP_TRY(hipMemsetAsync(L.d_ssid, 0, sizeof(int32_t) * L.n_planes * 64, st));
    hipLaunchKernelGGL(k_fill_sell<true>, dim3(rsem::ceil_div(L.n_sell_rows, kBlock)), dim3(kBlock), 0, st, L.d_shapes, L.n_shapes, L.T, L.n_sell_rows, L.d_order,
                       d_row_ptr, d_sid, (const double*)nullptr, (const double*)nullptr, L.d_ssid, (unsigned char*)nullptr, (double*)nullptr, (int16_t*)nullptr,
                       (int*)nullptr, (const uint32_t*)L.d_xanchor, L.x_row_base, kLayoutWindow, (const uint32_t*)L.d_xreach);
    RSEM_HIP_TRY(hipGetLastError());
    int rc = sell_masks(L, st);
    if (rc == RSEM_OK) rc = sell_build_far(L, st, d_row_ptr, d_sid, nullptr);
    if (rc == RSEM_OK) rc = sell_flag_far_units(L, units, d_units, st);
    return rc;
}

// The reads of the units that are NOT made of far-reaching reads but still have an id outside the unit's window -- a read
// whose foreign id happens to lie within a window's width of its anchor (so the sort key kept it among the compact reads),
// while the unit's window, as wide as its reads' ids need, ends before it.  A handful per unit are enough to send the
// whole unit through the loop with the global gather (kFar).  They are marked here; a second layout sorts them behind the
// compact reads with the other far-reaching ones (k_row_keys: also_apart), and the compact units stay clean.
__global__ __launch_bounds__(256) void k_mark_stray_reads(const Unit* __restrict__ units, const int32_t* __restrict__ ssid, const uint32_t* __restrict__ order,
                                                          uint32_t T, int32_t M, const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid,
                                                          unsigned char* also_apart, unsigned long long* n_marked) {
    const Unit U = units[blockIdx.x];
    if (U.pad[0] == 0) return;
    const Shape& S = U.S;
    const uint64_t p0 = (S.plane_base + (uint64_t)U.slice_begin * S.K) * 64, n = (uint64_t)U.n_slices * S.K * 64;
    unsigned long long mine = 0;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
        if (!unit_entry_is_far(U, ssid[p0 + i])) continue;
        const uint32_t sl = U.slice_begin + (uint32_t)(i / ((uint64_t)S.K * 64));  // slice within the shape
        const uint32_t r = (uint32_t)(i % 64) >> S.lg;                            // row slot within the slice
        uint32_t q;
        if (!slot_to_row(S, T, sl, r, q)) continue;  // (cannot happen: an empty slot holds id 0)
        const uint32_t orig = order[S.row_base + q];
        int e = 0;
        const uint64_t key = row_key_of(orig, M, row_ptr, sid, nullptr, 0, kLayoutWindow, &e);
        if (((key >> kKeyApartBit) & 1ull) == 0ull && also_apart[orig] == 0) {  // (a far-reaching read is where it belongs already)
            also_apart[orig] = 1;
            ++mine;
        }
    }
    if (mine) atomicAdd(n_marked, mine);  // (a read with two such ids may count twice: only "any?" matters)
}

// sell_build + units + far flags, and once more with the stray reads sorted apart if there are any.  d_units: device copy of
// `units` (allocated here; the caller owns it).  RSEM_HIP_APART=0 switches the apart bit and this refinement off.
inline int sell_build_refined(SellLayout& L, hipStream_t st, uint64_t N1, int32_t M, const uint64_t* d_row_ptr, const int32_t* d_sid,
                              uint32_t target_waves, uint32_t forced_T, const double* d_cp_for_q32, int range_bits, int window_cap,
                              std::vector<Unit>& units, Unit** d_units, unsigned long long* n_strays = nullptr, int split = 0) {
    unsigned char* d_also = nullptr;
    unsigned long long* d_n = nullptr;
    if (n_strays) *n_strays = 0;
    const char* knob = getenv("RSEM_HIP_APART");
    const bool refine = !(knob && atoi(knob) == 0);
    for (int pass = 0; pass < 2; pass++) {
        int rc = sell_build(L, st, N1, M, d_row_ptr, d_sid, target_waves, forced_T, d_cp_for_q32, range_bits, d_also, split);
        if (rc == RSEM_OK) rc = sell_build_units(L, units, window_cap);
        if (rc != RSEM_OK) { (void)hipFree(d_also); return rc; }
        (void)hipFree(*d_units);
        *d_units = nullptr;
        hipError_t e = dmalloc(d_units, units.size());
        if (e == hipSuccess && !units.empty()) e = hipMemcpyAsync(*d_units, units.data(), sizeof(Unit) * units.size(), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { (void)hipFree(d_also); RSEM_HIP_TRY(e); }
        rc = sell_flag_far_units(L, units, *d_units, st);
        if (rc != RSEM_OK) { (void)hipFree(d_also); return rc; }
        size_t n_far_units = 0;
        for (const Unit& u : units) n_far_units += u.pad[0] != 0;
        if (pass == 1 || !n_far_units || N1 == 0 || !refine) break;
        unsigned long long n = 0;
        e = hipMalloc((void**)&d_also, N1);
        if (e == hipSuccess) e = hipMalloc((void**)&d_n, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(d_also, 0, N1, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_n, 0, sizeof(unsigned long long), st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_mark_stray_reads, dim3((unsigned)units.size()), dim3(256), 0, st, (const Unit*)*d_units, (const int32_t*)L.d_ssid,
                               (const uint32_t*)L.d_order, L.T, M, d_row_ptr, d_sid, d_also, d_n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&n, d_n, sizeof(n), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_n);
        if (e != hipSuccess) { (void)hipFree(d_also); RSEM_HIP_TRY(e); }
        if (n == 0) break;  // every far unit is made of far-reaching reads: nothing to gain
        // a second build costs what the first did: only where the stray reads matter -- one read in a thousand, or one unit
        // in a hundred running the loop with the global gather because of them
        if (n * 1000ull < N1 && n_far_units * 100 < units.size()) break;
        if (n_strays) *n_strays = n;
        sell_free(L);       // second pass with the marks
    }
    (void)hipFree(d_also);
    if (L.n_x_rows && !(getenv("RSEM_HIP_X_REFINE") && atoi(getenv("RSEM_HIP_X_REFINE")) == 0))  // (measurement knob: 0 = the rows keep their own windows)
        return sell_refine_split_windows(L, st, d_row_ptr, d_sid, units, *d_units);
    return RSEM_OK;
}

}  // namespace
