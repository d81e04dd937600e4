// sell_shape.hpp -- the shape record of the sliced layout (sell_layout.hpp) and the index arithmetic from a sorted read to its
// place in the planes.  Included INSIDE an anonymous namespace by sell_layout.hpp (em.hip, gibbs.hip, the CPU emulators) and
// by model.hip, whose round kernel writes the alignment probabilities straight into the value planes.
#pragma once

// value plane formats (sell_layout.hpp).  F64X: doubles, rows that are only the IN-WINDOW part of a read whose other
// alignments live in the far-entry side arrays (split rows); such a row carries an extra term in its normaliser and hands
// its reciprocal on.
constexpr int kFmtF64 = 0, kFmtQ32 = 1, kFmtF64X = 2;

struct Shape {
    uint64_t plane_base;  // first plane of this shape (one plane = 64 entries)
    uint32_t slice_base;  // first slice
    uint32_t n_slices;
    uint32_t row_base;    // first sorted row
    uint32_t n_rows;
    uint32_t slot_base;   // first row slot (slot = slice * rows_per_slice + r)
    int32_t K;            // planes per slice
    int32_t lg;           // log2(lanes per read)
    int32_t fmt;          // kFmtF64 / kFmtQ32
    uint64_t val_base;    // byte offset of this shape's value planes (512 B per F64 plane, 256 B per Q32 plane)
};

__host__ __device__ inline uint32_t plane_bytes(int fmt) { return fmt == kFmtQ32 ? 256u : 512u; }

__host__ __device__ inline int shape_G(const Shape& S) {  // lanes per read
    return 1 << S.lg;
}
__host__ __device__ inline uint32_t shape_R(const Shape& S) {  // reads per slice
    return 64u >> S.lg;
}

// sorted read q of a shape  ->  (slice within the shape, row slot within the slice)
__host__ __device__ inline void row_to_slot(const Shape& S, uint32_t T, uint32_t q, uint32_t& slice_local, uint32_t& r) {
    const uint32_t R = shape_R(S), rpb = R * T;
    const uint32_t b = q / rpb, qb = q % rpb;
    const uint32_t left = S.n_rows - b * rpb;
    const uint32_t nb = left < rpb ? left : rpb;
    const uint32_t Tb = (nb + R - 1) / R;
    r = qb / Tb;
    slice_local = b * T + qb % Tb;
}

